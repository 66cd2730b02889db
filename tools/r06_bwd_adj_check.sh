# round 6: box7x9_adj in the photometric backward — same-box A/B against the build with the compiler-scheduled three-plane sums
# (tools/build_variant.sh adj3 photo_tile.hip -DSQD_BWD_ADJ3), outputs compared bit for bit
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06g; O=gpurun_out/r06g/ab_bwd_adj.txt; : > $O
for r in 1 2 3; do
  echo "adj3: $(timeout 200 python tools/bench_fused.py --which coef,bwd --lib tools/bin/libsqd_adj3.so --dump /tmp/g_adj3.pt 2>&1 | tail -1)" >> $O
  echo "adj9: $(timeout 200 python tools/bench_fused.py --which coef,bwd --dump /tmp/g_adj9.pt 2>&1 | tail -1)" >> $O
done
python - >> $O <<'PY'
import torch
a, b = torch.load("/tmp/g_adj3.pt"), torch.load("/tmp/g_adj9.pt")
print("outputs:", len(a), "bit-equal:", [bool(torch.equal(x, y)) for x, y in zip(a, b)], "max |diff|:", [float((x.float() - y.float()).abs().max()) for x, y in zip(a, b)])
PY
cut -c1-220 $O
