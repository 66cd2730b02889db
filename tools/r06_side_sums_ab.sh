# round 6: do the BatchNorm sums taken by the passes that write a BatchNorm's whole gradient pay in the step?  same-box A/B, one switch at a time
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r06k; O=gpurun_out/r06k/side_sums_ab.txt; : > $O
for k in res pool upcat; do
  echo "== nnkernels.FUSE_BN_SIDE_SUMS[$k]=False" >> $O
  timeout 900 python tools/ab_bench.py "nnkernels.FUSE_BN_SIDE_SUMS[$k]=False" --rounds 3 2>&1 | tail -3 >> $O
done
cat $O
