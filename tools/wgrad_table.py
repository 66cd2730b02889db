"""Per-layer weight-gradient efficiency at a bench configuration (dev tool): runs the Trainer's first steps with plan timing logged and
tabulates, per weight-gradient geometry, the chosen plan, its time (incl. the split reduction) and the fp32-MFMA fraction.
usage: python tools/wgrad_table.py [extra trainer args...]"""
import ast
import contextlib
import io
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from sqd import nnkernels  # noqa: E402
from trainer import Trainer  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402

nnkernels.TUNE_SPACE["log"] = True
opts = MonodepthOptions().parse(bench.CONFIG_B + ["--sqd_no_graph"] + sys.argv[1:])
tr = Trainer(opts)
tr.set_train()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    for i in range(2):
        tr.train_step(synthetic_batch(opts.batch_size, opts.height, opts.width, start=i * opts.batch_size, device=tr.device))
    torch.cuda.synchronize()
rows, uses = [], {}
for line in buf.getvalue().splitlines():
    m = re.match(r"sqd conv plan wgrad (\(.*?\)) (\(.*?\)) model splits (\d+)", line)
    if m:
        geom, best = ast.literal_eval(m.group(1)), ast.literal_eval(m.group(2))
        N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
        gflop = 2.0 * N * Ho * Wo * K * C * R * S / 1e9
        us = best[0] / 3 * 1e3
        rows.append((us, geom, best[1:], gflop))
# how often each geometry occurs in the model (convolutions sharing a geometry share the plan)
for m_ in tr.models.values():
    for mod in m_.modules():
        if hasattr(mod, "weight") and mod.weight is not None and mod.weight.dim() == 4:
            uses[tuple(mod.weight.shape)] = uses.get(tuple(mod.weight.shape), 0) + 1
for mode in ("fwd", "dgrad"):
    rows2 = []
    for line in buf.getvalue().splitlines():
        m = re.match(r"sqd conv plan %s (\(.*?\)) (\(.*?\))$" % mode, line)
        if m:
            geom, best = ast.literal_eval(m.group(1)), ast.literal_eval(m.group(2))
            N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
            gflop = 2.0 * N * Ho * Wo * K * C * R * S / 1e9
            rows2.append((best[0] / 3 * 1e3, geom, best[1:], gflop))
    print("\n## %s\n\n| us (1 call) | TFLOP/s | of 157.3 | plan (bm, bn, z, bk) | N Ho Wo C K RxS stride |\n|---:|---:|---:|---|---|" % mode)
    tu = tg = 0.0
    for us, geom, plan, gflop in sorted(rows2, reverse=True):
        N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
        print("| %.1f | %.1f | %.2f | %s | %d %d %d %d %d %dx%d s%d |" % (us, gflop / us * 1e3, gflop / us * 1e3 / 157.3, plan, N, Ho, Wo, C, K, R, S, stride))
        tu += us
        tg += gflop
    print("\nsum over distinct geometries: %.0f us, %.1f GFLOP, %.1f TFLOP/s" % (tu, tg, tg / max(tu, 1e-9) * 1e3))
print("\n## wgrad\n")
tot_us = tot_gf = 0.0
print("| us (1 call) | TFLOP/s | of 157.3 | plan (impl, splits) | N Ho Wo C K RxS stride |")
print("|---:|---:|---:|---|---|")
for us, geom, plan, gflop in sorted(rows, reverse=True):
    N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
    print("| %.1f | %.1f | %.2f | %s | %d %d %d %d %d %dx%d s%d |" % (us, gflop / us * 1e3, gflop / us * 1e3 / 157.3, plan, N, Ho, Wo, C, K, R, S, stride))
    tot_us += us
    tot_gf += gflop
print("\nsum over distinct geometries: %.0f us, %.1f GFLOP, %.1f TFLOP/s" % (tot_us, tot_gf, tot_gf / tot_us * 1e3))
