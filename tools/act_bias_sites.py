#!/usr/bin/env python3
"""Which convolution backward nodes of a configs[1] step run a stand-alone activation-gradient pass (sqd_act_bwd) and / or leave their bias
gradient to the two-stage column sums (colsum_kernel)?  (dev tool behind DESIGN 7: one eager step with the pinned plans)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import bench  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from sqd import nnkernels, lib as _l  # noqa: E402
from trainer import Trainer  # noqa: E402

opts = MonodepthOptions().parse(bench.bench_args() + ["--sqd_no_graph"])
tr = Trainer(opts)
tr.set_train()
names = {}
for net, m in tr.models.items():
    for n, p in m.named_parameters():
        names[p.data_ptr()] = "%s.%s" % (net, n)
log = []
orig = nnkernels.Conv2d.backward if hasattr(nnkernels, "Conv2d") else None
cls = [c for c in vars(nnkernels).values() if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c.__name__.startswith("Conv2d")]
print("conv nodes:", [c.__name__ for c in cls])
for c in cls:
    o = c.backward

    def backward(ctx, *a, _o=o, _c=c):
        if a[0] is not None and (getattr(ctx, "act", None) is not None or getattr(ctx, "has_bias", False)):
            N, H, W, C, K, R, S, stride, pad, Ho, Wo = ctx.geom
            log.append((_c.__name__, names.get(getattr(ctx, "wkey", None), "?"), tuple(ctx.geom), ctx.act, ctx.has_bias, nnkernels._colsum_get(a[0]) is not None,
                        nnkernels.CHOSEN_PLANS.get(("wgrad", N, Ho, Wo, C, K, R, S))))
        return _o(ctx, *a)
    c.backward = staticmethod(backward)
batch = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
for i in range(2):
    log.clear()
    tr.train_step(dict(batch))
print("%d convolution backward nodes with an activation or a bias:" % len(log))
for l in log:
    N, H, W, C, K, R, S, stride, pad, Ho, Wo = l[2]
    print("  %-14s %-44s x [%d,%d,%d,%d] -> K %d %dx%d/%d  dy %.1f MB  act %s  bias %s  colsum tag on dy %s  wgrad plan %s" % (l[0], l[1], N, C, H, W, K, R, S, stride, N * Ho * Wo * K * 4 / 1e6, l[3], l[4], l[5], l[6]))
