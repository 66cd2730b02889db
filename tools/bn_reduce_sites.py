#!/usr/bin/env python3
"""Which BatchNorm-backward nodes of a configs[1] step still run their own reduction pass (bn_reduce_kernel<1>) instead of taking the two sums
from a data-gradient epilogue?  (dev tool behind DESIGN 7: eager step, the node's input shape and what produced its gradient)"""
import os
import sys
import traceback

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import bench  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from sqd import nnkernels  # noqa: E402
from trainer import Trainer  # noqa: E402

opts = MonodepthOptions().parse(bench.CONFIG_B + ["--sqd_no_graph"])
tr = Trainer(opts)
tr.set_train()
names = {}
for net, m in tr.models.items():
    for n, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            names[mod.weight.data_ptr()] = "%s.%s" % (net, n)
orig = nnkernels.BatchNormAct.backward
log = []


def backward(ctx, dy):
    sh = getattr(ctx, "shared", None)
    x, mask, gamma = ctx.saved_tensors[:3]
    fused = sh is not None and sh.get("dx") is dy and sh.get("rows", 0) > 0
    log.append((names.get(gamma.data_ptr(), "?"), tuple(x.shape), fused, "no shared record" if sh is None else
                "gradient is another tensor" if sh.get("dx") is not dy else "no rows"))
    return orig(ctx, dy)


nnkernels.BatchNormAct.backward = staticmethod(backward)
batch = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
for i in range(2):
    log.clear()
    tr.train_step(dict(batch))
own = [l for l in log if not l[2]]
print("%d BatchNorm backward nodes, %d run their own reduction:" % (len(log), len(own)))
for n, shp, _, why in own:
    print("  %-50s %-22s %s" % (n, shp, why))
