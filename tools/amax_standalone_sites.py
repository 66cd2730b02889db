#!/usr/bin/env python3
"""Which tensors of a configs[1] step still get a stand-alone max |x| pass (sqd_amax) because their producer left no operand-scale record?
(dev tool: one eager step with the pinned plans; prints shape and the autograd node that produced the tensor where known)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import bench  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from sqd import nnkernels  # noqa: E402
from trainer import Trainer  # noqa: E402

opts = MonodepthOptions().parse(bench.bench_args() + ["--sqd_no_graph"])
tr = Trainer(opts)
tr.set_train()
log = []
orig = nnkernels.amax_of


def amax_of(t, site):
    tagged = nnkernels._amax_get(t) is not None
    if not tagged:
        log.append((site, tuple(t.shape), "grad_fn=%s" % (type(t.grad_fn).__name__ if t.grad_fn is not None else None)))
    return orig(t, site)


nnkernels.amax_of = amax_of
batch = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
for i in range(3):
    log.clear()
    tr.train_step(dict(batch))
print("%d stand-alone passes in the third step:" % len(log))
for l in log:
    print("  %-40s %-24s %s" % l)
