#!/usr/bin/env python3
"""What share of the pixels does an identity candidate win (auto-mask: no gradient — the coefficient planes of sqd_photo_coef are zeros there)?
Measured on the bench's synthetic configs[1] batches after a few training steps (dev tool behind DESIGN 3.2).  usage: python tools/identity_fraction.py [steps]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import bench  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
opts = MonodepthOptions().parse(bench.CONFIG_B)
tr = Trainer(opts)
tr.set_train()
for scene in ("waves", "road"):
    batches = [synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, start=i * opts.batch_size, device=tr.device, scene=scene) for i in range(3)]
    for i in range(steps):
        outputs, losses = tr.train_step(dict(batches[i % 3]))
        if i in (0, steps - 1):
            sel = outputs["identity_selection/0"]
            print("scene %-5s step %3d: reprojection wins %.3f of the pixels, an identity candidate %.3f (loss %.4f)"
                  % (scene, i, float(sel.mean()), 1.0 - float(sel.mean()), float(losses["loss"])), flush=True)
