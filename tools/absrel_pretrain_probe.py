#!/usr/bin/env python3
"""Does self-supervised training on the synthetic "road" scenes bring abs_rel down, and after how many steps?  (dev tool behind
tests/test_gpu_abs_rel.py's comparison from trained weights).  usage: python tools/absrel_pretrain_probe.py [steps] [batch] [backbone-args...]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
from datasets.synthetic import synthetic_batch  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 192, 640
args = ["--backbone", "resnet", "--num_layers", "50", "--num_features", "256", "--model_dim", "32", "--patch_size", "16", "--query_nums", "64",
        "--dim_out", "64", "--height", str(H), "--width", str(W), "--batch_size", str(B), "--min_depth", "0.001", "--max_depth", "80.0",
        "--num_workers", "0", "--sqd_synthetic", "--sqd_device_noise", "--log_dir", "/tmp/sqd_absrel_probe"] + sys.argv[3:]
torch.manual_seed(0)
tr = Trainer(MonodepthOptions().parse(args))
tr.set_train()
NB = 32
batches = [synthetic_batch(B, H, W, start=B * i, scene="road", device="cuda") for i in range(NB)]
held = synthetic_batch(4, H, W, start=10 ** 5, with_gt=True, scene="road", device="cuda")
for i in range(steps + 1):
    if i % 250 == 0:
        tr.set_eval()
        with torch.no_grad():
            outputs, losses = tr.process_batch(dict(held))
            tr.compute_depth_losses(held, outputs, losses)
        print("step %5d  held-out %s" % (i, " ".join("%s %.4f" % (n.split("/")[-1], float(losses[n])) for n in tr.depth_metric_names)), flush=True)
        tr.set_train()
    if i < steps:
        loss = float(tr.train_step(dict(batches[i % NB]))[1]["loss"])
        if i % 250 == 0:
            print("           train loss %.5f" % loss, flush=True)
