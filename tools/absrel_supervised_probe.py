#!/usr/bin/env python3
"""Supervised pre-fit (the metric-depth finetune step: SILog on the ground truth) of the ResNet-50 SQLdepth model on the synthetic "road"
scenes, then the self-supervised Trainer's held-out abs_rel with those weights: does it give weights whose abs_rel is well below 0.5?
(dev tool behind tests/test_gpu_abs_rel.py's comparison from trained weights).  usage: python tools/absrel_supervised_probe.py [steps] [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
from datasets.synthetic import synthetic_batch  # noqa: E402
from finetune.train_ft_SQLdepth import FinetuneArgs, FinetuneTrainer  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 192, 640
args = ["--backbone", "resnet", "--num_layers", "50", "--num_features", "256", "--model_dim", "32", "--patch_size", "16", "--query_nums", "64",
        "--dim_out", "64", "--height", str(H), "--width", str(W), "--batch_size", str(B), "--min_depth", "0.001", "--max_depth", "80.0",
        "--num_workers", "0", "--sqd_synthetic", "--sqd_device_noise", "--log_dir", "/tmp/sqd_absrel_probe"]
torch.manual_seed(0)
opt = MonodepthOptions().parse(args)
ft = FinetuneTrainer(opt, FinetuneArgs(bs=B, epochs=1, lr=1e-4), steps_per_epoch=steps)
NB = 32
batches = []
for i in range(NB):
    s = synthetic_batch(B, H, W, start=B * i, scene="road", with_gt=True, device="cuda")
    batches.append({"image": s[("color_aug", 0, 0)], "depth": F.interpolate(s["depth_gt"], [H, W], mode="nearest")})
held = synthetic_batch(4, H, W, start=10 ** 5, with_gt=True, scene="road", device="cuda")
tr = Trainer(MonodepthOptions().parse(args))


def held_out():
    tr.models["encoder"].load_state_dict(ft.model.encoder.state_dict())
    tr.models["depth"].load_state_dict(ft.model.depth_decoder.state_dict())
    tr.set_eval()
    with torch.no_grad():
        outputs, losses = tr.process_batch(dict(held))
        tr.compute_depth_losses(held, outputs, losses)
    return " ".join("%s %.4f" % (n.split("/")[-1], float(losses[n])) for n in tr.depth_metric_names)


for i in range(steps + 1):
    if i % 100 == 0:
        print("step %5d  held-out %s" % (i, held_out()), flush=True)
        ft.model.train()
    if i < steps:
        loss, _ = ft.train_step(batches[i % NB])
        if i % 100 == 0:
            print("           SILog %.5f" % float(loss), flush=True)
