#!/usr/bin/env python3
"""Measure the convolution plans of configs[1] on this box and write the pinned set bench.py ships
(sfmnext-impl_amd/plans/configB_resnet50_192x640_b12.json): export_plans() of a Trainer whose first step timed every plan,
plus the hash of the convolution kernel sources the plans name kernels of.

    python tools/make_pinned_plans.py [out.json]
"""
import contextlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else bench.PINNED_PLANS
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    from sqd import nnkernels
    nnkernels.TUNE_SPACE["rounds"] = int(os.environ.get("SQD_TUNE_ROUNDS", "1"))      # measurements per candidate plan (3 rounds measured no better set: tools/ab_plans.sh)
    opts = MonodepthOptions().parse(bench.CONFIG_B)
    with contextlib.redirect_stdout(sys.stderr):
        tr = Trainer(opts)
    tr.set_train()
    batches = [synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, start=i * opts.batch_size, device=tr.device)
               for i in range(bench.NBATCH)]
    for i in range(8):
        tr.train_step(dict(batches[i % len(batches)]))
    torch.cuda.synchronize()
    rec = nnkernels.export_plans()
    rec["conv_source_hash"] = bench.conv_source_hash()
    rec["config"] = "configs[1]: " + " ".join(bench.CONFIG_B)
    rec["measured_on"] = "%s, %s" % (torch.cuda.get_device_name(0), time.strftime("%Y-%m-%d"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(rec, f, indent=0)
    print("wrote %s: %d plans, mix %s" % (out, len(rec["plans"]), nnkernels.plan_mix()))


if __name__ == "__main__":
    main()
