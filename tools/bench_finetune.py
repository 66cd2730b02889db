"""Times the supervised finetune step (sfmnext-impl_amd/finetune/train_ft_SQLdepth.py) on one MI355X: the reference's conf/cvnXt.txt
model (ConvNeXt-L U-Net, model_dim 32, patch 32, Q 64, dim_out 64, decoder 1024-512-256-128) on finetune/txt_args/train/inc_kitti.txt's
320x1024 crops with sparse ground truth of the same size; `--bs` is the per-GPU batch.  Prints one JSON line (not bench.py's headline:
a measurement of SURVEY.md §8 row f4).

    python tools/bench_finetune.py --bs 4 --steps 20 --warmup 3"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sfmnext-impl_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=1024)
    a = ap.parse_args()
    from finetune.train_ft_SQLdepth import FinetuneArgs, FinetuneTrainer, synthetic_batch
    from options import MonodepthOptions
    opt = MonodepthOptions().parse(["--backbone", "convnext_large", "--model_dim", "32", "--patch_size", "32", "--query_nums", "64",
                                    "--dim_out", "64", "--dec_channels", "1024", "512", "256", "128", "--height", str(a.height),
                                    "--width", str(a.width), "--min_depth", "0.001", "--max_depth", "80.0", "--sqd_synthetic"])
    fa = FinetuneArgs(bs=a.bs, lr=1e-5, wd=0.01, epochs=5, div_factor=10, final_div_factor=100, same_lr=True)
    tr = FinetuneTrainer(opt, fa, steps_per_epoch=max(a.steps + a.warmup, 8))
    batch = {k: v.cuda() for k, v in synthetic_batch(a.bs, a.height, a.width).items()}
    for _ in range(a.warmup):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, _ = tr.train_step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    from sqd import nnops
    print(json.dumps({"metric": "finetune images/sec, ConvNeXt-L U-Net %dx%d" % (a.width, a.height), "value": round(a.bs / ms * 1e3, 2),
                      "unit": "images/s", "ms_per_step": round(ms, 2), "batch": a.bs, "dtype": "f32", "data": "synthetic",
                      "final_loss": round(float(loss), 5), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                      "operator_backends": nnops.backend_report()}))


if __name__ == "__main__":
    main()
