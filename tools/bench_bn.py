"""Micro-benchmark of the BatchNorm kernels on the config-B layer shapes (dev): python tools/bench_bn.py [--lib other/libsqd.so]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
from sqd import lib as _l  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
args = ap.parse_args()
if args.lib:
    _l.SO_PATH = os.path.abspath(args.lib)
    _l.needs_build = lambda: False
from sqd import nnkernels  # noqa: E402
import torch.nn as nn  # noqa: E402


def timeit(fn, iters=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


x0 = torch.randn(1 << 26, device="cuda")
for _ in range(50):
    x0.mul_(1.0)                       # warm the process up (see tools/bench_fused.py)
for N, C, H, W, res in [(12, 64, 96, 320, False), (12, 256, 48, 160, True), (12, 64, 48, 160, False), (12, 512, 24, 80, True), (12, 128, 24, 80, False),
                        (12, 1024, 12, 40, True), (12, 2048, 6, 20, True)]:
    bn = nn.BatchNorm2d(C).cuda()
    x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
    y = nnkernels.batch_norm_act(x, bn, "relu", r)
    g = torch.randn_like(y)
    tf = timeit(lambda: nnkernels.batch_norm_act(x, bn, "relu", r))
    tb = timeit(lambda: torch.autograd.grad(y, [x] + ([r] if res else []), g, retain_graph=True))
    el = N * C * H * W
    bf = el * (4 + 4 + 0.25 + (4 if res else 0)) + el * 4          # forward: stats read, apply read (+res), mask, write
    bb = el * (4 + 4 + 0.25) * 2 + el * 4 * (2 if res else 1)       # backward: reduce reads dy, x, mask; apply reads again, writes dx (+dres)
    print("[%d,%d,%d,%d]%s  fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (N, C, H, W, " +res" if res else "", tf, bf / tf / 1e6, tb, bb / tb / 1e6), flush=True)
