#!/usr/bin/env python3
"""Timing of the stem's max-pool backward gather at configs[1]'s shape (12 x 64 x 96 x 320), with and without the skip gradient and the BatchNorm
sums it takes on the way (sqd_maxpool3x3s2_bwd / _bwd_bn): HIP events over back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sfmnext-impl_amd"))
import torch
from sqd import lib as _l
from sqd.ops import _ptr, _stream

N, C, H, W = 12, 64, 96, 320
L = _l.lib()
dev = torch.device("cuda")
x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
y = torch.empty(N, C, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
idx = torch.empty(N * Ho * Wo * C, device=dev, dtype=torch.uint8)
_l.check(L.sqd_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _ptr(idx), N, H, W, C, _stream()), "fwd")
dy = torch.randn_like(y)
add = torch.randn_like(x)
dx = torch.empty_like(x)
xb = torch.randn_like(x)
mask = torch.randint(0, 16, (N * H * W * C // 4,), device=dev, dtype=torch.uint8)
mean, rstd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
rows = L.sqd_maxpool3x3s2_bwd_bn_rows(N, H, W, C)
part = torch.empty(rows * C * 2, device=dev)


def t(fn, n=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


MB = 1e-6 * 4 * x.numel()
for r in range(3):
    a = t(lambda: L.sqd_maxpool3x3s2_bwd(_ptr(dy), _ptr(idx), None, _ptr(dx), N, H, W, C, _stream()))
    b = t(lambda: L.sqd_maxpool3x3s2_bwd(_ptr(dy), _ptr(idx), _ptr(add), _ptr(dx), N, H, W, C, _stream()))
    c = t(lambda: L.sqd_maxpool3x3s2_bwd_bn(_ptr(dy), _ptr(idx), _ptr(add), _ptr(dx), N, H, W, C, _ptr(xb), _ptr(mask), _ptr(mean), _ptr(rstd), 1, _ptr(part), _stream()))
    print("gather %.1f us (%.0f MB)   + skip gradient %.1f us (%.0f MB)   + BatchNorm sums %.1f us (%.0f MB)" % (a, 1.3125 * MB, b, 2.3125 * MB, c, 3.375 * MB))
