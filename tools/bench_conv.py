"""Per-layer timing of the native implicit-GEMM convolution vs ATen/MIOpen on the config-B layer shapes
(N=12, 192x640 input).  usage: python tools/bench_conv.py [--iters 20]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from sqd import nnkernels  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--N", type=int, default=12)
ap.add_argument("--only", default="", help="comma-separated substrings of layer names")
ap.add_argument("--tune", action="store_true", help="time the registered plans per layer first (what the Trainer's first step does)")
ap.add_argument("--split3", action="store_true", help="also time the three-term bf16 plans (bk + 1024) per layer: best tile / split-K")
ap.add_argument("--halo", action="store_true", help="like --split3 for the input-patch plans (bk + 1024 + 2048; 3x3 / stride 1 / pad 1 layers only)")
args = ap.parse_args()
if args.halo:
    args.split3 = True
PLAN_FLAG = 32 + 1024 + (2048 if args.halo else 0)
N = args.N
L = [  # name, C, H, W, K, R, stride, pad, count (occurrences per forward)
    ("l1 1x1 64->64", 64, 48, 160, 64, 1, 1, 0, 1), ("l1 3x3 64->64", 64, 48, 160, 64, 3, 1, 1, 3),
    ("l1 1x1 64->256", 64, 48, 160, 256, 1, 1, 0, 4), ("l1 1x1 256->64", 256, 48, 160, 64, 1, 1, 0, 2),
    ("l2 1x1 256->128", 256, 48, 160, 128, 1, 1, 0, 1), ("l2 3x3s2 128", 128, 48, 160, 128, 3, 2, 1, 1),
    ("l2 3x3 128", 128, 24, 80, 128, 3, 1, 1, 3), ("l2 1x1 128->512", 128, 24, 80, 512, 1, 1, 0, 4),
    ("l2 1x1 512->128", 512, 24, 80, 128, 1, 1, 0, 3), ("l2 ds 256->512 s2", 256, 48, 160, 512, 1, 2, 0, 1),
    ("l3 1x1 512->256", 512, 24, 80, 256, 1, 1, 0, 1), ("l3 3x3s2 256", 256, 24, 80, 256, 3, 2, 1, 1),
    ("l3 3x3 256", 256, 12, 40, 256, 3, 1, 1, 5), ("l3 1x1 256->1024", 256, 12, 40, 1024, 1, 1, 0, 6),
    ("l3 1x1 1024->256", 1024, 12, 40, 256, 1, 1, 0, 5), ("l3 ds 512->1024 s2", 512, 24, 80, 1024, 1, 2, 0, 1),
    ("l4 1x1 1024->512", 1024, 12, 40, 512, 1, 1, 0, 1), ("l4 3x3s2 512", 512, 12, 40, 512, 3, 2, 1, 1),
    ("l4 3x3 512", 512, 6, 20, 512, 3, 1, 1, 2), ("l4 1x1 512->2048", 512, 6, 20, 2048, 1, 1, 0, 3),
    ("l4 1x1 2048->512", 2048, 6, 20, 512, 1, 1, 0, 2), ("l4 ds 1024->2048 s2", 1024, 12, 40, 2048, 1, 2, 0, 1),
    ("dec conv2 1x1p1", 2048, 6, 20, 256, 1, 1, 1, 1), ("dec up1a 1280->128", 1280, 12, 40, 128, 3, 1, 1, 1),
    ("dec up1b 128", 128, 12, 40, 128, 3, 1, 1, 1), ("dec up2a 640->64", 640, 24, 80, 64, 3, 1, 1, 1),
    ("dec up2b 64", 64, 24, 80, 64, 3, 1, 1, 1), ("dec up3a 320->32", 320, 48, 160, 32, 3, 1, 1, 1),
    ("dec up3b 32", 32, 48, 160, 32, 3, 1, 1, 1), ("dec up4a 96->16", 96, 96, 320, 16, 3, 1, 1, 1),
    ("dec up4b 16", 16, 96, 320, 16, 3, 1, 1, 1), ("dec conv3 16->32", 16, 96, 320, 32, 3, 1, 1, 1),
    ("qtr conv3x3 32", 32, 96, 320, 32, 3, 1, 1, 1), ("qtr patch16 32", 32, 96, 320, 32, 16, 16, 0, 1),
    ("pose 5x5s2 16->32", 16, 96, 320, 32, 5, 2, 2, 2), ("pose 3x3s2 32->64", 32, 48, 160, 64, 3, 2, 1, 2),
    ("pose 3x3s2 64->128", 64, 24, 80, 128, 3, 2, 1, 2), ("pose 3x3s2 128->256", 128, 12, 40, 256, 3, 2, 1, 2),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


import ctypes
from sqd import lib as _l
LIB = _l.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = {"n_f": 0, "n_d": 0, "n_w": 0, "a_f": 0, "a_d": 0, "a_w": 0}
print("%-22s %7s | %7s %7s | %7s %7s | %7s %7s   (us per launch, C-ABI called back to back on preallocated buffers)" %
      ("layer", "GFLOP", "nat fwd", "at fwd", "nat dg", "at dg", "nat wg", "at wg"))
for name, C, H, W, K, R, st, pad, cnt in L:
    if args.only and not any(k in name for k in args.only.split(",")):
        continue
    if args.halo and not (R == 3 and st == 1 and pad == 1 and C % 4 == 0 and K % 4 == 0):
        continue
    conv = nn.Conv2d(C, K, R, st, pad, bias=False).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = conv.weight.detach()
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    gflop = 2.0 * N * Ho * Wo * K * C * R * R / 1e9
    geom = (N, H, W, C, K, R, R, st, pad, Ho, Wo)
    y = torch.empty(N, K, Ho, Wo, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(y)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    if args.tune:
        nnkernels._tune_conv(0, geom, lambda ws: LIB.sqd_conv_fwd(P(x), P(w), None, P(y), P(ws), None, *geom, 0, ST()))
        nnkernels._tune_conv(1, geom, lambda ws: LIB.sqd_conv_dgrad(P(dy), P(w), None, P(dx), P(ws), *geom, ST()))
        nnkernels._tune_wgrad(geom, False, lambda part: LIB.sqd_conv_wgrad(P(dy), P(x), P(dw), None, P(part), *geom, ST()))
    ws0, ws1 = nnkernels._conv_ws(0, geom, x.device), nnkernels._conv_ws(1, geom, x.device)
    sp, pf = ctypes.c_int(0), ctypes.c_int64(0)
    LIB.sqd_conv_wgrad_plan(N, Ho, Wo, C, K, R, R, ctypes.byref(sp), ctypes.byref(pf))
    part = torch.empty(pf.value, device="cuda")
    t_nf = timeit(lambda: LIB.sqd_conv_fwd(P(x), P(w), None, P(y), P(ws0), None, *geom, 0, ST()), args.iters)
    t_nd = timeit(lambda: LIB.sqd_conv_dgrad(P(dy), P(w), None, P(dx), P(ws1), *geom, ST()), args.iters)
    t_nw = timeit(lambda: LIB.sqd_conv_wgrad(P(dy), P(x), P(dw), None, P(part), *geom, ST()), args.iters)
    if args.split3:
        res = []
        for mode, run in ((0, lambda ws: LIB.sqd_conv_fwd(P(x), P(w), None, P(y), P(ws), None, *geom, 0, ST())),
                          (1, lambda ws: LIB.sqd_conv_dgrad(P(dy), P(w), None, P(dx), P(ws), *geom, ST()))):
            best = None
            for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64), (128, 32), (64, 32)):
                for z in (1, 2, 3, 4, 6, 8):
                    if LIB.sqd_conv_set_plan(mode, *geom, bm, bn, z, PLAN_FLAG) != 0:
                        continue
                    nnkernels._PLAN_CACHE.pop((mode,) + tuple(geom), None)
                    ws = nnkernels._conv_ws(mode, geom, x.device)
                    t = timeit(lambda: run(ws), 10)
                    if best is None or t < best[0]:
                        best = (t, bm, bn, z)
            res.append(best or (float("inf"), 0, 0, 0))
            LIB.sqd_conv_set_plan(mode, *geom, 0, 0, 0, 16)
            nnkernels._PLAN_CACHE.pop((mode,) + tuple(geom), None)
        print("%-22s %7.2f | fwd f32 %7.1f  3xbf16 %7.1f (%dx%d z%d) | dgrad f32 %7.1f  3xbf16 %7.1f (%dx%d z%d)" %
              (name, gflop, t_nf, res[0][0], res[0][1], res[0][2], res[0][3], t_nd, res[1][0], res[1][1], res[1][2], res[1][3]), flush=True)
        tot["n_f"] += cnt * t_nf; tot["n_d"] += cnt * t_nd; tot["a_f"] += cnt * min(t_nf, res[0][0]); tot["a_d"] += cnt * min(t_nd, res[1][0])
        continue
    cb = torch.ops.aten.convolution_backward
    a = (None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1)
    t_af = timeit(lambda: conv(x), args.iters)
    t_ad = timeit(lambda: cb(dy, x, w, *a, [True, False, False]), args.iters)
    t_aw = timeit(lambda: cb(dy, x, w, *a, [False, True, False]), args.iters)
    print("%-22s %7.2f | %7.1f %7.1f | %7.1f %7.1f | %7.1f %7.1f" % (name, gflop, t_nf, t_af, t_nd, t_ad, t_nw, t_aw), flush=True)
    for k, v in (("n_f", t_nf), ("n_d", t_nd), ("n_w", t_nw), ("a_f", t_af), ("a_d", t_ad), ("a_w", t_aw)):
        tot[k] += cnt * v
print("weighted totals per step (us): native fwd %.0f dgrad %.0f wgrad %.0f | aten fwd %.0f dgrad %.0f wgrad %.0f" %
      (tot["n_f"], tot["n_d"], tot["n_w"], tot["a_f"], tot["a_d"], tot["a_w"]))
