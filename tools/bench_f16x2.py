"""Two-term fp16 operands (f16x2, round 5) against the three-term bf16 plans, layer by layer on the config-B shapes (dev tool).
For each layer: the best plan of either arithmetic for forward / data gradient (implicit-GEMM and input-patch tiles, 4- and 8-wave,
split-K 1..8) and for the weight gradient (direct-operand kernel, impl 6 vs impl 7, every register tile and pixel split), timed
through the C ABI on preallocated buffers; with --check the error of fp32-MFMA / bf16x3 / f16x2 against float64 on reduced batches.
usage: python tools/bench_f16x2.py [--only l3,l4] [--check] [--N 12]"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
from sqd import lib as _l  # noqa: E402
from sqd import nnkernels  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=12)
ap.add_argument("--only", default="")
ap.add_argument("--check", action="store_true")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--heavy", action="store_true", help="inputs with a heavy-tailed per-pixel magnitude (exp(3 N(0,1))) instead of N(0,1)")
args = ap.parse_args()
N = args.N
LAYERS = [  # name, C, H, W, K, R, stride, pad, count per forward
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
    ("qtr conv3x3 32", 32, 96, 320, 32, 3, 1, 1, 1),
    ("pose 5x5s2 16->32", 16, 96, 320, 32, 5, 2, 2, 2), ("pose 3x3s2 32->64", 32, 48, 160, 64, 3, 2, 1, 2),
    ("pose 3x3s2 64->128", 64, 24, 80, 128, 3, 2, 1, 2), ("pose 3x3s2 128->256", 128, 12, 40, 256, 3, 2, 1, 2),
]
LIB = _l.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def amax_of(t):
    a = torch.zeros(nnkernels.AMAX_REC, device="cuda")
    _l.check(LIB.sqd_amax(P(t), t.numel(), P(a), ST()), "amax")
    return a


def best_gemm(mode, geom, run, flags, tiles=((128, 128), (128, 64), (64, 128), (64, 64), (128, 32), (64, 32))):
    best = None
    for fl in flags:
        for bm, bn in tiles:
            for z in (1, 2, 3, 4, 6, 8):
                if LIB.sqd_conv_set_plan(mode, *geom, bm, bn, z, fl) != 0:
                    continue
                nnkernels._PLAN_CACHE.pop((mode,) + tuple(geom), None)
                ws = nnkernels._conv_ws(mode, geom, torch.device("cuda"))
                if run(ws) != 0:
                    raise RuntimeError(LIB.sqd_last_error().decode())
                t = timeit(lambda: run(ws), args.iters)
                if best is None or t < best[0]:
                    best = (t, bm, bn, z, fl)
    LIB.sqd_conv_set_plan(mode, *geom, 0, 0, 0, 16)
    nnkernels._PLAN_CACHE.pop((mode,) + tuple(geom), None)
    return best or (float("inf"), 0, 0, 0, 0)


def best_wgrad(geom, run, impl):
    Nn, H, W, C, K, R, S, st, pad, Ho, Wo = geom
    best = None
    for v in range(8):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128):
            if LIB.sqd_conv_wgrad_set_plan(Nn, Ho, Wo, C, K, R, S, impl | (v << 4), sp) != 0:
                continue
            spl, pf = ctypes.c_int(0), ctypes.c_int64(0)
            LIB.sqd_conv_wgrad_plan(Nn, Ho, Wo, C, K, R, S, ctypes.byref(spl), ctypes.byref(pf))
            part = torch.empty(max(pf.value, 1), device="cuda")
            if run(part) != 0:
                raise RuntimeError(LIB.sqd_last_error().decode())
            t = timeit(lambda: run(part), args.iters)
            if best is None or t < best[0]:
                best = (t, v, spl.value)
    LIB.sqd_conv_wgrad_set_plan(Nn, Ho, Wo, C, K, R, S, -1, 0)
    return best or (float("inf"), 0, 0)


def rel_err(a, ref):
    d = (a.double().cpu() - ref).abs()
    return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


tot = {k: 0.0 for k in ("f3", "f2", "d3", "d2", "w3", "w2")}
print("%-22s %6s | fwd bf16x3 -> f16x2 | dgrad bf16x3 -> f16x2 | wgrad bf16x3 -> f16x2   (us per launch, best plan of each arithmetic)" % ("layer", "GFLOP"))
for name, C, H, W, K, R, st, pad, cnt in LAYERS:
    if args.only and not any(k in name for k in args.only.split(",")):
        continue
    torch.manual_seed(1)
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    geom = (N, H, W, C, K, R, R, st, pad, Ho, Wo)
    x = torch.randn(N, H, W, C, device="cuda")
    dy = torch.randn(N, Ho, Wo, K, device="cuda")
    if args.heavy:
        x = x * torch.exp(3 * torch.randn(N, H, W, 1, device="cuda"))
        dy = dy * torch.exp(3 * torch.randn(N, Ho, Wo, 1, device="cuda")) * 1e-7
    w = torch.randn(K, R, R, C, device="cuda") * (2.0 / (C * R * R)) ** 0.5
    y, dx, dw = torch.empty(N, Ho, Wo, K, device="cuda"), torch.empty_like(x), torch.empty_like(w)
    ax, aw, ady = amax_of(x), amax_of(w), amax_of(dy)
    gflop = 2.0 * N * Ho * Wo * K * C * R * R / 1e9
    halo = R == 3 and st == 1 and pad == 1
    f3 = [32 + 1024, 32 + 1024 + 256] + ([32 + 1024 + 2048, 32 + 1024 + 2048 + 256] if halo else [])
    f2 = [f + 4096 for f in f3]
    run_f = lambda ws: LIB.sqd_conv_fwd_scaled(P(x), P(w), None, P(y), P(ws), None, P(ax), P(aw), None, *geom, 0, ST())
    run_d = lambda ws: LIB.sqd_conv_dgrad_scaled(P(dy), P(w), None, P(dx), P(ws), None, None, None, None, 0, None, P(ady), P(aw), None, *geom, ST())
    run_w = lambda part: LIB.sqd_conv_wgrad_scaled(P(dy), P(x), P(dw), None, P(part), P(ady), P(ax), *geom, None, ST())
    bf3, bf2 = best_gemm(0, geom, run_f, f3), best_gemm(0, geom, run_f, f2)
    bd3, bd2 = best_gemm(1, geom, run_d, f3), best_gemm(1, geom, run_d, f2)
    bw3, bw2 = best_wgrad(geom, run_w, 6), best_wgrad(geom, run_w, 7)
    print("%-22s %6.2f | %6.1f (%dx%d z%d %d) -> %6.1f (%dx%d z%d %d) | %6.1f (%dx%d z%d %d) -> %6.1f (%dx%d z%d %d) | %6.1f (v%d s%d) -> %6.1f (v%d s%d)" %
          ((name, gflop) + bf3 + bf2 + bd3 + bd2 + bw3 + bw2), flush=True)
    for k, b in (("f3", bf3), ("f2", bf2), ("d3", bd3), ("d2", bd2), ("w3", bw3), ("w2", bw2)):
        tot[k] += cnt * b[0] if b[0] != float("inf") else 0.0
    if args.check:
        # float64 reference on a reduced batch (CPU), each arithmetic on its best plan
        nb = min(N, 2)
        xs, dys = x[:nb].contiguous(), dy[:nb].contiguous()
        g2 = (nb,) + geom[1:]
        xr = xs.double().cpu().permute(0, 3, 1, 2)
        wr = w.double().cpu().permute(0, 3, 1, 2)
        yr = torch.nn.functional.conv2d(xr, wr, None, st, pad)
        gyr = dys.double().cpu().permute(0, 3, 1, 2)
        dxr = torch.nn.grad.conv2d_input(xr.shape, wr, gyr, st, pad)
        dwr = torch.nn.grad.conv2d_weight(xr, wr.shape, gyr, st, pad)
        ys, dxs = torch.empty(nb, Ho, Wo, K, device="cuda"), torch.empty_like(xs)
        a_x, a_dy = amax_of(xs), amax_of(dys)
        out = []
        for label, fl, wimpl in (("fp32", 16, 1), ("bf16x3", 32 + 1024, 6), ("f16x2", 32 + 1024 + 4096, 7)):
            e = []
            for mode in (0, 1):
                ok = False
                for bm, bn in ((64, 64), (128, 32), (128, 64)):
                    if LIB.sqd_conv_set_plan(mode, *g2, bm, bn, 1, fl) == 0:
                        ok = True
                        break
                nnkernels._PLAN_CACHE.pop((mode,) + tuple(g2), None)
                if not ok:
                    e.append((float("nan"),) * 2)
                    continue
                if mode == 0:
                    _l.check(LIB.sqd_conv_fwd_scaled(P(xs), P(w), None, P(ys), None, None, P(a_x), P(aw), None, *g2, 0, ST()), "fwd")
                    e.append(rel_err(ys.permute(0, 3, 1, 2), yr))
                else:
                    _l.check(LIB.sqd_conv_dgrad_scaled(P(dys), P(w), None, P(dxs), None, None, None, None, None, 0, None, P(a_dy), P(aw), None, *g2, ST()), "dgrad")
                    e.append(rel_err(dxs.permute(0, 3, 1, 2), dxr))
                LIB.sqd_conv_set_plan(mode, *g2, 0, 0, 0, 16)
            if wimpl == 1 or LIB.sqd_conv_wgrad_set_plan(nb, Ho, Wo, C, K, R, R, wimpl | (1 << 4) if C % 64 or K % 64 else wimpl, 2) == 0:
                if wimpl == 1:
                    LIB.sqd_conv_wgrad_set_plan(nb, Ho, Wo, C, K, R, R, 1, 2)
                spl, pf = ctypes.c_int(0), ctypes.c_int64(0)
                LIB.sqd_conv_wgrad_plan(nb, Ho, Wo, C, K, R, R, ctypes.byref(spl), ctypes.byref(pf))
                part = torch.empty(max(pf.value, 1), device="cuda")
                _l.check(LIB.sqd_conv_wgrad_scaled(P(dys), P(xs), P(dw), None, P(part), P(a_dy), P(a_x), *g2, None, ST()), "wgrad")
                e.append(rel_err(dw.permute(0, 3, 1, 2), dwr))
                LIB.sqd_conv_wgrad_set_plan(nb, Ho, Wo, C, K, R, R, -1, 0)
            else:
                e.append((float("nan"),) * 2)
            out.append("%s fwd %.2e/%.2e dgrad %.2e/%.2e wgrad %.2e/%.2e" % ((label,) + e[0] + e[1] + e[2]))
        print("    max/rms error vs float64: " + " | ".join(out), flush=True)
print("sum over a step's launches (us): forward %.0f -> %.0f, data gradient %.0f -> %.0f, weight gradient %.0f -> %.0f" %
      (tot["f3"], tot["f2"], tot["d3"], tot["d2"], tot["w3"], tot["w2"]))
