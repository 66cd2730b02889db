"""Error of every forward / data-gradient plan of one geometry against float64, per arithmetic (dev tool, round 5):
python tools/diag_f16x2_err.py N C H W K R stride pad [normal|relu|heavy]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from sqd import lib as _l  # noqa: E402
from sqd import nnkernels  # noqa: E402

N, C, H, W, K, R, st, pad = (int(v) for v in sys.argv[1:9])
dist = sys.argv[9] if len(sys.argv) > 9 else "normal"
L = _l.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)
Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
geom = (N, H, W, C, K, R, R, st, pad, Ho, Wo)
x = torch.randn(N, H, W, C, device="cuda")
dy = torch.randn(N, Ho, Wo, K, device="cuda")
w = torch.randn(K, R, R, C, device="cuda") * (2.0 / (C * R * R)) ** 0.5
if dist == "heavy":
    x = x * torch.exp(3 * torch.randn(N, H, W, 1, device="cuda"))
    dy = dy * torch.exp(3 * torch.randn(N, Ho, Wo, 1, device="cuda")) * 1e-7
elif dist == "relu":
    x = F.relu(x)
xr, wr, gyr = x.double().cpu().permute(0, 3, 1, 2), w.double().cpu().permute(0, 3, 1, 2), dy.double().cpu().permute(0, 3, 1, 2)
yr = F.conv2d(xr, wr, None, st, pad)
dxr = torch.nn.grad.conv2d_input(xr.shape, wr, gyr, st, pad)


def amax_of(t):
    a = torch.zeros(nnkernels.AMAX_REC, device="cuda")
    _l.check(L.sqd_amax(P(t), t.numel(), P(a), ST()), "amax")
    return a


ax, aw, ady = amax_of(x), amax_of(w), amax_of(dy)
print("amax x %g (torch %g)  w %g (%g)  dy %g (%g)" % (nnkernels.amax_value(ax), float(x.abs().max()), nnkernels.amax_value(aw), float(w.abs().max()), nnkernels.amax_value(ady), float(dy.abs().max())))
y, dx = torch.empty(N, Ho, Wo, K, device="cuda"), torch.empty_like(x)


def err(a, ref):
    d = (a.double().cpu().permute(0, 3, 1, 2) - ref).abs()
    return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


for fl, name in ((16, "fp32"), (32 + 1024, "bf16x3"), (32 + 1024 + 4096, "f16x2"), (32 + 1024 + 256, "bf16x3 8w"), (32 + 1024 + 4096 + 256, "f16x2 8w"),
                 (32 + 1024 + 2048, "bf16x3 patch"), (32 + 1024 + 2048 + 4096, "f16x2 patch")):
    for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64), (128, 32)):
        for z in (1, 2):
            line = "%-13s %3dx%-3d z%d:" % (name, bm, bn, z)
            any_ok = False
            for mode in (0, 1):
                if L.sqd_conv_set_plan(mode, *geom, bm, bn, z, fl) != 0:
                    line += "   %s -" % ("fwd" if mode == 0 else "dgrad")
                    continue
                any_ok = True
                nnkernels._PLAN_CACHE.pop((mode,) + tuple(geom), None)
                ws = nnkernels._conv_ws(mode, geom, torch.device("cuda"))
                if mode == 0:
                    y.fill_(float("nan"))
                    _l.check(L.sqd_conv_fwd_scaled(P(x), P(w), None, P(y), P(ws), None, P(ax), P(aw), None, *geom, 0, ST()), "fwd")
                    line += "   fwd max %.2e rms %.2e" % err(y, yr)
                else:
                    dx.fill_(float("nan"))
                    _l.check(L.sqd_conv_dgrad_scaled(P(dy), P(w), None, P(dx), P(ws), None, None, None, None, 0, None, P(ady), P(aw), None, *geom, ST()), "dgrad")
                    line += "   dgrad max %.2e rms %.2e" % err(dx, dxr)
                L.sqd_conv_set_plan(mode, *geom, 0, 0, 0, 16)
                nnkernels._PLAN_CACHE.pop((mode,) + tuple(geom), None)
            if any_ok:
                print(line, flush=True)
