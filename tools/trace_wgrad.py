"""Phase timestamps of conv_wgrad_direct3_kernel (dev tool; needs a library built with -DSQD_WGRAD_TRACE: tools/build_variant.sh trace conv.hip -DSQD_WGRAD_TRACE):
python tools/trace_wgrad.py --lib tools/bin/libsqd_trace.so N H W C K R impl splits
Ablations: add -DSQD_WG_NOCVT (operands packed without the three-term split) and / or -DSQD_WG_NOLOAD (only the first step's loads are issued) to the
variant build: loop length with the MFMAs alone / + conversions / + loads."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
from sqd import lib as _l  # noqa: E402

i = sys.argv.index("--lib")
_l.SO_PATH = os.path.abspath(sys.argv[i + 1])
_l.needs_build = lambda: False
del sys.argv[i:i + 2]
N, H, W, C, K, R, impl, splits = (int(v) for v in sys.argv[1:9])
pad = (R - 1) // 2
L = _l.lib()
geom = (N, H, W, C, K, R, R, 1, pad, H, W)
x = torch.randn(N, H, W, C, device="cuda")
dy = torch.randn(N, H, W, K, device="cuda")
dw = torch.empty(K, R, R, C, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
assert L.sqd_conv_wgrad_set_plan(N, H, W, C, K, R, R, impl, splits) == 0
spl, pf = ctypes.c_int(0), ctypes.c_int64(0)
L.sqd_conv_wgrad_plan(N, H, W, C, K, R, R, ctypes.byref(spl), ctypes.byref(pf))
nwg_max = 8192
part = torch.zeros(pf.value + nwg_max * 4 * 4 * 2, device="cuda")
for _ in range(3):
    L.sqd_conv_wgrad(P(dy), P(x), P(dw), None, P(part), *geom, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.sqd_conv_wgrad(P(dy), P(x), P(dw), None, P(part), *geom, S)
e1.record()
torch.cuda.synchronize()
tr = part[pf.value:].view(torch.int64).view(-1, 4).cpu()
tr = tr[tr[:, 0] != 0]
t0 = int(tr[:, 0].min())
ent, loop, end = (tr[:, 0] - t0).double(), (tr[:, 1] - t0).double(), (tr[:, 2] - t0).double()
q = lambda v: "min %.0f  p10 %.0f  med %.0f  p90 %.0f  max %.0f" % tuple(float(torch.quantile(v, p)) for p in (0.0, 0.1, 0.5, 0.9, 1.0))
print("splits %d, waves traced %d, launch+reduce %.1f us by events (ticks below: s_memtime)" % (spl.value, tr.shape[0], e0.elapsed_time(e1) * 1e3))
print("loop length:", q(loop - ent))
print("epilogue   :", q(end - loop))
