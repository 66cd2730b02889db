"""Times every weight-gradient plan the library accepts for one geometry (dev tool):
python tools/bench_wgrad.py N H W C K R [stride pad]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
from sqd import lib as _l  # noqa: E402

if "--lib" in sys.argv:                       # another build of libsqd.so (tools/build_alt_lib.sh) for same-box A/B runs
    i = sys.argv.index("--lib")
    _l.SO_PATH = os.path.abspath(sys.argv[i + 1])
    _l.needs_build = lambda: False
    del sys.argv[i:i + 2]
N, H, W, C, K, R = (int(v) for v in sys.argv[1:7])
st, pad = (int(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (1, (R - 1) // 2)
L = _l.lib()
Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
geom = (N, H, W, C, K, R, R, st, pad, Ho, Wo)
x = torch.randn(N, H, W, C, device="cuda")
dy = torch.randn(N, Ho, Wo, K, device="cuda")
dw = torch.empty(K, R, R, C, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
gflop = 2.0 * N * Ho * Wo * K * C * R * R / 1e9
names = {0: "lds-tiled fp32", 1: "direct fp32", 2: "shared fp32", 3: "shared 3xbf16", 4: "row-window fp32", 6: "direct 3xbf16"}
only = [int(v) for v in os.environ.get('BENCH_WGRAD_IMPLS', '').split(',') if v]
for impl, variants in ((1, [0]), (0, [0]), (2, range(4)), (3, range(4)), (4, [0]), (6, range(8))):
    if only and impl not in only:
        continue
    for v in variants:
        best = None
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128) + ((192, 256, 384, 512, 768, 1024, 1536) if impl == 4 else ()):
            code = impl | (v << 4) if impl >= 2 else impl
            if L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, code, sp) != 0:
                continue
            spl, pf = ctypes.c_int(0), ctypes.c_int64(0)
            L.sqd_conv_wgrad_plan(N, Ho, Wo, C, K, R, R, ctypes.byref(spl), ctypes.byref(pf))
            part = torch.empty(max(pf.value, 1), device="cuda")
            run = lambda: L.sqd_conv_wgrad(P(dy), P(x), P(dw), None, P(part), *geom, S)
            for _ in range(2):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5 * 1e3
            if best is None or t < best[0]:
                best = (t, spl.value)
        if best:
            print("%-16s variant %d: %8.1f us (%5.1f TFLOP/s) at %d splits" % (names[impl], v, best[0], gflop * 1e3 / best[0], best[1]), flush=True)
L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, -1, 0)
