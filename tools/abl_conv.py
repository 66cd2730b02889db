"""Ablation timing of one convolution plan against a given build of the library (dev tool):
python tools/abl_conv.py <libsqd.so> — times a few (layer, mode, plan) cases, us per launch."""
import ctypes
import sys
import torch

L = ctypes.CDLL(sys.argv[1])
P = lambda t: ctypes.c_void_p(t.data_ptr())
HALO = [  # the input-patch kernel (bk 32 + 1024 + 2048)
    ("l1 3x3 64 fwd patch 128x64", (12, 48, 160, 64, 64, 3, 1, 1), 0, (128, 64, 1, 3104)),
    ("l1 3x3 64 fwd patch 64x64", (12, 48, 160, 64, 64, 3, 1, 1), 0, (64, 64, 1, 3104)),
    ("l2 3x3 128 fwd patch 64x64", (12, 24, 80, 128, 128, 3, 1, 1), 0, (64, 64, 1, 3104)),
    ("l2 3x3 128 dgrad patch 64x64", (12, 24, 80, 128, 128, 3, 1, 1), 1, (64, 64, 1, 3104)),
    ("l3 3x3 256 fwd patch 64x64", (12, 12, 40, 256, 256, 3, 1, 1), 0, (64, 64, 1, 3104)),
    ("up2a 640->64 fwd patch 128x64 z4", (12, 24, 80, 640, 64, 3, 1, 1), 0, (128, 64, 4, 3104)),
    ("up3a 320->32 fwd patch 128x32", (12, 48, 160, 320, 32, 3, 1, 1), 0, (128, 32, 1, 3104)),
    ("qtr 32 fwd patch 128x32", (12, 96, 320, 32, 32, 3, 1, 1), 0, (128, 32, 1, 3104)),
]
ONE = [  # 1x1 layers with few pixels and many channels
    ("l3 1x1 256->1024 fwd 64x128", (12, 12, 40, 256, 1024, 1, 1, 0), 0, (64, 128, 1, 1056)),
    ("l3 1x1 256->1024 fwd 64x64", (12, 12, 40, 256, 1024, 1, 1, 0), 0, (64, 64, 1, 1056)),
    ("l3 1x1 256->1024 fwd 128x128", (12, 12, 40, 256, 1024, 1, 1, 0), 0, (128, 128, 1, 1056)),
    ("l3 1x1 1024->256 fwd 64x64", (12, 12, 40, 1024, 256, 1, 1, 0), 0, (64, 64, 1, 1056)),
    ("l3 1x1 1024->256 fwd 64x64 z2", (12, 12, 40, 1024, 256, 1, 1, 0), 0, (64, 64, 2, 1056)),
    ("l4 1x1 512->2048 fwd 64x128", (12, 6, 20, 512, 2048, 1, 1, 0), 0, (64, 128, 1, 1056)),
    ("l4 1x1 512->2048 fwd 64x64", (12, 6, 20, 512, 2048, 1, 1, 0), 0, (64, 64, 1, 1056)),
    ("l3 1x1 256->1024 dgrad 64x64", (12, 12, 40, 256, 1024, 1, 1, 0), 1, (64, 64, 1, 1056)),
    ("l1 1x1 64->256 fwd 64x128", (12, 48, 160, 64, 256, 1, 1, 0), 0, (64, 128, 1, 1056)),
    ("l2 1x1 128->512 fwd 64x128", (12, 24, 80, 128, 512, 1, 1, 0), 0, (64, 128, 1, 1056)),
]
CASES = [  # name, (N,H,W,C,K,R,stride,pad), mode, (bm,bn,z,bk)
    ("l1 3x3 64 fwd 128x64", (12, 48, 160, 64, 64, 3, 1, 1), 0, (128, 64, 1, 1056)),
    ("l2 3x3 128 fwd 64x128 z2", (12, 24, 80, 128, 128, 3, 1, 1), 0, (64, 128, 2, 1056)),
    ("l2 3x3 128 fwd 128x128 z2", (12, 24, 80, 128, 128, 3, 1, 1), 0, (128, 128, 2, 1056)),
    ("l1 1x1 64->256 fwd 64x128", (12, 48, 160, 64, 256, 1, 1, 0), 0, (64, 128, 1, 1056)),
    ("l1 1x1 64->256 fwd 128x128", (12, 48, 160, 64, 256, 1, 1, 0), 0, (128, 128, 1, 1056)),
    ("up1a 1280->128 fwd 64x128 z8", (12, 12, 40, 1280, 128, 3, 1, 1), 0, (64, 128, 8, 1056)),
    ("l2 3x3 128 dgrad 128x64 z2", (12, 24, 80, 128, 128, 3, 1, 1), 1, (128, 64, 2, 1056)),
    ("l2 3x3 128 fwd f32 64x64 bk32 single", (12, 24, 80, 128, 128, 3, 1, 1), 0, (64, 64, 1, 544)),
]
for name, (N, H, W, C, K, R, st, pad), mode, plan in (HALO if len(sys.argv) > 2 and sys.argv[2] == "patch" else ONE if len(sys.argv) > 2 and sys.argv[2] == "1x1" else CASES):
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    geom = (N, H, W, C, K, R, R, st, pad, Ho, Wo)
    x = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(K, R, R, C, device="cuda") * 0.05
    y = torch.randn(N, Ho, Wo, K, device="cuda")
    dx = torch.empty_like(x)
    if L.sqd_conv_set_plan(mode, *geom, *plan) != 0:
        print(name, "plan refused")
        continue
    wsf = ctypes.c_int64(0)
    L.sqd_conv_plan(mode, *geom, ctypes.byref(wsf))
    ws = torch.empty(max(wsf.value, 1), device="cuda")
    st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if mode == 0:
        run = lambda: L.sqd_conv_fwd(P(x), P(w), None, P(y), P(ws), None, *geom, 0, st_)
    else:
        run = lambda: L.sqd_conv_dgrad(P(y), P(w), None, P(dx), P(ws), *geom, st_)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    gf = 2.0 * N * Ho * Wo * K * C * R * R / 1e9
    t = e0.elapsed_time(e1) / 20 * 1e3
    print("%-40s %7.1f us  %6.1f TFLOP/s" % (name, t, gf / t * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / (t * 1e-6) / 1e3), flush=True)
