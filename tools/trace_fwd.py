"""Phase timeline of the fused warp+SSIM forward from per-wave s_memtime stamps (a libsqd.so built with -DSQD_PHOTO_TRACE:
tools/build_variant.sh trace photo_tile.hip -DSQD_PHOTO_TRACE).  usage: python tools/trace_fwd.py --lib tools/bin/libsqd_trace.so [--variant 0]"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sqd import lib as _l, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--B", type=int, default=12)
ap.add_argument("--H", type=int, default=192)
ap.add_argument("--W", type=int, default=640)
ap.add_argument("--layout", default="hwc", choices=("hwc", "planar"))
args = ap.parse_args()
_l.SO_PATH = os.path.abspath(args.lib)
_l.needs_build = lambda: False
B, H, W, S = args.B, args.H, args.W, 2
dev = torch.device("cuda")
torch.manual_seed(0)
L = _l.lib()
raw = ctypes.CDLL(_l.SO_PATH)
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1).contiguous()
inv_K = torch.linalg.pinv(K).contiguous()
tgt = torch.rand(B, 3, H, W, device=dev)
srcs = [torch.rand(B, 3, H, W, device=dev) for _ in range(S)]
disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
depth, part = ops.depth_up_fwd(disp, H, W)
aa, tr = 0.01 * torch.randn(B, S, 3, device=dev), 0.5 * torch.randn(B, S, 3, device=dev)
mid, T, P = ops.pose_mats_fwd(aa, tr, [1, 0], K, part, H * W)
noise = torch.randn(B, S, H, W, device=dev)
if args.layout == "hwc":
    srcs = ops.pack_pixels(srcs)
ident = ops.identity_fwd(tgt, srcs, noise, 0)
_l.check(L.sqd_photo_set_fwd_variant(args.variant), "variant")
call, keep = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, prepared_only=True)
nt = L.sqd_photo_ntasks(B, H, W, 0)          # tiles x waves
buf = torch.zeros(nt * 8, dtype=torch.int64, device=dev)      # (nt = 16 slots per tile >= 8 waves x 8 stamps)
for _ in range(50):
    ops.photo_fwd_relaunch(call)
torch.cuda.synchronize()
raw.sqd_photo_trace.argtypes = [ctypes.c_void_p]
assert raw.sqd_photo_trace(buf.data_ptr()) == 0
ops.photo_fwd_relaunch(call)
ops.photo_fwd_relaunch(call)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8, 8).astype(np.int64)[:nt // 16]      # [tile][wave][stamp]
raw.sqd_photo_trace(None)
ntile = t.shape[0]
t0 = t[:, :, 0].min()
st = t[:, :, :6] - t0
hw = t[:, 0, 7]
xcc = t[:, 0, 6] & 0xf
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
MHZ = 100.0          # s_memtime ticks: 100 MHz constant clock on this part if the spans come out ~5000 ticks; printed raw
print("tiles %d, waves per tile %d; all times in s_memtime ticks since the first wave started" % (ntile, t.shape[1]))
span = (t[:, :, 5].max() - t0)
print("kernel span (first wave start -> last wave end): %d ticks" % span)
names = ["start", "staged", "warped", "barrier", "stored", "end"]
for i, n in enumerate(names):
    v = st[:, :, i]
    print("  %-8s min %7d  p10 %7d  median %7d  p90 %7d  max %7d" % (n, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
d = np.diff(st, axis=2)
for i in range(5):
    v = d[:, :, i]
    print("  %-8s->%-8s per wave: min %6d median %6d mean %6.0f max %6d" % (names[i], names[i + 1], v.min(), np.median(v), v.mean(), v.max()))
# tiles per CU and their order
from collections import defaultdict
per = defaultdict(list)
for ti in range(ntile):
    per[int(cuid[ti])].append((int(st[ti, :, 0].min()), int(st[ti, :, 5].max()), ti))
cnt = np.array([len(v) for v in per.values()])
print("CUs seen %d; tiles per CU: min %d max %d; histogram %s" % (len(per), cnt.min(), cnt.max(), dict(zip(*np.unique(cnt, return_counts=True)))))
ends = np.array([max(e for _, e, _ in v) for v in per.values()])
print("per-CU finish time: min %d median %d max %d" % (ends.min(), np.median(ends), ends.max()))
for key in list(per.keys())[:3]:
    base = min(s for s, _, _ in per[key])
    print("  CU %d: " % key + "  ".join("tile %d [%d..%d]" % (ti, s - base, e - base) for s, e, ti in sorted(per[key])))
