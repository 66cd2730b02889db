"""Unit timeline of the stream forward (photo_fwd_s_kernel) from a -DSQD_PHOTO_TRACE build: python tools/trace_stream.py --lib tools/bin/libsqd_trace.so"""
import argparse, ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import numpy as np
import torch
from sqd import lib as _l, ops
ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--tiles", default="0,1,500")
args = ap.parse_args()
_l.SO_PATH = os.path.abspath(args.lib); _l.needs_build = lambda: False
B, H, W, S = 12, 192, 640, 2
dev = torch.device("cuda"); torch.manual_seed(0)
L = _l.lib(); raw = ctypes.CDLL(_l.SO_PATH)
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1).contiguous()
inv_K = torch.linalg.pinv(K).contiguous()
tgt = torch.rand(B, 3, H, W, device=dev); srcs = [torch.rand(B, 3, H, W, device=dev) for _ in range(S)]
disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
depth, part = ops.depth_up_fwd(disp, H, W)
aa, tr = 0.01 * torch.randn(B, S, 3, device=dev), 0.5 * torch.randn(B, S, 3, device=dev)
mid, T, P = ops.pose_mats_fwd(aa, tr, [1, 0], K, part, H * W)
ident = ops.identity_fwd(tgt, srcs, torch.randn(B, S, H, W, device=dev), 0)
_l.check(L.sqd_photo_set_fwd_variant(0), "variant")
call, keep = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, prepared_only=True)
ntile = L.sqd_photo_ntasks(B, H, W, 0) // 16
buf = torch.zeros(ntile * 512, dtype=torch.int64, device=dev)
for _ in range(20):
    ops.photo_fwd_relaunch(call)
torch.cuda.synchronize()
raw.sqd_photo_trace.argtypes = [ctypes.c_void_p]
assert raw.sqd_photo_trace(buf.data_ptr()) == 0
ops.photo_fwd_relaunch(call)
torch.cuda.synchronize()
raw.sqd_photo_trace(None)
t = buf.cpu().numpy().reshape(ntile, 128, 4)
names = {0: "stage", 1: "row", 2: "pair"}
dur = {0: [], 1: [], 2: []}
spans = []
for ti in range(ntile):
    n = int(t[ti, 0, 0]); rec = t[ti, 1:min(n, 127) + 1]
    if not len(rec): continue
    for r in rec: dur[int(r[0]) >> 8].append(int(r[3] - r[2]))
    spans.append(int(rec[:, 3].max() - rec[:, 2].min()))
for k, v in dur.items():
    v = np.array(v); print("%-5s units %6d  cycles min %6d median %6d mean %7.0f p90 %6d max %6d" % (names[k], len(v), v.min(), np.median(v), v.mean(), np.percentile(v, 90), v.max()))
spans = np.array(spans); print("tile span (first unit start -> last unit end): min %d median %d max %d" % (spans.min(), np.median(spans), spans.max()))
for ti in [int(x) for x in args.tiles.split(",")]:
    n = int(t[ti, 0, 0]); rec = t[ti, 1:min(n, 127) + 1]; t0 = rec[:, 2].min()
    print("tile %d: %d units" % (ti, n))
    for r in sorted(rec.tolist(), key=lambda r: r[2]):
        print("   wave %d  %-5s %2d  [%6d .. %6d]  %5d" % (r[1], names[r[0] >> 8], r[0] & 255, r[2] - t0, r[3] - t0, r[3] - r[2]))
