"""Micro-benchmark of the photometric kernels alone (for rocprofv3 --pmc passes and TH sweeps).
usage: python tools/bench_fused.py [--rows 8,12,16,24,32] [--iters 200] [--B 12 --H 192 --W 640] [--which fwd,ident,bwd]"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
from sqd import lib as _l, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="0")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--B", type=int, default=12)
ap.add_argument("--H", type=int, default=192)
ap.add_argument("--W", type=int, default=640)
ap.add_argument("--which", default="fwd,ident,coef,bwd")
ap.add_argument("--cold", action="store_true", help="evict L2 and the Infinity Cache before every timed call (a 1 GB fill): what the kernels see "
                                                     "inside the training step, where their inputs were produced milliseconds earlier")
ap.add_argument("--layout", default="hwc", choices=("hwc", "planar"), help="memory layout of the source frames: hwc = channels_last, what the captured "
                "training step keeps them in (trainer._capture) — the default, so that the kernels profiled here are the step's; planar = [B,3,H,W]")
ap.add_argument("--dump", default=None, help="save the backward's outputs here (bit-compare two builds)")
ap.add_argument("--lib", default=None, help="another build of libsqd.so (tools/build_alt_lib.sh) for same-box A/B runs")
args = ap.parse_args()
if args.lib:
    _l.SO_PATH = os.path.abspath(args.lib)
    _l.needs_build = lambda: False
B, H, W = args.B, args.H, args.W
dev = torch.device("cuda")
torch.manual_seed(0)
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1).contiguous()
inv_K = torch.linalg.pinv(K).contiguous()
tgt = torch.rand(B, 3, H, W, device=dev)
srcs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
if args.layout == "hwc" and ops.sources_hwc_ok(B, 2, H, W):
    srcs = ops.pack_pixels(srcs)
disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
depth, part = ops.depth_up_fwd(disp, H, W)
aa, tr = 0.01 * torch.randn(B, 2, 3, device=dev), 0.5 * torch.randn(B, 2, 3, device=dev)
mid, T, P = ops.pose_mats_fwd(aa, tr, [1, 0], K, part, H * W)
noise = torch.randn(B, 2, H, W, device=dev)


_evict = None


def timeit(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    if args.cold:
        global _evict
        if _evict is None:
            _evict = torch.empty(256 << 20, device=dev, dtype=torch.float32)
        tot, n = 0.0, min(iters, 20)
        for i in range(n):
            _evict.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / n * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


px = B * H * W
# the first measurement of a process reads 5-10 % slow (clocks / caches settle over the first ~100 ms of work): burn that here, or the
# first entry of a sweep is penalised (round 3's first tile-height sweeps were: profiles/r03m_photo_tile_heights.md)
_w = ops.identity_fwd(tgt, srcs, noise, 0)
for _ in range(300):
    ops.identity_fwd(tgt, srcs, noise, 0)
torch.cuda.synchronize()
for rows in [int(r) for r in args.rows.split(",")]:
    ident = ops.identity_fwd(tgt, srcs, noise, rows)
    res = {}
    which = args.which.split(",")
    if "fwd" in which:
        call, keep = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, rows_per_task=rows, prepared_only=True)
        res["fwd"] = timeit(lambda: ops.photo_fwd_relaunch(call), args.iters)
    if "ident" in which:
        res["identity"] = timeit(lambda: ops.identity_fwd(tgt, srcs, noise, rows), 50)
    if "coef" in which or "bwd" in which:
        out = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, rows_per_task=rows)
    if "coef" in which:
        res["coef"] = timeit(lambda: ops.photo_coef(tgt, out["warped"], out["idx"], rows), 50)
    if "bwd" in which:
        res["coef+bwd(+reduce,alloc)"] = timeit(lambda: ops.photo_bwd(depth, inv_K, P, tgt, srcs, out["sample"], out["warped"], out["idx"], 1.0 / px, rows), 50)
    if args.dump and "bwd" in which:
        g = ops.photo_bwd(depth, inv_K, P, tgt, srcs, out["sample"], out["warped"], out["idx"], 1.0 / px, rows)
        torch.save([x.cpu() for x in (g if isinstance(g, (tuple, list)) else [g]) if torch.is_tensor(x)], args.dump)
    print("rows_per_task=%d  " % rows + "  ".join("%s %.1f us (%.0f GB/s @93B/px)" % (k, v, 93 * px / v / 1e3) for k, v in res.items()), flush=True)
