"""Same-box A/B of the fused warp+SSIM forward's kernel variants (sqd_photo_set_fwd_variant): outputs compared bit for bit,
then alternating timed runs.  usage: python tools/ab_fwd_variant.py [--B 12 --H 192 --W 640 --S 2] [--rounds 5] [--iters 200]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
from sqd import lib as _l, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=12)
ap.add_argument("--H", type=int, default=192)
ap.add_argument("--W", type=int, default=640)
ap.add_argument("--S", type=int, default=2)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--variants", default="1,0")
args = ap.parse_args()
B, H, W, S = args.B, args.H, args.W, args.S
dev = torch.device("cuda")
torch.manual_seed(0)
L = _l.lib()
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1).contiguous()
inv_K = torch.linalg.pinv(K).contiguous()
tgt = torch.rand(B, 3, H, W, device=dev)
srcs = [torch.rand(B, 3, H, W, device=dev) for _ in range(S)]
disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
depth, part = ops.depth_up_fwd(disp, H, W)
aa, tr = 0.01 * torch.randn(B, S, 3, device=dev), 0.5 * torch.randn(B, S, 3, device=dev)
mid, T, P = ops.pose_mats_fwd(aa, tr, [1, 0, 0, 0][:S], K, part, H * W)
noise = torch.randn(B, S, H, W, device=dev)
ident = ops.identity_fwd(tgt, srcs, noise, 0)
variants = [int(v) for v in args.variants.split(",")]
outs = {}
for v in variants:
    _l.check(L.sqd_photo_set_fwd_variant(v), "variant")
    outs[v] = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, want_taps=True, want_reproj=True)
    outs[(v, "lean")] = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident)          # (the production call: no tap / reprojection dumps)
torch.cuda.synchronize()
for v in variants:          # the production call of every variant against the first variant's full call
    o, r = outs[(v, "lean")], outs[variants[0]]
    ok = all(torch.equal(o[k], r[k]) for k in ("sel", "idx")) and all(torch.equal(x, y) for k in ("sample", "warped") for x, y in zip(o[k], r[k]))
    print("variant %d production call: sel / idx / sample / warped equal to variant %d's: %s; loss sum %.9g vs %.9g" % (
        v, variants[0], ok, o["loss_part"].double().sum().item(), r["loss_part"].double().sum().item()))
ref = outs[variants[0]]
for v in variants[1:]:
    o = outs[v]
    for key in ("sel", "idx", "loss_part", "reproj"):
        print("variant %d vs %d  %-9s equal: %s" % (v, variants[0], key, torch.equal(o[key], ref[key])))
    for key in ("sample", "warped", "x0y0"):
        print("variant %d vs %d  %-9s equal: %s" % (v, variants[0], key, all(torch.equal(x, y) for x, y in zip(o[key], ref[key]))))


def timeit(v):
    _l.check(L.sqd_photo_set_fwd_variant(v), "variant")
    call, keep = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, prepared_only=True)
    for _ in range(20):
        ops.photo_fwd_relaunch(call)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        ops.photo_fwd_relaunch(call)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


for _ in range(300):
    ops.identity_fwd(tgt, srcs, noise, 0)
torch.cuda.synchronize()
px = B * H * W
for r in range(args.rounds):
    print("round %d  " % r + "  ".join("variant %d: %.1f us (%.3f of 8 TB/s)" % (v, t, 93 * px / t / 1e3 / 8000) for v, t in ((v, timeit(v)) for v in variants)), flush=True)
_l.check(L.sqd_photo_set_fwd_variant(0), "variant")
