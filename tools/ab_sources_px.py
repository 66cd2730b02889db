# round 6: the lean forward on pixel-interleaved sources (sqd_photo_args::sources_px) against the planar layout — same box, same inputs:
# every output compared bit for bit, then the launch timed both ways (and the packing pass that produces the copies).
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sfmnext-impl_amd"))
import ctypes
import torch
from sqd import lib as _l, ops
B, H, W = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (12, 192, 640))]
dev = torch.device("cuda"); torch.manual_seed(0); L = _l.lib()
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1).contiguous()
inv_K = torch.linalg.pinv(K).contiguous()
tgt = torch.rand(B, 3, H, W, device=dev); srcs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
depth, part = ops.depth_up_fwd(disp, H, W)
aa, tr = 0.01 * torch.randn(B, 2, 3, device=dev), 0.5 * torch.randn(B, 2, 3, device=dev)
mid, T, P = ops.pose_mats_fwd(aa, tr, [1, 0], K, part, H * W)
noise = torch.randn(B, 2, H, W, device=dev)
ident = ops.identity_fwd(tgt, srcs, noise, 0)
px = ops.pack_pixels(srcs)
print("pack == channels_last copy:", all(torch.equal(p, s) and p.is_contiguous(memory_format=torch.channels_last) for p, s in zip(px, srcs)))
o0 = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident)
o1 = ops.photo_fwd(depth, inv_K, P, tgt, px, ident)
i1 = ops.identity_fwd(tgt, px, noise, 0)
g0 = ops.photo_bwd(depth, inv_K, P, tgt, srcs, o0["sample"], o0["warped"], o0["idx"], 1.0 / (B * H * W))
g1 = ops.photo_bwd(depth, inv_K, P, tgt, px, o0["sample"], o0["warped"], o0["idx"], 1.0 / (B * H * W))
print("identity equal:", torch.equal(ident, i1), " backward equal:", [torch.equal(a, b) for a, b in zip(g0, g1)])
eq = {k: (all(torch.equal(a, b) for a, b in zip(o0[k], o1[k])) if isinstance(o0[k], list) else torch.equal(o0[k], o1[k])) for k in o0 if o0[k] is not None}
print("outputs equal:", eq)
def t(fn, n=200):
    for _ in range(30): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
c0, _k0 = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, prepared_only=True)
c1, _k1 = ops.photo_fwd(depth, inv_K, P, tgt, px, ident, prepared_only=True)
def with_variant(v, fn):
    _l.check(L.sqd_photo_set_fwd_variant(v), "variant")
    try:
        return fn()
    finally:
        _l.check(L.sqd_photo_set_fwd_variant(0), "variant")
o6 = with_variant(6, lambda: ops.photo_fwd(depth, inv_K, P, tgt, px, ident))
print("late rows behind the barrier (variant 6), same bits as the default:", all((all(torch.equal(a, b) for a, b in zip(o6[k], o1[k])) if isinstance(o1[k], list) else torch.equal(o6[k], o1[k])) for k in o1 if o1[k] is not None and k != "loss_part"),
      " loss:", float(o6["loss_part"].double().sum()), float(o1["loss_part"].double().sum()))
for r in range(3):
    print("forward, late rows behind the barrier (variant 6): planar %.1f us, pixel-interleaved %.1f us" % (with_variant(6, lambda: t(lambda: ops.photo_fwd_relaunch(c0))), with_variant(6, lambda: t(lambda: ops.photo_fwd_relaunch(c1)))))
o40 = with_variant(0x40, lambda: ops.photo_fwd(depth, inv_K, P, tgt, px, ident))
print("two rows in flight (0x40) on pixel-interleaved sources, same bits:", all((all(torch.equal(a, b) for a, b in zip(o40[k], o1[k])) if isinstance(o1[k], list) else torch.equal(o40[k], o1[k])) for k in o1 if o1[k] is not None))
for r in range(3):
    print("forward, two rows of a wave in flight (variant 0x40): planar %.1f us, pixel-interleaved %.1f us" % (with_variant(0x40, lambda: t(lambda: ops.photo_fwd_relaunch(c0))), with_variant(0x40, lambda: t(lambda: ops.photo_fwd_relaunch(c1)))))
for r in range(4):
    print("forward: planar %.1f us, pixel-interleaved %.1f us   identity: %.1f / %.1f us   coef + backward (+reduce, alloc): %.1f / %.1f us   pack (2 frames) %.1f us" % (
        t(lambda: ops.photo_fwd_relaunch(c0)), t(lambda: ops.photo_fwd_relaunch(c1)), t(lambda: ops.identity_fwd(tgt, srcs, noise, 0), 50), t(lambda: ops.identity_fwd(tgt, px, noise, 0), 50),
        t(lambda: ops.photo_bwd(depth, inv_K, P, tgt, srcs, o0["sample"], o0["warped"], o0["idx"], 1.0 / (B * H * W)), 50),
        t(lambda: ops.photo_bwd(depth, inv_K, P, tgt, px, o0["sample"], o0["warped"], o0["idx"], 1.0 / (B * H * W)), 50), t(lambda: ops.pack_pixels(srcs, px))))
