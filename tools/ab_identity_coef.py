import os, sys
sys.path.insert(0, "/root/repo/sfmnext-impl_amd")
import torch
from sqd import lib as _l, ops
B, H, W = 12, 192, 640
dev = torch.device("cuda"); torch.manual_seed(0); L = _l.lib()
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1).contiguous()
inv_K = torch.linalg.pinv(K).contiguous()
tgt = torch.rand(B, 3, H, W, device=dev); srcs = [torch.rand(B, 3, H, W, device=dev) for _ in range(2)]
disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
depth, part = ops.depth_up_fwd(disp, H, W)
aa, tr = 0.01 * torch.randn(B, 2, 3, device=dev), 0.5 * torch.randn(B, 2, 3, device=dev)
mid, T, P = ops.pose_mats_fwd(aa, tr, [1, 0], K, part, H * W)
noise = torch.randn(B, 2, H, W, device=dev)
def t(fn, n=100):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
res = {}
for v in (1, 0):
    _l.check(L.sqd_photo_set_fwd_variant(v), "v")
    ident = ops.identity_fwd(tgt, srcs, noise, 0)
    out = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident)
    coef = ops.photo_coef(tgt, out["warped"], out["idx"], 0)
    res[v] = (ident.clone(), coef.clone())
print("identity equal:", torch.equal(res[0][0], res[1][0]), " coef equal:", torch.equal(res[0][1], res[1][1]), float((res[0][1]-res[1][1]).abs().max()))
for r in range(3):
    line = []
    for v in (1, 0):
        _l.check(L.sqd_photo_set_fwd_variant(v), "v")
        line.append("variant %d: identity %.1f us, coef %.1f us" % (v, t(lambda: ops.identity_fwd(tgt, srcs, noise, 0)), t(lambda: ops.photo_coef(tgt, out["warped"], out["idx"], 0))))
    print("  ".join(line))
