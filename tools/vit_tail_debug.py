"""diagnose EncoderTail vs fp64 composite: where do the errors sit, and are they ReLU-kink flips?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sfmnext-impl_amd"))
import torch, torch.nn.functional as F
from sqd import nnkernels
rows, E, Fh = 1440, 32, 1024
g = torch.Generator().manual_seed(rows * 3 + E + Fh)
x, sa = torch.randn(rows, E, generator=g), torch.randn(rows, E, generator=g)
prm = [1.0 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g),
       torch.randn(Fh, E, generator=g) / E ** 0.5, 0.1 * torch.randn(Fh, generator=g),
       torch.randn(E, Fh, generator=g) / Fh ** 0.5, 0.1 * torch.randn(E, generator=g),
       1.0 + 0.2 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)]
gout = torch.randn(rows, E, generator=g)
keep = (torch.rand(rows * (2 * E + Fh), generator=g) < 0.9).to(torch.uint8)
m1, mf, m2 = torch.split(keep, [rows * E, rows * Fh, rows * E])
scale = 1 / 0.9
dm = lambda m, shape: m.view(shape).double() * scale
ref_in = [t.clone().double().requires_grad_(True) for t in [x, sa] + prm]
rx, rsa, g1, be1, W1, b1, W2, b2, g2, be2 = ref_in
x1 = F.layer_norm(rx + rsa * dm(m1, (rows, E)), (E,), g1, be1, 1e-5)
pre = F.linear(x1, W1, b1)
ff = F.linear(F.relu(pre) * dm(mf, (rows, Fh)), W2, b2)
ref = F.layer_norm(x1 + ff * dm(m2, (rows, E)), (E,), g2, be2, 1e-5)
ref.backward(gout.double())
dev = [t.clone().cuda().requires_grad_(True) for t in [x, sa] + prm]
d1, df, d2 = torch.split(keep.cuda(), [rows * E, rows * Fh, rows * E])
out = nnkernels.EncoderTail.apply(dev[0], dev[1], d1, df, d2, *dev[2:], scale, 1e-5, 1e-5)
out.backward(gout.cuda())
print("out err", (out.cpu().double() - ref).abs().max().item())
names = ("g_x", "g_sa", "g_gamma1", "g_beta1", "g_W1", "g_b1", "g_W2", "g_b2", "g_gamma2", "g_beta2")
for name, d, r in zip(names, dev, ref_in):
    err = (d.grad.cpu().double() - r.grad).abs()
    sc = r.grad.abs().max().item()
    bad = (err > 2e-4 * sc)
    print(name, "max err %.3e scale %.3e bad %d" % (err.max().item(), sc, int(bad.sum())), "rows", sorted(set(bad.nonzero()[:, 0].tolist()))[:10] if bad.any() and bad.dim() == 2 else "")
amb = (pre.abs() < 3e-6) & (mf.view(rows, Fh) > 0)
print("ambiguous pre-activations:", amb.nonzero().tolist(), pre[amb].tolist())
