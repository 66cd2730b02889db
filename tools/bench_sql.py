"""Timing of the Self Query Layer kernels alone (dev tool). usage: python tools/bench_sql.py [B E Q h w]   (default: config B; config C: 8 32 128 160 512)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "sfmnext-impl_amd"))
import torch
from sqd import lib as _l
if os.environ.get('SQD_LIB'):
    _l.SO_PATH = os.path.abspath(os.environ['SQD_LIB']); _l.needs_build = lambda: False      # another build (tools/build_variant.sh)
from sqd import ops
B, E, Q, h, w = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (12, 32, 64, 96, 320)
torch.manual_seed(0)
x = torch.randn(B, E, h, w, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
K = (0.3 * torch.randn(B, Q, E, device="cuda")).requires_grad_(True)
gy, gs = torch.randn(B, Q, h, w, device="cuda"), torch.randn(B, Q, E, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    tf = t(lambda: ops.SelfQueryLayer.apply(x, K))
tfb = t(lambda: torch.autograd.grad(ops.SelfQueryLayer.apply(x, K), (x, K), (gy, gs)))
ybytes = B * Q * h * w * 4
print("B=%d E=%d Q=%d N=%d: fwd (+merge) %.1f us  (y write alone at 8 TB/s: %.1f us)   fwd+bwd %.1f us" % (B, E, Q, h * w, tf, ybytes / 8e6, tfb))
