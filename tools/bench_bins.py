"""Timing of the bins-head kernels alone (config B sizes). usage: python tools/bench_bins.py
(a sweep of the workgroup count, 256..1536, is flat within 5 %: 768 stays)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "sfmnext-impl_amd"))
import torch
from sqd import lib as _l
if os.environ.get('SQD_LIB'):
    _l.SO_PATH = os.path.abspath(os.environ['SQD_LIB']); _l.needs_build = lambda: False      # another build (tools/build_variant.sh)
from sqd import ops
B, Q, D, h, w = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (12, 64, 64, 96, 320)      # config C: 8 128 128 160 512
torch.manual_seed(0)
a = [torch.randn(B, Q, h, w, device="cuda").requires_grad_(True), (0.3 * torch.randn(D, Q, 1, 1, device="cuda")).requires_grad_(True),
     torch.randn(D, device="cuda").requires_grad_(True), (torch.rand(B, D, device="cuda") * 80).requires_grad_(True)]
g = torch.randn(B, 1, h, w, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
from sqd import lib
for arith, name in ((1, "f16x2"), (0, "fp32 MFMA")):
    lib.check(lib.lib().sqd_bins_set_arith(arith), "bins_set_arith")
    print(name, end=": ")
    _bench = True
    print("fwd %.1f us  fwd+bwd %.1f us" % (t(lambda: ops.BinsHead.apply(*a)), t(lambda: torch.autograd.grad(ops.BinsHead.apply(*a), a, g))))
lib.check(lib.lib().sqd_bins_set_arith(1), "bins_set_arith")
