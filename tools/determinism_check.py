"""Run-to-run and eager-vs-graph parameter drift after 7 training steps of the small test configuration (tests/test_gpu_graph.py)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "sfmnext-impl_amd"), os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import torch
import test_gpu_graph as T
def diff(pa, pb):
    worst = ("", 0.0)
    for k in pa:
        a, b = pa[k].float(), pb[k].float()
        d = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-3)
        if d > worst[1]: worst = (k, d)
    return worst
_, l1, p1 = T.run(["--sqd_no_graph"])
_, l2, p2 = T.run(["--sqd_no_graph"])
_, l3, p3 = T.run([])
_, l4, p4 = T.run([])
print("eager vs eager", diff(p1, p2), [abs(a-b) for a, b in zip(l1, l2)][-1])
print("graph vs graph", diff(p3, p4), [abs(a-b) for a, b in zip(l3, l4)][-1])
print("eager vs graph", diff(p1, p3), [abs(a-b) for a, b in zip(l1, l3)][-1])
