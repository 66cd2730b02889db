"""Dev tool (round 4): which parameters reach the bucket reducer twice in one captured multi-rank step, and from where.
Run on a GPU box as a 1-rank job: SQD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python tools/ddp_deferred_debug.py
(found: torch fires post-accumulate hooks for an AccumulateGrad node that received NO gradient — the deferred filters were counted by the hook AND by their
announcement; ddp.GradBucketReducer._on_grad now ignores the hook call for nnkernels.DEFERRED_FILTERS)"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "sfmnext-impl_amd"), os.path.join(REPO, "tests"), os.path.join(REPO, "tests", "golden")]
import torch
import test_gpu_graph as T
from sqd import ddp as _ddp
orig = _ddp.GradBucketReducer._on_grad
names = {}
log = []
seen = {}
import traceback
oz = _ddp.GradBucketReducer.zero_grad
def zg(self):
    seen.clear()
    return oz(self)
_ddp.GradBucketReducer.zero_grad = zg
dups = [0]
def on_grad(self, p, announced=False):
    bi = self._bucket_of.get(p)
    if id(p) in seen and dups[0] < 2:
        dups[0] += 1
        print("DUPLICATE arrival of", names.get(id(p)), "\nFIRST:\n", seen[id(p)], "\nSECOND:\n", "".join(traceback.format_stack(limit=8)), flush=True)
    seen[id(p)] = "".join(traceback.format_stack(limit=8))
    log.append((names.get(id(p), "?") + (" [announced]" if announced else ""), bi, None if bi is None else self._pending[bi], p.grad is None))
    if bi is not None and self._pending[bi] == 1:
        missing = [names.get(id(q), "?") for q in self.buckets[bi] if q.grad is None]
        if missing:
            print("BUCKET", bi, "completes with missing grads:", missing, flush=True)
            cnt = {}
            for n, b, pend, gn in log:
                if b == bi: cnt[n] = cnt.get(n, 0) + 1
            print("announce counts >1:", {k: v for k, v in cnt.items() if v > 1}, flush=True)
    return orig(self, p, announced=announced)
_ddp.GradBucketReducer._on_grad = on_grad
from trainer import Trainer
oi = Trainer.__init__
def init(self, *a, **k):
    oi(self, *a, **k)
    for mn, m in self.models.items():
        for n, p in m.named_parameters():
            names[id(p)] = mn + "." + n
Trainer.__init__ = init
tr, losses, params = T.run([], steps=6)
print("mode", tr.graph_mode(), getattr(tr, "capture_failures", []))
_ddp.shutdown()
