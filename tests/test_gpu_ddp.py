"""The multi-rank code path on the device: a 1-rank RCCL process group makes the Trainer use GradBucketReducer (bucket
views, autograd hooks, all-reduce calls, FusedAdam on bucket memory).  With one rank the averaged gradient equals the local
one, so 5 steps must train like the plain single-process run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys
sys.path[:0] = [%(repo)r, os.path.join(%(repo)r, "sfmnext-impl_amd"), os.path.join(%(repo)r, "tests"), os.path.join(%(repo)r, "tests", "golden")]
import torch
import test_gpu_graph as T
tr, losses, params = T.run(sys.argv[1:], steps=7)
out = {"reducer": tr.reducer is not None, "graph": tr._graph is not None, "losses": losses,
       "sums": {k: float(v.double().abs().sum()) for k, v in params.items() if v.dtype.is_floating_point}}
if tr.reducer is not None:
    p = next(p for p in tr.models["encoder"].parameters() if p.dim() == 4 and p.shape[2] == 3)
    out["grad_is_bucket_view"] = bool(p.grad.stride() == p.stride() and p.grad._base is not None)
print("RESULT " + json.dumps(out))
"""


DIST = {"SQD_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1"}


def _free_port():
    """a port nobody listens on right now (fixed ports collide with the TIME_WAIT sockets of a previous run of this test)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _run(env_extra, args=()):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"repo": REPO}, *args], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(line[-1][7:])


def _same_training(plain, dist):
    for a, b in zip(plain["losses"], dist["losses"]):
        assert abs(a - b) <= 1e-5 * abs(a) + 1e-7, (plain["losses"], dist["losses"])
    worst = max(abs(plain["sums"][k] - dist["sums"][k]) / (abs(plain["sums"][k]) + 1e-3) for k in plain["sums"])
    assert worst < 2e-2, worst


def test_one_rank_rccl_reducer_trains_like_single_process():
    plain = _run({}, ["--sqd_no_graph"])
    dist = _run(dict(DIST, MASTER_PORT=_free_port()), ["--sqd_no_graph"])          # eager: hooks overlap the all-reduces with backward
    assert not plain["reducer"] and dist["reducer"] and dist["grad_is_bucket_view"] and not dist["graph"]
    _same_training(plain, dist)
    # default multi-rank mode: ONE hipGraph holding forward, backward, the bucket gathers + RCCL all-reduces launched by the
    # autograd hooks (graph branches next to the rest of backward) and Adam
    graphed = _run(dict(DIST, MASTER_PORT=_free_port()), [])
    assert graphed["reducer"] and graphed["graph"] and graphed["grad_is_bucket_view"]
    _same_training(plain, graphed)
    post = _run(dict(DIST, MASTER_PORT=_free_port()), ["--sqd_graph_ddp", "post"])    # forward+backward as a hipGraph, then all-reduce + Adam
    assert post["reducer"] and post["graph"] and post["grad_is_bucket_view"]
    _same_training(plain, post)
