"""The multi-rank code path on the device (file name sorts last on purpose: it spawns processes).

One GPU per box, so the communicator has ONE rank: the averaged gradient equals the local one and every leg must train
like the plain single-process run — but every RCCL call of the N-rank path is issued (unique id, communicator init,
parameter broadcast, bucketed all-reduces from the autograd hooks, in-graph all-reduces, barrier, max-reduce of the time).
The world-2 arithmetic of the same reducer runs on CPU over gloo in tests/test_host_logic.py.
Each leg is its own test and prints the child's whole stderr on failure."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys
sys.path[:0] = [%(repo)r, os.path.join(%(repo)r, "sfmnext-impl_amd"), os.path.join(%(repo)r, "tests"), os.path.join(%(repo)r, "tests", "golden")]
import torch
import test_gpu_graph as T
from sqd import ddp as _ddp
_announced = []
_orig_announce = _ddp.GradBucketReducer.on_deferred_grad
_ddp.GradBucketReducer.on_deferred_grad = lambda self, w: (_announced.append(1), _orig_announce(self, w))[1]
if os.environ.get("SQD_TEST_FAIL_OVERLAP_CAPTURE"):
    # the first capture attempt (all-reduces inside the step graph) dies in the reducer's join: the Trainer must fall back to "post"
    _orig_finish = _ddp.GradBucketReducer.finish
    def _finish(self):
        if torch.cuda.is_current_stream_capturing() and not getattr(self, "_failed_once", False):
            self._failed_once = True
            raise RuntimeError("injected capture failure")
        return _orig_finish(self)
    _ddp.GradBucketReducer.finish = _finish
args = [a for a in sys.argv[1:]]
tr, losses, params = T.run(args, steps=7)
out = {"reducer": tr.reducer is not None, "graph": tr._graph is not None, "losses": losses, "announced": len(_announced),
       "graph_mode": tr.graph_mode(), "capture_failures": getattr(tr, "capture_failures", []),
       "sums": {k: float(v.double().abs().sum()) for k, v in params.items() if v.dtype.is_floating_point}}
if tr.reducer is not None:
    p = next(p for p in tr.models["encoder"].parameters() if p.dim() == 4 and p.shape[2] == 3)
    out["grad_is_bucket_view"] = bool(p.grad.stride() == p.stride() and p.grad._base is not None)
    out["comm"] = type(tr.reducer.comm).__name__
    out["process_group_nccl"] = "nccl" in str(torch.distributed.get_backend()) if torch.distributed.is_initialized() else False
print("RESULT " + json.dumps(out), flush=True)
from sqd import ddp
ddp.shutdown()
"""


def _free_port():
    """a port nobody listens on right now (fixed ports collide with the TIME_WAIT sockets of a previous run)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _dist_env():
    return {"SQD_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
            "MASTER_PORT": _free_port(), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def _run(env_extra, args=(), what=""):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"repo": REPO}, *args], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line and r.returncode == 0, "%s: rc %d\n--- stdout ---\n%s\n--- stderr ---\n%s" % (what, r.returncode, r.stdout[-4000:], r.stderr[-12000:])
    return json.loads(line[-1][7:])


@pytest.fixture(scope="module")
def plain():
    return _run({}, ["--sqd_no_graph"], "plain eager run")


def _same_training(plain, dist):
    for a, b in zip(plain["losses"], dist["losses"]):
        assert abs(a - b) <= 1e-5 * abs(a) + 1e-7, (plain["losses"], dist["losses"])
    worst = max(abs(plain["sums"][k] - dist["sums"][k]) / (abs(plain["sums"][k]) + 1e-3) for k in plain["sums"])
    assert worst < 2e-2, worst


def _check_dist(res, graph):
    assert res["reducer"] and res["grad_is_bucket_view"] and res["graph"] == graph
    assert res["comm"] == "RcclComm" and not res["process_group_nccl"]      # the library's communicator, no ProcessGroupNCCL


def test_communicator_collectives_eager_and_captured():
    """sqd_comm_* directly: sum / average / max all-reduce, broadcast, barrier — and an all-reduce captured into a hipGraph
    on a side stream between two kernels, replayed twice."""
    code = r"""
import os, sys
sys.path[:0] = [os.path.join(%(repo)r, "sfmnext-impl_amd")]
import torch
from sqd import ddp
rank, world, local = ddp.init_from_env()
c = ddp.COMM
assert type(c).__name__ == "RcclComm" and c.world == 1 and c.rank == 0
x = torch.arange(1000, device="cuda", dtype=torch.float32)
for op in ("sum", "avg", "max", "min"):
    y = x.clone(); c.all_reduce(y, op); assert torch.equal(y, x), op
d = torch.tensor([3.5], device="cuda", dtype=torch.float64); c.all_reduce(d, "max"); assert float(d) == 3.5
i = torch.arange(10, device="cuda", dtype=torch.int32); c.broadcast(i, 0); assert int(i.sum()) == 45
c.barrier()
# captured: kernel -> (side stream) all-reduce -> join -> kernel
buf = torch.zeros(1 << 20, device="cuda"); out = torch.zeros(1 << 20, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):                       # warm-up on the capture stream (RCCL sets up its channels lazily)
        buf.add_(1.0); c.stream.wait_stream(s); c.all_reduce(buf, "avg", stream=c.stream); s.wait_stream(c.stream); out.copy_(buf * 2)
torch.cuda.synchronize()
buf.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    buf.add_(1.0)
    c.stream.wait_stream(torch.cuda.current_stream())
    c.all_reduce(buf, "avg", stream=c.stream)
    torch.cuda.current_stream().wait_stream(c.stream)
    out.copy_(buf * 2)
assert float(buf[0]) == 0.0                  # nothing ran during the capture
g.replay(); g.replay(); torch.cuda.synchronize()
assert float(buf[7]) == 2.0 and float(out[7]) == 4.0, (float(buf[7]), float(out[7]))
try:
    c.all_reduce(torch.zeros(4, device="cuda", dtype=torch.int32), "avg")
    raise SystemExit("integer average must be refused")
except RuntimeError as e:
    assert "average of an integer" in str(e)
ddp.shutdown()
print("RESULT {}")
""" % {"repo": REPO}
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **_dist_env()), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RESULT" in r.stdout, "rc %d\n%s\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-12000:])


def test_eager_hooks_overlap_leg(plain):
    dist = _run(_dist_env(), ["--sqd_no_graph"], "eager multi-rank leg")      # hooks launch the all-reduces next to backward
    assert not plain["reducer"]
    _check_dist(dist, graph=False)
    _same_training(plain, dist)


@pytest.mark.parametrize("attempt", range(3))
def test_step_graph_with_captured_allreduces_leg(plain, attempt):
    """default multi-rank mode: ONE hipGraph holding forward, backward, the bucket gathers + RCCL all-reduces launched by the
    autograd hooks (graph branches next to the rest of backward) and Adam; a fresh process per attempt (round 2's
    ProcessGroupNCCL-based version of this leg died intermittently on a fresh box)"""
    graphed = _run(_dist_env(), [], "graph-overlap leg, attempt %d" % attempt)
    _check_dist(graphed, graph=True)
    _same_training(plain, graphed)
    # round 4: the multi-rank step runs the kernels the single-rank step is measured with — the sums of the weight gradients' pixel
    # splits ride on the BatchNorm-backward launches and the filters are announced to the reducer by hand
    assert graphed["graph_mode"] == "overlap" and graphed["announced"] > 0 and not graphed["capture_failures"], graphed


def test_overlap_capture_failure_falls_back_to_post(plain):
    """a step graph with the all-reduces inside that cannot be captured must not end the run: the Trainer retries as
    --sqd_graph_ddp post (graph of forward + backward, exchange and Adam after the replay) and says so"""
    res = _run(dict(_dist_env(), SQD_TEST_FAIL_OVERLAP_CAPTURE="1"), [], "overlap capture failure -> post")
    _check_dist(res, graph=True)
    assert res["graph_mode"] == "post" and len(res["capture_failures"]) == 1 and "injected" in res["capture_failures"][0], res
    _same_training(plain, res)


def test_graph_then_allreduce_leg(plain):
    post = _run(_dist_env(), ["--sqd_graph_ddp", "post"], "graph-post leg")    # forward+backward as a hipGraph, then all-reduce + Adam
    _check_dist(post, graph=True)
    _same_training(plain, post)


def test_bench_under_torchrun_one_rank():
    """the driver's launch route: torch.distributed.run -> bench.py, one rank, the RCCL path forced on"""
    env = dict(os.environ, SQD_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               SQD_BENCH_EXTRA="--backbone resnet18_lite --num_layers 18 --model_dim 16 --patch_size 8 --query_nums 12 --dim_out 24 "
                               "--height 64 --width 96 --batch_size 2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "5",
           "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, "rc %d\n%s\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-12000:])
    rec = json.loads(lines[-1])
    ex = rec["config"]["exchange"]
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and ex["communicator"] == "RcclComm"
    # what a multi-GPU line must say about itself
    assert ex["ranks_joined"] == 1 and ex["rccl_version"] and ex["rccl_version"][0].isdigit() and ex["defer_wgrad_reduce"] is True
    assert ex["graph_mode"] == "overlap" and ex["buckets"] == len(ex["bucket_bytes"]) >= 1 and not ex["capture_failures"]
