"""Debug helper: run the bench configuration under a 1-rank RCCL group and report any collective issued while a stream is
capturing.  usage (GPU box): SQD_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 python tools/ddp_capture_debug.py"""
import os
import sys
import traceback

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "sfmnext-impl_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

for name in ("all_reduce", "broadcast", "barrier", "all_gather"):
    orig = getattr(dist, name)

    def wrap(*a, _orig=orig, _name=name, **k):
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            print("COLLECTIVE IN CAPTURE:", _name, flush=True)
            traceback.print_stack(limit=8)
        return _orig(*a, **k)
    setattr(dist, name, wrap)

import bench  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402

opts = MonodepthOptions().parse(bench.CONFIG_B + sys.argv[1:])
tr = Trainer(opts)
tr.set_train()
inputs = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
for i in range(8):
    tr.train_step(dict(inputs))
    if os.environ.get("DBG_SYNC", "1") == "1":
        torch.cuda.synchronize()
    print("step", i, "graph" if tr._graph is not None else "eager", flush=True)
torch.cuda.synchronize()
print("done", flush=True)
