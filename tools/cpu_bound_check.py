"""Is the training step launch-bound?  Time to ENQUEUE K steps (host returns) vs time until the device drains.
usage: python tools/cpu_bound_check.py [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
opts = MonodepthOptions().parse(bench.CONFIG_B + os.environ.get("SQD_BENCH_EXTRA", "").split())
tr = Trainer(opts)
tr.set_train()
inputs = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
for _ in range(5):
    tr.train_step(dict(inputs))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    tr.train_step(dict(inputs))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, device drained after %.2f ms/step (host idle tail %.2f ms total)" %
      ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.train_step(dict(inputs))
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
