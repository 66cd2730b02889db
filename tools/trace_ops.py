"""Attribute the small ATen kernels of one training step to source lines (torch.profiler, with_stack).
usage: python tools/trace_ops.py [aten-op-substring ...]     default: copy_ add fill_ sum"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import bench  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402

keys = sys.argv[1:] or ["copy_", "aten::add", "fill_", "aten::sum", "aten::mul", "aten::cat"]
opts = MonodepthOptions().parse(bench.CONFIG_B + os.environ.get("SQD_BENCH_EXTRA", "").split())
tr = Trainer(opts)
tr.set_train()
inputs = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
for _ in range(3):
    tr.train_step(dict(inputs))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(dict(inputs))
    torch.cuda.synchronize()
def chain(e):
    names = []
    while e is not None and len(names) < 6:
        names.append(e.name[:48])
        e = e.cpu_parent
    return " <- ".join(names)


rows = {}
for e in prof.events():
    if any(k in e.name for k in keys) and e.device_time_total > 0 and e.name.startswith("aten::"):
        k = (e.name, str(e.input_shapes)[:60], chain(e.cpu_parent))
        t, c = rows.get(k, (0.0, 0))
        rows[k] = (t + e.device_time_total, c + 1)
for (name, shp, ch), (t, c) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:50]:
    print("%8.1f us x%-3d %-14s %-60s %s" % (t, c, name, shp, ch))
