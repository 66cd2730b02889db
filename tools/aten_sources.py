"""Which Python lines still run ATen operators on device tensors in a training step (dev tool): bench.py's configuration, eager
(no hipGraph), single-threaded autograd, a TorchDispatchMode that records every aten op touching a CUDA tensor together with the
innermost frame inside this repository."""
import collections
import os
import sys
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
from torch.utils._pytree import tree_flatten  # noqa: E402

import bench  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402

SKIP = ("aten::empty", "aten::view", "aten::as_strided", "aten::detach", "aten::alias", "aten::_unsafe_view", "aten::reshape", "aten::permute",
        "aten::transpose", "aten::t", "aten::expand", "aten::select", "aten::slice", "aten::unsqueeze", "aten::squeeze", "aten::empty_like",
        "aten::empty_strided", "aten::_local_scalar_dense", "aten::record_stream", "aten::unbind", "aten::split", "aten::lift_fresh", "aten::is_pinned",
        "aten::set_", "aten::resize_", "aten::narrow", "aten::unflatten", "aten::new_empty")
rows = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if not name.startswith(SKIP):
            flat, _ = tree_flatten((args, kwargs, out))
            ts = [t for t in flat if isinstance(t, torch.Tensor)]
            if any(t.is_cuda for t in ts):
                fr = [f for f in traceback.extract_stack() if "sfmnext-impl_amd" in f.filename and "tools/" not in f.filename]
                where = "%s:%d %s" % (fr[-1].filename.replace(REPO + "/sfmnext-impl_amd/", ""), fr[-1].lineno, fr[-1].name) if fr else "?"
                shape = next((tuple(t.shape) for t in ts if t.is_cuda), ())
                rows[(name, where, str(shape)[:40])] += 1
        return out


opts = MonodepthOptions().parse(bench.CONFIG_B + ["--sqd_no_graph"] + os.environ.get("SQD_BENCH_EXTRA", "").split())
tr = Trainer(opts)
tr.set_train()
inputs = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
for _ in range(3):
    tr.train_step(dict(inputs))
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with Log():
    tr.train_step(dict(inputs))
torch.cuda.synchronize()
agg = collections.Counter()
for (name, where, shape), n in rows.items():
    agg[(name, where)] += n
for (name, where), n in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    shapes = sorted({s for (nm, wh, s), _ in rows.items() if nm == name and wh == where})[:3]
    print("%3d  %-26s %-62s %s" % (n, name, where, " ".join(shapes)))
