"""Bisect what breaks hipGraph capture of the step: python tools/graph_bisect.py <stage 1|2|3> [small]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402

stage = int(sys.argv[1])
opts = MonodepthOptions().parse(bench.CONFIG_B + ["--sqd_no_graph"] + os.environ.get("SQD_BENCH_EXTRA", "").split())
tr = Trainer(opts)
tr.set_train()
inputs = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, device=tr.device)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        tr.train_step(dict(inputs))
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
opt = tr.model_optimizer
opt.zero_grad(set_to_none=True)
opt.begin_capture()
opt.refresh_hyper()
g = torch.cuda.CUDAGraph()
tr._capturing = True
static = {k: v.clone() for k, v in inputs.items()}
print("capturing stage", stage, flush=True)
with torch.cuda.graph(g):
    if stage == 0:
        with torch.no_grad():
            out = tr.models["encoder"](tr._fmt(static["color_aug", 0, 0]))
    else:
        outputs, losses = tr.process_batch(static)
        if stage >= 2:
            losses["loss"].backward()
        if stage >= 3:
            opt.step()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed stage", stage, "ok", flush=True)
