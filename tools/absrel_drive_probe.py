"""Probe for tests/test_gpu_abs_rel.py on the consistent "drive" scenes (dev): supervised pre-fit of the depth network, a self-supervised
warm-up on the device at the reference's learning rate, then two 50-step continuations under different convolution plans (default fp32 / timed) as
a stand-in for two arithmetic orders; abs_rel on a held-out batch along the way."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from datasets.synthetic import synthetic_batch  # noqa: E402
from finetune.train_ft_SQLdepth import FinetuneArgs, FinetuneTrainer  # noqa: E402
from options import MonodepthOptions  # noqa: E402
from trainer import Trainer  # noqa: E402
from sqd import nnkernels  # noqa: E402

H, W, B = 192, 640, 2
ARGS = ["--backbone", "resnet", "--num_layers", "50", "--num_features", "256", "--model_dim", "32", "--patch_size", "16",
        "--query_nums", "64", "--dim_out", "64", "--height", str(H), "--width", str(W), "--batch_size", str(B),
        "--min_depth", "0.001", "--max_depth", "80.0", "--num_workers", "0", "--sqd_synthetic", "--log_dir", "/tmp/sqd_absrel_probe"]
PRE_STEPS, PRE_B, PRE_NB = int(os.environ.get("PRE_STEPS", 400)), 4, 16
WARM, NB = int(os.environ.get("WARM", 400)), 16
DSCALE = 20.0          # the depth network is fitted to depth / 20 (abs_rel is median-scaled): the pose network's translation output then is O(0.02)
POSE_STEPS = int(os.environ.get("POSE_STEPS", 600))
torch.set_num_threads(16)
torch.manual_seed(0)
t0 = time.time()


def no_dropout(models):
    for m in models:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0


a = list(ARGS)
a[a.index("--batch_size") + 1] = str(PRE_B)
ft = FinetuneTrainer(MonodepthOptions().parse(a), FinetuneArgs(bs=PRE_B, epochs=1, lr=1e-4), steps_per_epoch=PRE_STEPS)
pre = []
for i in range(PRE_NB):
    s = synthetic_batch(PRE_B, H, W, start=PRE_B * i, scene="drive", with_gt=True, device="cuda")
    pre.append({"image": s[("color_aug", 0, 0)], "depth": F.interpolate(s["depth_gt"], [H, W], mode="nearest") / DSCALE})
print("pre-fit batches generated %.0f s" % (time.time() - t0), flush=True)
ft.model.train()
for i in range(PRE_STEPS):
    ft.train_step(pre[i % PRE_NB])
torch.cuda.synchronize()
state = {"encoder": {k: v.detach().clone() for k, v in ft.model.encoder.state_dict().items()},
         "depth": {k: v.detach().clone() for k, v in ft.model.depth_decoder.state_dict().items()}}
batches = [synthetic_batch(B, H, W, start=1000 + B * i, scene="drive", with_gt=True) for i in range(NB)]
held = synthetic_batch(4, H, W, start=10 ** 5, with_gt=True, scene="drive")
print("all batches generated %.0f s" % (time.time() - t0), flush=True)


def metrics(tr):
    tr.set_eval()
    with torch.no_grad():
        inputs = {k: v.cuda() for k, v in held.items()}
        outputs, losses = tr.process_batch(inputs)
        tr.compute_depth_losses(inputs, outputs, losses)
    tr.set_train()
    return [float(losses[n]) for n in tr.depth_metric_names]


def run(extra, start_state, steps, seed, report_every=0):
    nnkernels.reset_plans()
    tr = Trainer(MonodepthOptions().parse(ARGS + extra))
    tr.set_train()
    no_dropout(tr.models.values())
    for name, sd in start_state.items():
        tr.models[name].load_state_dict(sd)
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(steps):
        dev = {k: v.cuda() for k, v in batches[i % NB].items() if k != "depth_gt"}
        dev[("noise", 0)] = torch.randn(B, 2, H, W, generator=g).cuda()
        loss = float(tr.train_step(dev)[1]["loss"].detach())
        out.append(loss)
        if report_every and (i + 1) % report_every == 0:
            print("   step %4d loss %.5f abs_rel %.4f" % (i + 1, loss, metrics(tr)[0]), flush=True)
    m = metrics(tr)
    st = {n: {k: v.detach().clone() for k, v in tr.models[n].state_dict().items()} for n in ("encoder", "depth", "pose")}
    nnkernels.reset_plans()
    return m, out, st


# ---- supervised pre-fit of the pose network on the known motions (device kernels, torch's Adam)
from datasets.synthetic import drive_motion, _rodrigues  # noqa: E402


def pose_targets(start, n, depth_gt):
    """what PoseCNN must output for frames -1 / +1 of samples start .. start + n - 1 (reference trainer.py:319-337,417-421: the pair is
    ordered in time, frame -1's transform is inverted, the translation is multiplied by the mean inverse depth)"""
    mid = (DSCALE / depth_gt).mean((1, 2, 3)).double()                                    # mean inverse (scaled) depth per sample
    aa, tt = torch.zeros(n, 2, 3, dtype=torch.float64), torch.zeros(n, 2, 3, dtype=torch.float64)
    for i in range(n):
        for j, f in enumerate((-1, 1)):
            w, t = drive_motion(start + i, f)
            R = _rodrigues(w)
            if f < 0:
                aa[i, j], tt[i, j] = -w, -(R.T @ t) / DSCALE / mid[i]
            else:
                aa[i, j], tt[i, j] = w, t / DSCALE / mid[i]
    return aa.float().cuda(), tt.float().cuda()


trp = Trainer(MonodepthOptions().parse(ARGS + ["--sqd_no_graph", "--sqd_no_conv_tune"]))
no_dropout(trp.models.values())
pose = trp.models["pose"]
pose.train()
popt = torch.optim.Adam(pose.parameters(), float(os.environ.get("POSE_LR", 1e-4)))
ptg = [pose_targets(1000 + B * i, B, batches[i]["depth_gt"] if "depth_gt" in batches[i] else synthetic_batch(B, H, W, start=1000 + B * i, scene="drive", with_gt=True)["depth_gt"]) for i in range(NB)]
for it in range(POSE_STEPS):
    b = batches[it % NB]
    aug = {f: b[("color_aug", f, 0)].cuda() for f in (0, -1, 1)}
    nnkernels.begin_step()
    aa, tr_ = pose.forward_pairs([(aug[-1], aug[0]), (aug[0], aug[1])])
    ta, tt_ = ptg[it % NB]
    loss = ((aa.reshape(B, 2, 3) - ta) ** 2).mean() + ((tr_.reshape(B, 2, 3) - tt_) ** 2).mean()
    popt.zero_grad()
    loss.backward()
    popt.step()
    if (it + 1) % 100 == 0:
        print("   pose pre-fit step %d: mse %.3e (targets rms %.3e)" % (it + 1, float(loss), float((ta ** 2).mean().sqrt() + (tt_ ** 2).mean().sqrt())), flush=True)
state["pose"] = {k: v.detach().clone() for k, v in pose.state_dict().items()}
del trp
tr0 = Trainer(MonodepthOptions().parse(ARGS + ["--sqd_no_graph", "--sqd_no_conv_tune"]))
no_dropout(tr0.models.values())
for name, sd in state.items():
    tr0.models[name].load_state_dict(sd)
print("after the supervised pre-fit: abs_rel %.4f" % metrics(tr0)[0], flush=True)
del tr0
print("self-supervised warm-up on the device, lr 1e-4:")
m, losses, warm = run([], state, WARM, 3, report_every=50)
print("after the warm-up: metrics %s" % ["%.4f" % v for v in m])
for seed in (11, 12, 13):
    ma, la, _ = run(["--sqd_no_conv_tune"], warm, 50, seed)
    mb, lb, _ = run([], warm, 50, seed)
    print("seed %d: 50 steps at 1e-4: abs_rel default plans %.5f, timed plans %.5f, |delta| %.2e; worst per-step loss difference %.2e"
          % (seed, ma[0], mb[0], abs(ma[0] - mb[0]), max(abs(x - y) / abs(y) for x, y in zip(la, lb))), flush=True)
