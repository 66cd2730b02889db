"""abs_rel parity after equal steps (BASELINE.json north_star: "abs_rel within +-0.001 of reference after equal steps";
SURVEY.md §8d): the HIP trainer and the oracle restatement start from identical weights and see identical batches and
tie-break noise for K = 200 optimisation steps on the synthetic set; then both predict a held-out synthetic batch and
compute_depth_losses (reference trainer.py:551-579, layers.py:282-300: 375x1242, eigen crop, median scaling) must agree
on abs_rel to 1e-3.  Small shape (ResNet-18, 64x96) keeps the oracle's 200 CPU steps within a minute; every kernel of the path runs."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, B, STEPS, NBATCH = 64, 96, 2, 200, 8
ARGS = ["--backbone", "resnet18_lite", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
        "--height", str(H), "--width", str(W), "--batch_size", str(B), "--num_workers", "0", "--sqd_synthetic",
        "--log_dir", "/tmp/sqd_absrel_test", "--max_depth", "80.0", "--sqd_no_conv_tune"]


def _no_dropout(models):
    for m in models:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0


@pytest.fixture(scope="module")
def oracle_run():
    """the oracle's 200 steps (CPU), once for both device runs"""
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from datasets.synthetic import synthetic_batch
    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    enc = O.LiteResnetEncoderDecoder(model_dim=16)
    dep = O.QueryTrDecoder(16, 16, 8, 4, 12, 24, min_val=0.001, max_val=80.0, dim_feedforward=512, dropout=0.0)
    pose = O.PoseCNN(2)
    for m in (enc, dep, pose):
        m.train()
    state = {"encoder": {k: v.clone() for k, v in enc.state_dict().items()}, "depth": {k: v.clone() for k, v in dep.state_dict().items()},
             "pose": {k: v.clone() for k, v in pose.state_dict().items()}}
    g = torch.Generator().manual_seed(5)
    batches = [synthetic_batch(B, H, W, start=B * i) for i in range(NBATCH)]
    noises = [torch.randn(B, 2, H, W, generator=g) for _ in range(STEPS)]
    held = synthetic_batch(B, H, W, start=10 ** 5, with_gt=True)
    ref = O.RefTrainStep(enc, dep, pose, (0, -1, 1), H, W)
    ref_loss = [float(ref.step(dict(batches[i % NBATCH]), noises[i])[1]["loss"].detach()) for i in range(STEPS)]
    for m in (enc, dep, pose):
        m.eval()
    with torch.no_grad():
        out = dep(enc(held[("color_aug", 0, 0)]))
        depth = torch.nn.functional.interpolate(out[("disp", 0)], [H, W], mode="bilinear", align_corners=False)
    want = [float(v) for v in O.compute_depth_losses(depth, held["depth_gt"])]
    return {"state": state, "batches": batches, "noises": noises, "held": held, "ref_loss": ref_loss, "want": want}


@pytest.mark.parametrize("plans", ["default_plans", "tuned_plans"])
def test_abs_rel_after_200_steps(oracle_run, plans):
    """default_plans: the library's cost-model (fp32 MFMA) plans; tuned_plans: first-step plan timing on, as benchmarked
    (three-term bf16 / input-patch kernels where they are faster)"""
    from options import MonodepthOptions
    from trainer import Trainer
    from sqd import nnkernels
    R = oracle_run
    nnkernels.reset_plans()
    tr = Trainer(MonodepthOptions().parse(ARGS if plans == "default_plans" else [a for a in ARGS if a != "--sqd_no_conv_tune"]))
    tr.set_train()
    _no_dropout(tr.models.values())
    for name, sd in R["state"].items():
        tr.models[name].load_state_dict(sd)
    dev_loss = []
    try:
        for i in range(STEPS):
            dev = {k: v.cuda() for k, v in R["batches"][i % NBATCH].items()}
            dev[("noise", 0)] = R["noises"][i].cuda()
            dev_loss.append(float(tr.train_step(dev)[1]["loss"].detach()))
        assert tr._graph is not None                       # the replayed hipGraph did the training
        mix = nnkernels.plan_mix()
        tr.set_eval()
        with torch.no_grad():
            inputs = {k: v.cuda() for k, v in R["held"].items()}
            outputs, losses = tr.process_batch(inputs)
            tr.compute_depth_losses(inputs, outputs, losses)
        got = [float(losses[n]) for n in tr.depth_metric_names]
    finally:
        nnkernels.reset_plans()
    ref_loss, want = R["ref_loss"], R["want"]
    worst = max(abs(a - b) / abs(b) for a, b in zip(dev_loss, ref_loss))
    print("%s: loss after %d steps: device %.6f oracle %.6f (first step %.6f), worst per-step relative difference %.2e; depth metrics device %s oracle %s; plans %s"
          % (plans, STEPS, dev_loss[-1], ref_loss[-1], ref_loss[0], worst, ["%.5f" % v for v in got], ["%.5f" % v for v in want], mix))
    first, last = sum(ref_loss[:NBATCH]) / NBATCH, sum(ref_loss[-NBATCH:]) / NBATCH
    assert last < first                                # the model did train (same batches, 25 passes later)
    if plans == "tuned_plans":
        assert sum(mix.get("fwd", {}).values()) > 10, mix
    assert abs(got[0] - want[0]) <= 1e-3, ("abs_rel", got[0], want[0])
    # measured on MI355X (profiles/r03c): worst per-step relative loss difference over the 200 steps 9.6e-4 (default plans) / 3.1e-4
    # (measured plans), final loss equal to 1e-5, |d abs_rel| 2.3e-4 / 0.9e-4 — the bars are ~3x that
    assert worst <= 3e-3, worst
    assert abs(dev_loss[-1] - ref_loss[-1]) <= 5e-4 * abs(ref_loss[-1]), (dev_loss[-1], ref_loss[-1])


# ------------------------------------------------------------------------------------------------------------------------------------
# The same claim on the metric's own model (BASELINE.json: "KITTI 640x192 ResNet-50 ... abs_rel within +-0.001 of reference after equal
# steps"; reference trainer.py:551-579, layers.py:282-300): ResNet-50 + Depth_Decoder_QueryTr at 192x640 — configs[1]'s networks and
# image size; batch 2 keeps the oracle's CPU steps at ~1 s each — 50 optimisation steps through the replayed hipGraph with the plans
# measured in the first step (the benchmarked arithmetic), then compute_depth_losses on a held-out batch.
R50_H, R50_W, R50_B, R50_STEPS, R50_NBATCH = 192, 640, 2, 50, 5
R50_ARGS = ["--backbone", "resnet", "--num_layers", "50", "--num_features", "256", "--model_dim", "32", "--patch_size", "16",
            "--query_nums", "64", "--dim_out", "64", "--height", str(R50_H), "--width", str(R50_W), "--batch_size", str(R50_B),
            "--min_depth", "0.001", "--max_depth", "80.0", "--num_workers", "0", "--sqd_synthetic", "--log_dir", "/tmp/sqd_absrel_test"]


@pytest.fixture(scope="module")
def oracle_run_res50():
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from datasets.synthetic import synthetic_batch
    torch.manual_seed(0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    enc = O.ResnetEncoderDecoder(50, 256, 32)
    dep = O.QueryTrDecoder(32, 32, 16, 4, 64, 64, min_val=0.001, max_val=80.0, dim_feedforward=1024, dropout=0.0)
    pose = O.PoseCNN(2)
    for m in (enc, dep, pose):
        m.train()
    state = {"encoder": {k: v.clone() for k, v in enc.state_dict().items()}, "depth": {k: v.clone() for k, v in dep.state_dict().items()},
             "pose": {k: v.clone() for k, v in pose.state_dict().items()}}
    g = torch.Generator().manual_seed(7)
    batches = [synthetic_batch(R50_B, R50_H, R50_W, start=R50_B * i) for i in range(R50_NBATCH)]
    noises = [torch.randn(R50_B, 2, R50_H, R50_W, generator=g) for _ in range(R50_STEPS)]
    held = synthetic_batch(R50_B, R50_H, R50_W, start=10 ** 5, with_gt=True)
    ref = O.RefTrainStep(enc, dep, pose, (0, -1, 1), R50_H, R50_W)
    ref_loss = [float(ref.step(dict(batches[i % R50_NBATCH]), noises[i])[1]["loss"].detach()) for i in range(R50_STEPS)]
    for m in (enc, dep, pose):
        m.eval()
    with torch.no_grad():
        out = dep(enc(held[("color_aug", 0, 0)]))
        depth = torch.nn.functional.interpolate(out[("disp", 0)], [R50_H, R50_W], mode="bilinear", align_corners=False)
    want = [float(v) for v in O.compute_depth_losses(depth, held["depth_gt"])]
    return {"state": state, "batches": batches, "noises": noises, "held": held, "ref_loss": ref_loss, "want": want}


def test_abs_rel_resnet50_192x640_after_50_steps(oracle_run_res50):
    from options import MonodepthOptions
    from trainer import Trainer
    from sqd import nnkernels
    R = oracle_run_res50
    nnkernels.reset_plans()
    tr = Trainer(MonodepthOptions().parse(R50_ARGS))            # plan timing ON, graph replay ON: what bench.py runs
    tr.set_train()
    _no_dropout(tr.models.values())
    for name, sd in R["state"].items():
        tr.models[name].load_state_dict(sd)
    dev_loss = []
    try:
        for i in range(R50_STEPS):
            dev = {k: v.cuda() for k, v in R["batches"][i % R50_NBATCH].items()}
            dev[("noise", 0)] = R["noises"][i].cuda()
            dev_loss.append(float(tr.train_step(dev)[1]["loss"].detach()))
        assert tr._graph is not None                       # the replayed hipGraph did the training
        mix = nnkernels.plan_mix()
        tr.set_eval()
        with torch.no_grad():
            inputs = {k: v.cuda() for k, v in R["held"].items()}
            outputs, losses = tr.process_batch(inputs)
            tr.compute_depth_losses(inputs, outputs, losses)
        got = [float(losses[n]) for n in tr.depth_metric_names]
    finally:
        nnkernels.reset_plans()
    ref_loss, want = R["ref_loss"], R["want"]
    worst = max(abs(a - b) / abs(b) for a, b in zip(dev_loss, ref_loss))
    print("ResNet-50 192x640: loss after %d steps: device %.6f oracle %.6f (first step %.6f), worst per-step relative difference %.2e; "
          "depth metrics device %s oracle %s; plans %s"
          % (R50_STEPS, dev_loss[-1], ref_loss[-1], ref_loss[0], worst, ["%.5f" % v for v in got], ["%.5f" % v for v in want], mix))
    assert sum(mix.get("fwd", {}).values()) > 20 and sum(mix.get("wgrad", {}).values()) > 20, mix      # the layers were timed
    first, last = sum(ref_loss[:R50_NBATCH]) / R50_NBATCH, sum(ref_loss[-R50_NBATCH:]) / R50_NBATCH
    assert last < first                                # the model did train
    assert abs(got[0] - want[0]) <= 1e-3, ("abs_rel", got[0], want[0])          # BASELINE.json north_star
    assert worst <= 1e-2, worst


# ------------------------------------------------------------------------------------------------------------------------------------
# The comparison from weights that HAVE trained, at the reference's learning rate (round-5 verdict, item 7).
# Scenes: datasets/synthetic.py scene="drive" (round 6) — texture and depth analytic in the target coordinates, source frames by the exact
# inverse warp, exact rotations, nothing added after the warp: with the true depth and motion the reference's warp reproduces the target
# to 5e-4 (L1), and 10 % of depth error costs 16x that.  Round 5's scenes sampled the target texture at the FORWARD projection; their
# self-supervised optimum sat at abs_rel 0.31 and the compared stretch had to run at --learning_rate 1e-6.
# Pre-fit on the device alone: the depth network supervised (SILog on depth / 20: abs_rel is median-scaled, and the pose network's
# translation output, which the reference multiplies by the mean inverse depth, then is O(0.02)), the pose network supervised on the known
# motions, then 1500 self-supervised steps at 1e-4 — abs_rel stays at 0.11 - 0.15 (it went 0.07 -> 0.31 in ten steps on round 5's scenes).
# Compared stretch: 30 self-supervised steps at the default --learning_rate 1e-4, three seeds of tie-break noise.
# What can be asked of it: near its fixed point Adam's update is lr * sign-like in every weight whose gradient is at rounding level, so
# two CORRECT fp32 evaluations decorrelate there at lr per step — measured here on the oracle itself (the same 30 steps from weights
# perturbed by 1e-7 relative): |delta abs_rel| of that pair is the floor, the device has to stay within max(1e-3, 3 x the largest floor seen) of the oracle,
# and its FIRST compared step (identical weights) within 2e-5 of the oracle's loss.
PRE_STEPS, PRE_B, PRE_NBATCH, POSE_STEPS, WARM_STEPS, CMP_STEPS, CMP_NBATCH, DSCALE = 1600, 4, 16, 3000, 1500, 30, 16, 20.0


def _pose_targets(start, n, depth_gt):
    """what PoseCNN must output for frames -1 / +1 of "drive" samples start .. start + n - 1 (reference trainer.py:319-337,417-421: pairs in
    temporal order, frame -1's transform inverted, translation x mean inverse depth)"""
    from datasets.synthetic import drive_motion, _rodrigues
    mid = (DSCALE / depth_gt).mean((1, 2, 3)).double()
    aa, tt = torch.zeros(n, 2, 3, dtype=torch.float64), torch.zeros(n, 2, 3, dtype=torch.float64)
    for i in range(n):
        for j, f in enumerate((-1, 1)):
            w, t = drive_motion(start + i, f)
            if f < 0:
                aa[i, j], tt[i, j] = -w, -(_rodrigues(w).T @ t) / DSCALE / mid[i]
            else:
                aa[i, j], tt[i, j] = w, t / DSCALE / mid[i]
    return aa.float().cuda(), tt.float().cuda()


@pytest.fixture(scope="module")
def trained_state():
    """device-only pre-fit (supervised depth, supervised pose, self-supervised warm-up); CPU state dicts of the three networks + data"""
    import torch.nn.functional as F
    from datasets.synthetic import synthetic_batch
    from finetune.train_ft_SQLdepth import FinetuneArgs, FinetuneTrainer
    from options import MonodepthOptions
    from trainer import Trainer
    from sqd import nnkernels
    sys.path.insert(0, REPO)
    torch.manual_seed(0)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    args = [a for a in R50_ARGS]
    args[args.index("--batch_size") + 1] = str(PRE_B)
    ft = FinetuneTrainer(MonodepthOptions().parse(args), FinetuneArgs(bs=PRE_B, epochs=1, lr=1e-4), steps_per_epoch=PRE_STEPS)
    pre = []
    for i in range(PRE_NBATCH):
        s = synthetic_batch(PRE_B, R50_H, R50_W, start=PRE_B * i, scene="drive", with_gt=True, device="cuda")
        pre.append({"image": s[("color_aug", 0, 0)], "depth": F.interpolate(s["depth_gt"], [R50_H, R50_W], mode="nearest") / DSCALE})
    ft.model.train()
    for i in range(PRE_STEPS):
        ft.train_step(pre[i % PRE_NBATCH])
    torch.cuda.synchronize()
    state = {"encoder": {k: v.detach().clone() for k, v in ft.model.encoder.state_dict().items()},
             "depth": {k: v.detach().clone() for k, v in ft.model.depth_decoder.state_dict().items()}}
    del ft, pre
    full = [synthetic_batch(R50_B, R50_H, R50_W, start=1000 + R50_B * i, scene="drive", with_gt=True) for i in range(CMP_NBATCH)]
    batches = [{k: v for k, v in b.items() if k != "depth_gt"} for b in full]
    held = synthetic_batch(4, R50_H, R50_W, start=10 ** 5, with_gt=True, scene="drive")
    # the pose network on the known motions
    nnkernels.reset_plans()
    trp = Trainer(MonodepthOptions().parse(R50_ARGS + ["--sqd_no_graph", "--sqd_no_conv_tune"]))
    pose = trp.models["pose"]
    pose.train()
    popt = torch.optim.Adam(pose.parameters(), 5e-4)
    targets = [_pose_targets(1000 + R50_B * i, R50_B, full[i]["depth_gt"]) for i in range(CMP_NBATCH)]
    for it in range(POSE_STEPS):
        b = batches[it % CMP_NBATCH]
        aug = {f: b[("color_aug", f, 0)].cuda() for f in (0, -1, 1)}
        nnkernels.begin_step()
        aa, tr_ = pose.forward_pairs([(aug[-1], aug[0]), (aug[0], aug[1])])
        ta, tt_ = targets[it % CMP_NBATCH]
        loss = ((aa.reshape(R50_B, 2, 3) - ta) ** 2).mean() + ((tr_.reshape(R50_B, 2, 3) - tt_) ** 2).mean()
        popt.zero_grad()
        loss.backward()
        popt.step()
    pose_mse = float(loss.detach())
    state["pose"] = {k: v.detach().clone() for k, v in pose.state_dict().items()}
    del trp
    # self-supervised warm-up at the reference's learning rate (replayed hipGraph, timed plans)
    nnkernels.reset_plans()
    tr = Trainer(MonodepthOptions().parse(R50_ARGS))
    tr.set_train()
    _no_dropout(tr.models.values())
    for name, sd in state.items():
        tr.models[name].load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    try:
        for i in range(WARM_STEPS):
            dev = {k: v.cuda() for k, v in batches[i % CMP_NBATCH].items()}
            dev[("noise", 0)] = torch.randn(R50_B, 2, R50_H, R50_W, generator=g).cuda()
            tr.train_step(dev)
        torch.cuda.synchronize()
        warm = {n: {k: v.detach().cpu().clone() for k, v in tr.models[n].state_dict().items()} for n in ("encoder", "depth", "pose")}
    finally:
        nnkernels.reset_plans()
    return {"state": warm, "batches": batches, "held": held, "pose_mse": pose_mse}


def _oracle_metrics(O, enc, dep, held):
    for m in (enc, dep):
        m.eval()
    with torch.no_grad():
        out = dep(enc(held[("color_aug", 0, 0)]))
        depth = torch.nn.functional.interpolate(out[("disp", 0)], [R50_H, R50_W], mode="bilinear", align_corners=False)
    for m in (enc, dep):
        m.train()
    return [float(v) for v in O.compute_depth_losses(depth, held["depth_gt"])]


def _oracle_stretch(O, T, noises, perturb):
    enc, dep, pose = O.ResnetEncoderDecoder(50, 256, 32), O.QueryTrDecoder(32, 32, 16, 4, 64, 64, min_val=0.001, max_val=80.0, dim_feedforward=1024, dropout=0.0), O.PoseCNN(2)
    for m, n in ((enc, "encoder"), (dep, "depth"), (pose, "pose")):
        m.load_state_dict(T["state"][n])
        m.train()
    if perturb:
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for q in list(enc.parameters()) + list(dep.parameters()) + list(pose.parameters()):
                q.mul_(1 + perturb * torch.randn(q.shape, generator=g))
    start = _oracle_metrics(O, enc, dep, T["held"])
    ref = O.RefTrainStep(enc, dep, pose, (0, -1, 1), R50_H, R50_W)                  # lr 1e-4: the reference's default
    losses = [float(ref.step(dict(T["batches"][i % CMP_NBATCH]), noises[i])[1]["loss"].detach()) for i in range(CMP_STEPS)]
    return start, _oracle_metrics(O, enc, dep, T["held"]), losses


_ABSREL_RUNS = {}
_ABSREL_SEEDS = [11, 12, 13]
_ORACLE_STRETCH = {}


def _oracle_pair(O, T, seed):
    """the oracle's stretch from the trained state under the seed's tie-break noise, and the same from weights perturbed by 1e-7 (its own
    sensitivity) — CPU only, deterministic, computed once per seed"""
    if seed not in _ORACLE_STRETCH:
        g = torch.Generator().manual_seed(seed)
        noises = [torch.randn(R50_B, 2, R50_H, R50_W, generator=g) for _ in range(CMP_STEPS)]
        _ORACLE_STRETCH[seed] = (noises, _oracle_stretch(O, T, noises, 0.0), _oracle_stretch(O, T, noises, 1e-7))
    return _ORACLE_STRETCH[seed]


@pytest.mark.parametrize("seed", _ABSREL_SEEDS)
def test_abs_rel_resnet50_from_trained_weights(trained_state, seed):
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from options import MonodepthOptions
    from trainer import Trainer
    from sqd import nnkernels
    T = trained_state
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    noises, (start, want, ref_loss), (_, want_p, ref_loss_p) = _oracle_pair(O, T, seed)
    floor = abs(want[0] - want_p[0])
    # the bound of EVERY seed is three times the oracle's worst sensitivity over ALL seeds (known before any device run: whichever seed runs
    # first is held to the same bound; until round 6's last session the first seed was held to its own floor alone, and a box whose plan
    # timing picked another arithmetic for a few layers — the device trajectory follows the plans — could leave it)
    worst_floor = max(abs(w[0] - wp[0]) for _, (_, w, _), (_, wp, _) in (_oracle_pair(O, T, sd) for sd in _ABSREL_SEEDS))
    assert start[0] < 0.25, start                       # the pre-fit and the warm-up left a trained model
    nnkernels.reset_plans()
    tr = Trainer(MonodepthOptions().parse(R50_ARGS))    # --learning_rate at its default 1e-4; plan timing ON, graph replay ON: what bench.py runs
    tr.set_train()
    _no_dropout(tr.models.values())
    for n in ("encoder", "depth", "pose"):
        tr.models[n].load_state_dict(T["state"][n])
    dev_loss = []
    try:
        for i in range(CMP_STEPS):
            dev = {k: v.cuda() for k, v in T["batches"][i % CMP_NBATCH].items()}
            dev[("noise", 0)] = noises[i].cuda()
            dev_loss.append(float(tr.train_step(dev)[1]["loss"].detach()))
        assert tr._graph is not None
        tr.set_eval()
        with torch.no_grad():
            inputs = {k: v.cuda() for k, v in T["held"].items()}
            outputs, losses = tr.process_batch(inputs)
            tr.compute_depth_losses(inputs, outputs, losses)
        got = [float(losses[n]) for n in tr.depth_metric_names]
    finally:
        nnkernels.reset_plans()
    _ABSREL_RUNS[seed] = (abs(got[0] - want[0]), floor)
    print("ResNet-50 192x640 from trained weights, seed %d (pose pre-fit mse %.1e; abs_rel %.4f at the start of the stretch): after %d self-supervised "
          "steps at lr 1e-4: abs_rel device %.5f oracle %.5f perturbed oracle %.5f -> |device - oracle| %.2e, oracle's own sensitivity %.2e; "
          "first-step loss device %.7f oracle %.7f; last-step loss device %.6f oracle %.6f perturbed oracle %.6f"
          % (seed, T["pose_mse"], start[0], CMP_STEPS, got[0], want[0], want_p[0], abs(got[0] - want[0]), floor, dev_loss[0], ref_loss[0],
             dev_loss[-1], ref_loss[-1], ref_loss_p[-1]))
    assert want[0] < 0.25, want                          # the comparison happens where the metric measures the network
    # identical weights: the first step's loss at 2e-5 relative or 2e-6 absolute — at this trained state the loss is ~0.006, a mean of
    # per-pixel (1 - SSIM) / 2 and |difference| terms that are differences of O(1) numbers: fp32 leaves ~1e-6 of absolute error in it
    # (measured 8e-7; 3e-7 at the loss ~0.1 of the untrained full-step tests)
    assert abs(dev_loss[0] - ref_loss[0]) <= max(2e-5 * abs(ref_loss[0]), 2e-6), (dev_loss[0], ref_loss[0])
    assert abs(got[0] - want[0]) <= max(1e-3, 3.0 * worst_floor), ("abs_rel", got[0], want[0], want_p[0], worst_floor)
