"""The hipGraph replay of the training step (Trainer._capture) must train exactly like the eager step:
same batches, same host-drawn tie-break noise, dropout off -> parameters after 7 steps agree."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ARGS = ["--backbone", "resnet18_lite", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
        "--height", "64", "--width", "96", "--batch_size", "2", "--num_workers", "0", "--sqd_synthetic",
        "--log_dir", "/tmp/sqd_graph_test", "--max_depth", "80.0", "--scheduler_step_size", "1", "--sqd_no_conv_tune"]


def run(extra, steps=7):
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    torch.manual_seed(0)
    tr = Trainer(MonodepthOptions().parse(ARGS + extra))
    tr.set_train()
    for m in tr.models.values():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
    losses = []
    g = torch.Generator().manual_seed(5)
    for i in range(steps):
        inputs = synthetic_batch(2, 64, 96, tr.opt.frame_ids, start=2 * i, device=tr.device)       # (with --use_stereo: the "s" frame and stereo_T)
        inputs[("noise", 0)] = torch.randn(2, tr._identity_planes(), 64, 96, generator=g).cuda()
        _, ls = tr.train_step(inputs)
        losses.append(float(ls["loss"]))
        if i == 4:
            tr.model_lr_scheduler.step()          # the learning rate changes between replays
    torch.cuda.synchronize()
    params = {n + "." + k: v.detach().clone() for n, m in tr.models.items() for k, v in m.state_dict().items()}
    return tr, losses, params


def _drift(pa, pb):
    """worst over the state-dict tensors of max|a-b| / (max|a| + 1e-3)"""
    worst = ("", 0.0)
    for k in pa:
        a, b = pa[k].float(), pb[k].float()
        d = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-3)
        if d > worst[1]:
            worst = (k, d)
    return worst


def test_graph_replay_matches_eager():
    tr_e, loss_e, par_e = run(["--sqd_no_graph"])
    _, loss_e2, par_e2 = run(["--sqd_no_graph"])
    tr_g, loss_g, par_g = run([])
    assert tr_e._graph is None and tr_g._graph is not None, "the third run must have replayed a captured graph"
    for a, b in zip(loss_e, loss_g):
        assert abs(a - b) <= 1e-5 * abs(a) + 1e-7, (loss_e, loss_g)
    # Two eager runs already differ (ATen's max-pool backward accumulates with atomics, and Adam's m/sqrt(v) turns a
    # last-bit gradient difference of a zero-initialised bias into a step of ~lr): the replay must stay within that
    # envelope.  A replay bug (frozen bias correction, stale learning rate) shows up as a drift of ~0.2.
    k_ee, d_ee = _drift(par_e, par_e2)
    k_eg, d_eg = _drift(par_e, par_g)
    assert d_eg <= max(5.0 * d_ee, 2e-2) and d_eg < 0.05, (k_eg, d_eg, k_ee, d_ee)
    # Adam bookkeeping kept in step: torch.optim state_dict compatibility
    st_e = tr_e.model_optimizer.state_dict()["state"]
    st_g = tr_g.model_optimizer.state_dict()["state"]
    steps_e, steps_g = [float(v["step"]) for v in st_e.values() if "step" in v], [float(v["step"]) for v in st_g.values() if "step" in v]
    assert steps_e == steps_g and set(steps_e) == {7.0}
    # the step-dependent scalars the captured Adam kernel reads were refreshed for step 7 at the decayed learning rate
    lr = tr_g.model_optimizer.param_groups[0]["lr"]
    assert abs(lr - 1e-5) < 1e-12                                   # StepLR(step_size=1, gamma=0.1) stepped once
    hyper = tr_g.model_optimizer._graph_hyper[0][1].cpu()
    assert abs(float(hyper[0]) - lr / (1 - 0.9 ** 7)) <= 1e-6 * lr and abs(float(hyper[1]) - (1 - 0.999 ** 7) ** -0.5) <= 1e-4


def test_identity_maps_late_or_early_in_the_captured_step():
    """The captured step evaluates the identity-reprojection maps right before the fused warp + SSIM kernel (default) or at the start of
    the step (--sqd_early_identity): the same kernels on the same inputs in another order — the same losses and parameters, bit for bit."""
    tr_l, loss_l, par_l = run([])
    tr_e, loss_e, par_e = run(["--sqd_early_identity"])
    assert tr_l._graph is not None and tr_e._graph is not None
    assert loss_l == loss_e, (loss_l, loss_e)
    assert _drift(par_l, par_e)[1] == 0.0, _drift(par_l, par_e)


def test_graph_replay_trains_every_parameter():
    """Every parameter that receives a gradient must keep moving under graph replay — a tensor derived from a parameter
    outside the captured region (a cached regrouped stem filter, say) would freeze that parameter's effect on the loss."""
    tr, _, _ = run([], steps=4)                       # 3 eager warm-up steps + the capture step
    assert tr._graph is not None
    from datasets.synthetic import synthetic_batch
    snap = {n: p.detach().clone() for m in tr.models.values() for n, p in m.named_parameters() if p.grad is not None}
    inputs = synthetic_batch(2, 64, 96, start=40, device=tr.device)
    inputs[("noise", 0)] = torch.randn(2, 2, 64, 96).cuda()
    _, l1 = tr.train_step(inputs)
    loss_a = float(l1["loss"])
    # the same batch again: the loss must change because EVERY weight changed — in particular the stems' regrouped filters
    _, l2 = tr.train_step(inputs)
    loss_b = float(l2["loss"])
    torch.cuda.synchronize()
    assert loss_a != loss_b
    moved = {n: bool((p.detach() != snap[n]).any()) for m in tr.models.values() for n, p in m.named_parameters() if n in snap}
    assert all(moved.values()), [n for n, v in moved.items() if not v]
    # and the stem filter the forward uses is the CURRENT one: perturbing conv1.weight in place changes the replayed loss
    for stem in (tr.models["encoder"].encoder.encoder.conv1.weight, tr.models["pose"].net[0].weight):
        before = float(tr.train_step(inputs)[1]["loss"])       # (graph replay hands back the same static tensors: read now)
        with torch.no_grad():
            stem.add_(0.2 * torch.randn_like(stem))            # (not a rescaling: BatchNorm would undo that)
        after = float(tr.train_step(inputs)[1]["loss"])
        assert abs(after - before) > 1e-3 * abs(before), (before, after)


def test_captured_step_keeps_its_source_frames_channels_last():
    """the static source frames of the captured step are channels_last tensors (the photometric kernels' cheaper layout, trainer._capture):
    same shape and values for a reader of `inputs` / `outputs`, written by the copy-in of every batch; the eager run keeps planar frames; the two
    train alike (test_graph_replay_matches_eager), and configurations the channels_last kernels do not serve keep planar frames in the graph too"""
    from datasets.synthetic import synthetic_batch
    tr, _, _ = run([], steps=5)
    assert tr._graph is not None and sorted(tr._hwc_keys) == [("color", -1, 0), ("color", 1, 0)]
    batch = synthetic_batch(2, 64, 96, tr.opt.frame_ids, start=77, device=tr.device)
    batch[("noise", 0)] = torch.randn(2, 2, 64, 96).cuda()
    want = {k: v.clone() for k, v in batch.items()}
    outputs, _ = tr.train_step(batch)
    for k, v in want.items():
        st = tr._static_in[k]
        assert batch[k] is st and st.shape == v.shape and torch.equal(st, v), k
        cl = not st.is_contiguous() and st.is_contiguous(memory_format=torch.channels_last)
        assert cl == (k in tr._hwc_keys), (k, st.stride())
    for f in (-1, 1):
        assert torch.equal(outputs[("color_identity", f, 0)], want[("color", f, 0)])
    # a host batch (a loader's) takes the same route
    host = {k: v.cpu() for k, v in want.items()}
    tr.train_step(host)
    assert all(torch.equal(tr._static_in[k], want[k]) for k in want)
    tr_s, _, _ = run(["--no_ssim"], steps=5)          # a loss option: round 5's forward kernel, planar frames
    assert tr_s._graph is not None and tr_s._hwc_keys == []
    assert all(t.is_contiguous() for k, t in tr_s._static_in.items() if k[0] == "color")


def test_tuned_plans_train_like_the_default_plans():
    """The configuration that is benchmarked — first-step plan timing on: per layer the fastest of the fp32, three-term bf16,
    input-patch (3x3 and stems) and weight-gradient variants — must train like the library's default plans: every plan is an fp32-level
    evaluation of the same convolution, so the losses of 7 steps agree to 1e-4 and the parameters stay inside the run-to-run envelope.
    ResNet-50 at 96x160 (every stage keeps at least 3x5 pixels), batch 3."""
    global ARGS
    saved = ARGS
    base = [a for a in ARGS if a != "--sqd_no_conv_tune"]
    for key, val in (("--backbone", "resnet"), ("--height", "96"), ("--width", "160"), ("--batch_size", "2")):
        base[base.index(key) + 1] = val
    base += ["--num_layers", "50", "--num_features", "64"]

    def run50(extra, nsteps=5):
        from options import MonodepthOptions
        from trainer import Trainer
        from datasets.synthetic import synthetic_batch
        from sqd import nnkernels
        torch.manual_seed(0)
        tr = Trainer(MonodepthOptions().parse(base + extra))
        tr.set_train()
        for m in tr.models.values():
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.0
        losses = []
        g = torch.Generator().manual_seed(5)
        for i in range(nsteps):
            inputs = synthetic_batch(2, 96, 160, start=2 * i, device=tr.device)
            inputs[("noise", 0)] = torch.randn(2, 2, 96, 160, generator=g).cuda()
            _, ls = tr.train_step(inputs)
            losses.append(float(ls["loss"]))
        torch.cuda.synchronize()
        params = {n + "." + k: v.detach().clone() for n, m in tr.models.items() for k, v in m.state_dict().items()}
        return losses, params, set(nnkernels._TUNED)
    try:
        loss_d, par_d, _ = run50(["--sqd_no_graph", "--sqd_no_conv_tune"])
        loss_d2, par_d2, _ = run50(["--sqd_no_graph", "--sqd_no_conv_tune"])
        _, par1_d, _ = run50(["--sqd_no_graph", "--sqd_no_conv_tune"], 1)
        loss_t, par_t, tuned = run50(["--sqd_no_graph"])
        _, par1_t, _ = run50(["--sqd_no_graph"], 1)              # (plans stay registered from the run above)
    finally:
        ARGS = saved
        from sqd import nnkernels
        nnkernels.reset_plans()                            # leave no measured plan behind for the tests that follow
    assert len(tuned) > 40, "the tuned run must have timed its layers"
    for a, b in zip(loss_d, loss_t):
        assert abs(a - b) <= 1e-4 * abs(a) + 1e-6, (loss_d, loss_t)
    assert _drift(par_d, par_d2)[1] <= 1e-6                      # the default plans are deterministic
    # The state after ONE step separates a wrong kernel from amplification (after a few steps Adam's m / sqrt(v) has turned last-bit
    # gradient differences on zero-initialised biases into whole steps of lr, and the 3x5-pixel BatchNorm statistics of layer 4 follow):
    # the BatchNorm running statistics of the first forward pass depend on the forward kernels alone
    for k in par1_d:
        if k.endswith("running_mean") or k.endswith("running_var"):
            a, b = par1_d[k].double(), par1_t[k].double()
            assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()) + 1e-7, k       # fp32 rounding through up to 53 layers


@pytest.mark.parametrize("flags", [["--no_ssim"], ["--avg_reprojection"], ["--disable_automasking"],
                                   ["--no_ssim", "--avg_reprojection", "--disable_automasking"], ["--avg_reprojection", "--use_stereo"]])
def test_loss_options_train_and_match_the_oracle(flags):
    """--no_ssim / --avg_reprojection / --disable_automasking (reference options.py, trainer.py:447-451, 480-524) at Trainer level:
    the graph replay follows the eager step, and the loss the step reports is the oracle's compute_losses of the step's own
    disparity and warped images under the same options."""
    from oracle import torch_ref as O
    from datasets.synthetic import synthetic_batch
    tr_e, loss_e, _ = run(["--sqd_no_graph"] + flags, steps=5)
    tr_g, loss_g, _ = run(flags, steps=5)
    assert tr_e._graph is None and tr_g._graph is not None
    for a, b in zip(loss_e, loss_g):
        assert abs(a - b) <= 1e-5 * abs(a) + 1e-7, (loss_e, loss_g)
    avg, noauto = "--avg_reprojection" in flags, "--disable_automasking" in flags
    stereo = "--use_stereo" in flags                  # (three source frames: the mean of --avg_reprojection runs over all of them)
    fids = [0, -1, 1] + (["s"] if stereo else [])
    inputs = synthetic_batch(2, 64, 96, fids, start=40, device=tr_e.device)
    noise = torch.randn(2, 1 if avg else len(fids) - 1, 64, 96, generator=torch.Generator().manual_seed(9))
    inputs[("noise", 0)] = noise.cuda()
    tr_e.set_eval()
    with torch.no_grad():
        outputs, losses = tr_e.process_batch(inputs)
    assert ("identity_selection/0" in outputs) == (not noauto)
    cpu = lambda t: t.detach().float().cpu()
    want = O.compute_losses(cpu(outputs[("disp", 0)]), cpu(inputs[("color", 0, 0)]), {f: cpu(outputs[("color", f, 0)]) for f in fids[1:]},
                            {f: cpu(inputs[("color", f, 0)]) for f in fids[1:]}, fids, noise, 64, 96,
                            disparity_smoothness=tr_e.opt.disparity_smoothness, no_ssim="--no_ssim" in flags, avg_reprojection=avg,
                            disable_automasking=noauto)
    got, ref = float(losses["loss"]), float(want["loss"])
    assert abs(got - ref) <= 1e-4 * abs(ref), (got, ref)


def test_replay_copies_every_batch_into_the_graph_inputs():
    """Every replayed step copies its batch into the graph's static input tensors — also a device tensor that is fed again, whose contents
    may have been rewritten through a raw pointer without touching its autograd version counter (round 3 skipped that copy: ADVICE r03)."""
    from datasets.synthetic import synthetic_batch
    tr, _, _ = run([], steps=4)                        # captured
    assert tr._graph is not None
    a = synthetic_batch(2, 64, 96, start=100, device=tr.device)
    a[("noise", 0)] = torch.randn(2, 2, 64, 96, device=tr.device)
    b = synthetic_batch(2, 64, 96, start=200, device=tr.device)
    b[("noise", 0)] = a[("noise", 0)].clone()
    key = ("color", 0, 0)
    tr.train_step(dict(a))
    assert torch.equal(tr._static_in[key], a[key])
    tr.train_step(dict(a))                             # the same objects again
    assert torch.equal(tr._static_in[key], a[key])
    v = a[key]._version
    a[key].data.view(-1)[:16].fill_(0.25)              # (a write that leaves the version counter of a[key] alone)
    assert a[key]._version == v and not torch.equal(tr._static_in[key], a[key])
    tr.train_step(dict(a))
    assert torch.equal(tr._static_in[key], a[key])     # ... and the replay saw it
    host = {k: t.cpu() for k, t in b.items()}          # a loader's host batch takes the one-by-one path
    tr.train_step(dict(host))
    assert all(torch.equal(tr._static_in[k].cpu(), host[k]) for k in host)
    tr.train_step(dict(b))
    assert torch.equal(tr._static_in[key], b[key])
