"""End-to-end parity of one training step at BASELINE.json's flagship configurations — ResNet-50 + Depth_Decoder_QueryTr at
192x640 (configs[1]) and at 320x1024 (configs[2]: 320 patch tokens, the encoder path beyond the fused attention's 128) — against
the oracle restatement on the host: same weights, same batch, same tie-break noise, dropout off.  Batch 2 / 1 keeps the oracle's
CPU step in seconds; every kernel still runs at the full image size."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADS = {
    # kind: (network flags, oracle decoder arguments (patch, Q, dim_out, ff, min_depth))
    "res50": (["--backbone", "resnet", "--num_layers", "50", "--num_features", "256", "--query_nums", "64", "--dim_out", "64",
               "--patch_size", "16", "--min_depth", "0.001"], (16, 64, 64, 1024, 0.001)),
    # configs[2] with the head of the reference's own 320x1024 file (args_files/hisfog/kitti/resnet_320x1024.txt:6,12-19):
    # backbone resnet_lite (=> Lite_Depth_Decoder_QueryTr, feed-forward 512), patch 20, 128 queries, dim_out 128, min_depth 0.01
    "res50_c": (["--backbone", "resnet_lite", "--num_layers", "50", "--num_features", "256", "--query_nums", "128", "--dim_out", "128",
                 "--patch_size", "20", "--min_depth", "0.01"], (20, 128, 128, 512, 0.01)),
    # config B' = the old args_files/args_res50_kitti_192x640_train.txt:8-10 (model_dim 64, patch 16, 120 queries; num_features 512 and
    # dim_out 128 are the parser's defaults): the 64-wide patch-token encoder and Self Query Layer
    "res50_bp": (["--backbone", "resnet", "--num_layers", "50", "--num_features", "512", "--query_nums", "120", "--dim_out", "128",
                  "--patch_size", "16", "--min_depth", "0.001"], (16, 120, 128, 1024, 0.001)),
    "res18": (["--backbone", "resnet18_lite", "--query_nums", "120", "--dim_out", "128", "--patch_size", "16", "--min_depth", "0.001"],
              (16, 120, 128, 512, 0.001)),       # args_res18_kitti_192x640_tarin.txt:8-10
}


_STEM_REF = {}        # (H, W, B, kind) -> (float64 gradient of the encoder's stem filter, the oracle's own fp32 error against it)


def _stem_noise_floor(key, nets, cpu_inputs, noise, H, W):
    """The stem filter's gradient is the end of every gradient path of the trunk: two correct evaluations that round differently
    disagree on a handful of ReLU / max-pool gates among ~10^8 activations, and each flipped gate re-routes a path that ends here.
    How much that is gets MEASURED, per configuration: the oracle step in float64 (the reference value), and three float32 evaluations
    of the same oracle — as it is, and twice with every weight perturbed by 1e-7 relative (further draws of the rounding lottery).  Returns
    (g64, floor_max, floor_l2) with floor_* the largest of the three fp32 errors against float64 (relative to max|g| / to ||g||)."""
    if key in _STEM_REF:
        return _STEM_REF[key]
    import copy
    from oracle import torch_ref as O

    def grad(dtype, perturb, seed=1):
        e, d, p = [copy.deepcopy(m).to(dtype) for m in nets]
        if perturb:
            g = torch.Generator().manual_seed(seed)
            with torch.no_grad():
                for q in list(e.parameters()) + list(d.parameters()) + list(p.parameters()):
                    q.mul_(1 + perturb * torch.randn(q.shape, generator=g).to(dtype))
        step = O.RefTrainStep(e, d, p, (0, -1, 1), H, W)
        step.step({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in cpu_inputs.items()}, noise.to(dtype))
        name = "encoder.encoder.conv1.weight"
        return dict(e.named_parameters())[name].grad.double()
    g64 = grad(torch.float64, 0.0)
    errs = []
    for perturb, seed in ((0.0, 0), (1e-7, 1), (1e-7, 2)):          # three draws of the rounding lottery
        g32 = grad(torch.float32, perturb, seed)
        errs.append((float((g32 - g64).abs().max() / g64.abs().max()), float((g32 - g64).norm() / g64.norm())))
    _STEM_REF[key] = (g64, max(e[0] for e in errs), max(e[1] for e in errs), errs)
    return _STEM_REF[key]


def _args(H, W, B, extra, kind="res50"):
    return HEADS[kind][0] + ["--model_dim", "64" if kind == "res50_bp" else "32", "--height", str(H), "--width", str(W), "--batch_size", str(B),
                             "--max_depth", "80.0", "--num_workers", "0", "--sqd_synthetic", "--log_dir", "/tmp/sqd_full_cfg_test"] + extra


@pytest.mark.parametrize("plans", ["default_plans", "tuned_plans"])
@pytest.mark.parametrize("H,W,B,kind", [(192, 640, 2, "res50"), (320, 1024, 1, "res50_c"), (192, 640, 2, "res18"), (192, 640, 2, "res50_bp")])
def test_flagship_step_matches_oracle(H, W, B, kind, plans):
    """configs[1] and configs[2] (ResNet-50 + [Lite_]Depth_Decoder_QueryTr) and configs[0] at its real shape (ResNet-18 +
    Lite_Depth_Decoder_QueryTr, 192x640, batch 2, model_dim 32 / patch 16 / 120 queries / dim_out 128), each under the
    library's default (fp32 MFMA) plans and under first-step plan timing — the benchmarked arithmetic: per layer the fastest of the
    fp32, three-term bf16 and input-patch kernels — against the same oracle step at the same 1e-4."""
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    torch.manual_seed(0)
    from sqd import nnkernels
    nnkernels.reset_plans()
    tr = Trainer(MonodepthOptions().parse(_args(H, W, B, ["--sqd_no_graph"] + ([] if plans == "tuned_plans" else ["--sqd_no_conv_tune"]), kind)))
    tr.set_train()
    for m in tr.models.values():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
    patch, Q, dim_out, ff, min_depth = HEADS[kind][1]
    md = 64 if kind == "res50_bp" else 32
    enc = O.LiteResnetEncoderDecoder(model_dim=32) if kind == "res18" else O.ResnetEncoderDecoder(50, 512 if kind == "res50_bp" else 256, md)
    dep = O.QueryTrDecoder(md, md, patch, 4, Q, dim_out, min_val=min_depth, max_val=80.0, dim_feedforward=ff, dropout=0.0)
    pose = O.PoseCNN(2)
    for ref, mine in ((enc, tr.models["encoder"]), (dep, tr.models["depth"]), (pose, tr.models["pose"])):
        ref.load_state_dict({k: v.detach().cpu() for k, v in mine.state_dict().items()})
        ref.train()
    cpu_inputs = synthetic_batch(B, H, W)
    noise = torch.randn(B, 2, H, W)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    stem64, floor_max, floor_l2, floor_samples = _stem_noise_floor((H, W, B, kind), (enc, dep, pose), cpu_inputs, noise, H, W)
    ref = O.RefTrainStep(enc, dep, pose, (0, -1, 1), H, W)
    ref_out, ref_losses = ref.step(dict(cpu_inputs), noise)
    inputs = {k: v.cuda() for k, v in cpu_inputs.items()}
    inputs[("noise", 0)] = noise.cuda()
    from sqd import nnops
    nnops.ATEN_CALLS.clear()
    try:
        outputs, losses = tr.train_step(inputs)
        torch.cuda.synchronize()
        mix = nnkernels.plan_mix()
        assert not nnops.ATEN_CALLS, nnops.ATEN_CALLS          # every operator of the step ran in libsqd (model_dim 64 included)
    finally:
        nnkernels.reset_plans()
    if plans == "tuned_plans":
        assert sum(mix.get("fwd", {}).values()) > 20 and sum(mix.get("wgrad", {}).values()) > 20, mix      # the layers were timed
    got, want = float(losses["loss"]), float(ref_losses["loss"])
    disp, disp_ref = outputs[("disp", 0)].detach().cpu(), ref_out[("disp", 0)].detach()
    d_err = float((disp - disp_ref).abs().max()) / float(disp_ref.abs().max())
    print("%s %dx%d %s: loss %.7f oracle %.7f (rel %.2e), disparity max err %.2e of max; plans %s"
          % (kind, H, W, plans, got, want, abs(got - want) / abs(want), d_err, mix))
    # north_star: 1e-4 relative; measured on MI355X 1-2e-6 for loss and 1.4-2.4e-6 for the disparity, default and measured plans alike
    assert abs(got - want) <= 2e-5 * abs(want), (got, want)
    assert d_err <= 2e-5, "predicted disparity"
    # the optimiser step: Adam moves every weight by ~lr, so compare the UPDATE of a few tensors (first / last layers of each net)
    # (the two 7x7 stems are the non-leaf / regrouped filters of the eager path: ADVICE r1 asked for them to be checked here)
    for net, mine, name in ((pose, tr.models["pose"], "pose_conv.weight"), (enc, tr.models["encoder"], "decoder.conv3.weight"),
                            (dep, tr.models["depth"], "convert_to_prob.0.weight"), (pose, tr.models["pose"], "net.0.weight"),
                            (enc, tr.models["encoder"], "encoder.encoder.conv1.weight"), (pose, tr.models["pose"], "net.3.weight")):
        p_ref = dict(net.named_parameters())[name]
        w_ref, g_ref = p_ref.detach(), p_ref.grad.detach()
        p_got = dict(mine.named_parameters())[name]
        w_got, g_got = p_got.detach().cpu(), p_got.grad.detach().cpu()
        # (1) the gradient itself, relative to the tensor's largest entry.  Measured on MI355X: 1e-4 .. 6e-4 for the heads, PoseCNN and the
        # pose stem, 7.5e-3 .. 8.2e-3 for the encoder's stem at ResNet-50 depth — the same under the default and the measured plans and
        # with round 2's kernels: two fp32 evaluations with different summation orders disagree on a handful of ReLU / max-pool gates
        # among the ~10^8 activations of the trunk, and every flipped gate re-routes a gradient path that ends in this filter
        g_err = float((g_got - g_ref).abs().max()) / float(g_ref.abs().max())
        g_l2 = float((g_got - g_ref).norm()) / float(g_ref.norm())              # (a handful of re-routed paths barely move the norm)
        # (2) the optimiser step.  Adam's first update is lr * g/(|g| + eps) = lr * sign(g): an element whose gradient lies within that
        # noise of zero may legitimately move the other way — allowed only where |g_ref| is small against the tensor's scale, and
        # only for a few elements.
        bad = (w_got - w_ref).abs() > 5e-5
        noise_level = g_ref.abs() <= max(5e-3, 4.0 * g_err) * g_ref.abs().max()
        print("%s: gradient max err %.1e of max|g|, L2 err %.1e; %d of %d updated weights beyond 5e-5, all of them inside the gradient's noise band: %s"
              % (name, g_err, g_l2, int(bad.sum()), bad.numel(), bool((~bad | noise_level).all())))
        if name.endswith("encoder.conv1.weight"):
            # the encoder's stem: held to the float64 value of the oracle step at twice the oracle's OWN measured fp32 noise there
            # (_stem_noise_floor; measured on the host in the build container for res50 192x640: fp32 oracle vs float64 1.26e-2 of max /
            # 1.31e-2 in norm, a float64 run with weights perturbed by 1e-7 still 0.73e-2 — the filter sits at the end of a chaotic map)
            e64 = float((g_got.double() - stem64).abs().max() / stem64.abs().max())
            l64 = float((g_got.double() - stem64).norm() / stem64.norm())
            print("%s: against float64: device max err %.2e / L2 %.2e; the oracle's fp32 evaluations (as is, weights * (1 + 1e-7 n)): %s"
                  % (name, e64, l64, ["max %.2e / L2 %.2e" % e for e in floor_samples]))
            assert e64 <= 2.0 * floor_max, (name, e64, floor_max)
            assert l64 <= 2.0 * floor_l2, (name, l64, floor_l2)
        else:
            assert g_err <= 2e-3, (name, g_err)
            assert g_l2 <= 2e-3, (name, g_l2)
        assert bool((~bad | noise_level).all()), (name, float((w_got - w_ref).abs().max()))
        assert float(bad.float().mean()) <= 2e-2, (name, int(bad.sum()))


def test_bench_configuration_matches_oracle():
    """THE BENCHMARKED STEP: bench.py's CONFIG_B — ResNet-50 + Depth_Decoder_QueryTr, 192x640, batch 12, two sources — with the pinned plan
    set plans/configB_resnet50_192x640_b12.json (130 plans keyed by the batch-12 geometries: two-term fp16 / three-term bf16 / fp32 per
    layer), four steps so that the fourth is a REPLAY of the captured hipGraph, against four steps of the oracle at batch 12 with the same
    weights, batches and tie-break noise (host noise: --sqd_device_noise, the one bench flag left out, draws it on the device).  The
    step's fused warp + SSIM forward is the lean kernel the bench line's roofline block prices."""
    sys.path.insert(0, REPO)
    import bench
    from oracle import torch_ref as O
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    from sqd import nnkernels, nnops
    H, W, B = 192, 640, 12
    torch.manual_seed(0)
    nnkernels.reset_plans()
    plan_path, note = bench.pinned_plans()
    assert plan_path is not None, note                      # a stale plan file must fail here, not silently time other kernels
    args = [a for a in bench.CONFIG_B if a != "--sqd_device_noise"] + ["--sqd_conv_plans", plan_path]
    tr = Trainer(MonodepthOptions().parse(args))
    tr.set_train()
    for m in tr.models.values():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
    enc, dep, pose = O.ResnetEncoderDecoder(50, 256, 32), O.QueryTrDecoder(32, 32, 16, 4, 64, 64, min_val=0.001, max_val=80.0, dim_feedforward=1024, dropout=0.0), O.PoseCNN(2)
    for ref, mine in ((enc, tr.models["encoder"]), (dep, tr.models["depth"]), (pose, tr.models["pose"])):
        ref.load_state_dict({k: v.detach().cpu() for k, v in mine.state_dict().items()})
        ref.train()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ref = O.RefTrainStep(enc, dep, pose, (0, -1, 1), H, W)
    g = torch.Generator().manual_seed(3)
    try:
        for step in range(4):
            cpu_inputs = synthetic_batch(B, H, W, start=B * step)
            noise = torch.randn(B, 2, H, W, generator=g)
            ref_out, ref_losses = ref.step(dict(cpu_inputs), noise)
            inputs = {k: v.cuda() for k, v in cpu_inputs.items()}
            inputs[("noise", 0)] = noise.cuda()
            nnops.ATEN_CALLS.clear()
            outputs, losses = tr.train_step(inputs)
            torch.cuda.synchronize()
            assert not nnops.ATEN_CALLS, nnops.ATEN_CALLS
            got, want = float(losses["loss"]), float(ref_losses["loss"])
            disp, disp_ref = outputs[("disp", 0)].detach().cpu(), ref_out[("disp", 0)].detach()
            d_err = float((disp - disp_ref).abs().max()) / float(disp_ref.abs().max())
            d_l1 = float((disp - disp_ref).abs().mean()) / float(disp_ref.abs().mean())
            print("bench configuration, step %d (%s): loss %.7f oracle %.7f (rel %.2e), disparity max err %.2e of max, mean err %.2e of mean"
                  % (step + 1, tr.graph_mode() if step == 3 else "eager warm-up", got, want, abs(got - want) / abs(want), d_err, d_l1))
            # every step's loss at the full-step tests' bound.  The disparity: step 1 at that bound too; later steps start from weights that
            # differ by Adam's sign noise (its first updates are +-lr whatever |g| is, so an element whose gradient is within rounding of
            # zero may move the other way: test_flagship_step_matches_oracle / G16) — single pixels then differ at the 1e-2 level of the
            # largest disparity (measured 1e-3, 7e-3 after one and two updates) while the map as a whole stays at 1e-4
            assert abs(got - want) <= 2e-5 * abs(want), (step, got, want)
            assert d_err <= (2e-5 if step == 0 else 5e-2), (step, d_err)
            assert d_l1 <= (2e-5 if step == 0 else 1e-3), (step, d_l1)
        assert tr.graph_mode() == "graph" and tr._graph is not None           # step 4 was the replay
        mix = nnkernels.plan_mix()
        import json
        pinned = json.load(open(plan_path))
        print("plan mix of the run:", mix)
        # the run used exactly the pinned set: as many planned geometries per pass as the file holds, none timed live
        for p, n in (("fwd", sum(1 for e in pinned["plans"] if e["pass"] == "fwd")), ("dgrad", sum(1 for e in pinned["plans"] if e["pass"] == "dgrad")),
                     ("wgrad", sum(1 for e in pinned["plans"] if e["pass"] == "wgrad"))):
            assert sum(mix.get(p, {}).values()) == n, (p, mix.get(p), n)
    finally:
        nnkernels.reset_plans()
