"""Convolution plans as a pinned, exportable part of a run (VERDICT r02 "no way to pin or export the chosen plan set"), the
precision-aware plan table (under --sqd_bf16 a registered input-patch plan made BatchNorm read partial rows the kernel
never wrote) and the EfficientNet stem on the space-to-depth path."""
import json
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ARGS = ["--backbone", "resnet18_lite", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
        "--height", "64", "--width", "96", "--batch_size", "2", "--num_workers", "0", "--sqd_synthetic",
        "--log_dir", "/tmp/sqd_plans_test", "--max_depth", "80.0", "--sqd_no_graph"]


def _train(extra, steps=6):
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
    g = torch.Generator().manual_seed(5)
    losses = []
    for i in range(steps):
        inputs = synthetic_batch(2, 64, 96, start=2 * i, device=tr.device)
        inputs[("noise", 0)] = torch.randn(2, 2, 64, 96, generator=g).cuda()
        losses.append(float(tr.train_step(inputs)[1]["loss"].detach()))
    torch.cuda.synchronize()
    w = {n + "." + k: v.detach().clone() for n, m in tr.models.items() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    return losses, w


def test_exported_plans_pin_a_run_bit_for_bit(tmp_path):
    """run A times its layers and writes the plan file; run B loads it, times nothing, and trains to the same bits"""
    from sqd import nnkernels
    path = str(tmp_path / "plans.json")
    nnkernels.reset_plans()
    try:
        loss_a, w_a = _train(["--sqd_save_conv_plans", path])
        chosen_a = dict(nnkernels.CHOSEN_PLANS)
        rec = json.load(open(path))
        assert rec["precision"] == 0 and len(rec["plans"]) == len(chosen_a) > 30
        assert {e["pass"] for e in rec["plans"]} == {"fwd", "dgrad", "wgrad"}
        nnkernels.reset_plans()
        timed = []
        orig = nnkernels._time_launch
        nnkernels._time_launch = lambda launch, arg: (timed.append(1), orig(launch, arg))[1]
        try:
            loss_b, w_b = _train(["--sqd_conv_plans", path])
        finally:
            nnkernels._time_launch = orig
        assert not timed, "a pinned run must not time any plan"
        assert dict(nnkernels.CHOSEN_PLANS) == chosen_a
        assert loss_a == loss_b, (loss_a, loss_b)
        # (ATen's max-pool backward is not in this network's path; every kernel of the step is fixed-order: bit-equal weights)
        worst = max(float((w_a[k] - w_b[k]).abs().max()) for k in w_a)
        assert worst == 0.0, worst
    finally:
        nnkernels.reset_plans()


def test_plan_file_of_another_precision_is_refused(tmp_path):
    from sqd import nnkernels
    path = str(tmp_path / "plans.json")
    json.dump({"abi": 1, "precision": 2, "plans": []}, open(path, "w"))
    from options import MonodepthOptions
    from trainer import Trainer
    with pytest.raises(RuntimeError, match="convolution precision"):
        Trainer(MonodepthOptions().parse(ARGS + ["--sqd_conv_plans", path]))
    nnkernels.reset_plans()


def test_bf16_mode_has_no_fp32_only_plans_and_reports_the_rows_it_writes():
    """--sqd_bf16 runs one kernel family.  (1) the fp32-only variants (single-buffered, three-term, input-patch) are refused there;
    (2) switching the arithmetic drops plans registered under the other one; (3) with plan timing ON, conv -> BatchNorm on a
    geometry whose height is no multiple of the patch rows (20 x 64: an input-patch plan would report 12 N partial rows where the
    GEMM kernel writes 10 N) gives the batch statistics of the bf16-operand convolution."""
    from sqd import lib, nnkernels, nnops
    L = lib.lib()
    N, C, H, W, K = 2, 32, 20, 64, 32
    geom = (N, H, W, C, K, 3, 3, 1, 1, H, W)
    nnkernels.reset_plans()
    nnkernels.set_conv_precision(0)
    assert L.sqd_conv_set_plan(0, *geom, 128, 32, 1, 3104) == 0           # fp32 arithmetic: the input-patch plan exists
    assert L.sqd_conv_fwd_stats_rows(*geom) == N * 3 * 4                   # ceil(20/8) x ceil(64/16) patches per image
    nnkernels.set_conv_precision(2)
    try:
        # the table was dropped: the cost model's GEMM tile (or a split reduction, which writes no partials), not 12 N patch rows
        assert L.sqd_conv_fwd_stats_rows(*geom) in (0, (N * H * W + 127) // 128, (N * H * W + 63) // 64)
        for bk in (3104, 1056, 528, 272):
            assert L.sqd_conv_set_plan(0, *geom, 128, 32, 1, bk) != 0, bk
        assert b"fp32 arithmetic only" in L.sqd_last_error()
        torch.manual_seed(3)
        conv, bn = nn.Conv2d(C, K, 3, 1, 1, bias=False), nn.BatchNorm2d(K)
        x = torch.randn(N, C, H, W)
        rb = lambda t: t.bfloat16().double()
        y_ref = F.conv2d(rb(x), rb(conv.weight.detach()), None, 1, 1)
        mean_ref, var_ref = y_ref.mean((0, 2, 3)), y_ref.var((0, 2, 3), unbiased=True)
        conv_g, bn_g = conv.cuda().to(memory_format=torch.channels_last), bn.cuda()
        bn_g.train()
        saved = nnkernels.TUNE_CONV
        nnkernels.TUNE_CONV = True
        try:
            for _ in range(2):           # first call times the plans, second runs the registered one
                bn_g.running_mean.zero_(), bn_g.running_var.fill_(1.0)
                y = nnops.conv_bn_act(x.cuda().contiguous(memory_format=torch.channels_last), conv_g, bn_g, "relu")
                torch.cuda.synchronize()
                assert ("fwd",) + geom in nnkernels.CHOSEN_PLANS and not (nnkernels.CHOSEN_PLANS[("fwd",) + geom][3] & (1024 | 2048 | 512))
                got_mean, got_var = bn_g.running_mean.cpu().double() / 0.1, (bn_g.running_var.cpu().double() - 0.9) / 0.1
                assert torch.allclose(got_mean, mean_ref, rtol=1e-4, atol=1e-5), float((got_mean - mean_ref).abs().max())
                assert torch.allclose(got_var, var_ref, rtol=1e-4, atol=1e-5), float((got_var - var_ref).abs().max())
                y_want = F.relu((y_ref - mean_ref[None, :, None, None]) / (y_ref.var((0, 2, 3), unbiased=False) + 1e-5).sqrt()[None, :, None, None])
                assert float((y.cpu().double() - y_want).abs().max()) <= 2e-4
        finally:
            nnkernels.TUNE_CONV = saved
    finally:
        nnkernels.set_conv_precision(0)
        nnkernels.reset_plans()


@pytest.mark.parametrize("N,H,W,K", [(2, 64, 96, 48), (1, 30, 52, 16)])
def test_efficientnet_stem_on_the_space_to_depth_path(N, H, W, K):
    """3x3 / stride 2 / TensorFlow-SAME on the 3-channel frame == 3x3 / stride 1 / pad 1 on the space-to-depth(2) image with the
    scattered filter: output and filter gradient against float64 (reference networks/base_encoder.py:41,94: conv_stem)."""
    from sqd import nnkernels
    torch.manual_seed(K)
    conv = nn.Conv2d(3, K, 3, 2, 0, bias=False)
    x = torch.randn(N, 3, H, W)
    wd = conv.weight.detach().double().requires_grad_(True)
    y_ref = F.conv2d(F.pad(x.double(), (0, 1, 0, 1)), wd, None, 2)        # even sizes: SAME pads below / right only
    g = torch.randn_like(y_ref)
    (y_ref * g).sum().backward()
    conv_g = conv.cuda()
    y = nnkernels.conv2d_stem3_same_s2d(x.cuda(), conv_g)
    assert y.shape == y_ref.shape
    (y * g.float().cuda()).sum().backward()
    assert float((y.detach().cpu().double() - y_ref.detach()).abs().max()) <= 1e-5 * float(y_ref.abs().max())
    assert float((conv_g.weight.grad.cpu().double() - wd.grad).abs().max()) <= 2e-5 * float(wd.grad.abs().max())
    with pytest.raises(RuntimeError, match="even-sized"):
        nnkernels.conv2d_stem3_same_s2d(torch.randn(1, 3, 31, 52).cuda(), conv_g)
