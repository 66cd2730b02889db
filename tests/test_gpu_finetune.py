"""GPU parity of the metric-depth finetune step (csrc/finetune.hip, AdamW + clipping of csrc/adam.hip, sfmnext-impl_amd/finetune/)
against the reference's own vectors (G21), torch composites and the oracle's restatement of the step (oracle/finetune_ref.py)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_silog_golden_g21(golden):
    from finetune.loss import SILogLoss
    g = golden("g21_silog")
    depth = torch.from_numpy(g["depth"]).cuda()
    for key, lkey, gkey, interp in (("pred_lr", "loss_interp", "grad_lr", True), ("pred_hr", "loss_plain", "grad_hr", False)):
        p = torch.from_numpy(g[key]).cuda().requires_grad_(True)
        loss = SILogLoss()(p, depth, 1e-3, interpolate=interp)
        loss.backward()
        assert abs(float(loss.detach()) - float(g[lkey])) <= 2e-6 * abs(float(g[lkey])), (key, float(loss), float(g[lkey]))
        want = torch.from_numpy(g[gkey])
        err = float((p.grad.cpu() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()), (key, err)


@pytest.mark.parametrize("h,w,H,W", [(12, 20, 24, 40), (96, 320, 352, 1216), (7, 9, 7, 9), (5, 6, 13, 4), (1, 1, 3, 3)])
def test_resize_align_corners(h, w, H, W):
    from sqd import ops
    torch.manual_seed(h + W)
    x = torch.randn(2, 1, h, w)
    wgt = torch.randn(2, 1, H, W)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, (H, W), mode="bilinear", align_corners=True)
    (yr * wgt.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = ops.ResizeAlignCorners.apply(xg, H, W)
    (y * wgt.cuda()).sum().backward()
    # source coordinates are float32 products (dst * (in-1)/(out-1)), as ATen's upsample_bilinear2d computes them for float input:
    # tight against the float32 composite, looser against the float64 one (coordinate rounding grows along a row)
    y32 = F.interpolate(x, (H, W), mode="bilinear", align_corners=True)
    assert float((y.cpu() - y32).abs().max()) <= 2e-6 * float(y32.abs().max()) + 1e-7
    assert float((y.cpu().double() - yr).abs().max()) <= 1e-4 * float(yr.abs().max()) + 1e-7
    assert float((xg.grad.cpu().double() - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max()) + 1e-7


@pytest.mark.parametrize("quant", [False, True])
@pytest.mark.parametrize("crop", ["garg", "eigen", None])
def test_median_ratio_is_exact(crop, quant):
    from sqd import ops
    rs = np.random.RandomState(4 + quant)
    B, H, W = 5, 88, 304
    depth = rs.uniform(0.5, 90.0, (B, 1, H, W)).astype(np.float32)
    pred = rs.uniform(1.0, 40.0, (B, 1, H, W)).astype(np.float32)
    if quant:                                                       # repeated values around the middle (ties inside the middle pair)
        depth, pred = np.round(depth / 8) * 8 + 1, np.round(pred / 4) * 4 + 1
        depth, pred = depth.astype(np.float32), pred.astype(np.float32)
    depth[rs.uniform(size=depth.shape) > 0.3] = 0.0
    depth[3] = 0.0                                                  # a sample without measurements: ratio 1
    depth[2, 0, 50, 100] = depth[2, 0, 50, 100] if (np.logical_and(depth[2, 0] > 1e-3, depth[2, 0] < 80).sum() % 2) else 0.0
    got = ops.median_ratio(torch.from_numpy(pred).cuda(), torch.from_numpy(depth).cuda(), 4, 1e-3, 80.0, crop).cpu().numpy()
    for i in range(B):
        if i >= 4:
            assert got[i] == 1.0
            continue
        valid = np.logical_and(depth[i, 0] > 1e-3, depth[i, 0] < 80.0)
        em = np.zeros(valid.shape)
        if crop == "garg":
            em[int(0.40810811 * H):int(0.99189189 * H), int(0.03594771 * W):int(0.96405229 * W)] = 1
        elif crop == "eigen":
            em[int(0.3324324 * H):int(0.91351351 * H), int(0.0359477 * W):int(0.96405229 * W)] = 1
        else:
            em[:] = 1
        valid = np.logical_and(valid, em)
        want = np.float32(1.0) if valid.sum() == 0 else np.median(depth[i, 0][valid]) / np.median(pred[i, 0][valid])
        assert got[i] == np.float32(want), (i, got[i], want)


def test_adamw_with_clipping_matches_torch():
    from sqd.optim import FusedAdamW
    torch.manual_seed(2)
    shapes = [(64, 32, 3, 3), (128,), (17, 5), (1000, 33), (3,)]
    ps = [torch.randn(s) for s in shapes]
    ref = [p.clone().double().requires_grad_(True) for p in ps]
    mine = [p.clone().cuda().requires_grad_(True) for p in ps]
    o_ref = torch.optim.AdamW([{"params": ref[:2], "lr": 1e-4}, {"params": ref[2:], "lr": 1e-3}], lr=1e-3, weight_decay=0.1)
    o_mine = FusedAdamW([{"params": mine[:2], "lr": 1e-4}, {"params": mine[2:], "lr": 1e-3}], lr=1e-3, weight_decay=0.1, max_grad_norm=0.1)
    for step in range(4):
        grads = [torch.randn(s) * (0.01 if step == 2 else 1.0) for s in shapes]     # step 2: norm below the threshold -> coefficient 1
        for p, g in zip(ref, grads):
            p.grad = g.double()
        for p, g in zip(mine, grads):
            p.grad = g.cuda()
        tn = nn.utils.clip_grad_norm_(ref, 0.1)
        o_ref.step()
        o_mine.step()
        coef, norm = o_mine.clip_info.cpu().tolist()
        assert abs(norm - float(tn)) <= 1e-5 * float(tn)
        assert abs(coef - min(1.0, 0.1 / (float(tn) + 1e-6))) <= 1e-5
        for a, b in zip(mine, ref):
            assert float((a.detach().cpu().double() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max()) + 1e-8, step


def test_adamw_weight_decay_lives_in_the_param_groups():
    """the decay must be what state_dict() reports and what a loaded (reference AdamW) state dict sets — per group"""
    from sqd.optim import FusedAdamW
    torch.manual_seed(3)
    ps = [torch.randn(40, 8), torch.randn(16)]
    ref = [p.clone().double().requires_grad_(True) for p in ps]
    mine = [p.clone().cuda().requires_grad_(True) for p in ps]
    o_ref = torch.optim.AdamW([{"params": ref[:1], "weight_decay": 0.3}, {"params": ref[1:]}], lr=1e-2, weight_decay=0.05)
    o_mine = FusedAdamW([{"params": mine[:1]}, {"params": mine[1:]}], lr=1e-2, weight_decay=0.0)
    assert [g["weight_decay"] for g in o_mine.state_dict()["param_groups"]] == [0.0, 0.0]
    sd = o_ref.state_dict()                                   # no steps taken yet: hyper-parameters only
    o_mine.load_state_dict(sd)
    assert [g["weight_decay"] for g in o_mine.param_groups] == [0.3, 0.05]
    for step in range(3):
        grads = [torch.randn_like(p) for p in ps]
        for p, g in zip(ref, grads):
            p.grad = g.double()
        for p, g in zip(mine, grads):
            p.grad = g.cuda()
        o_ref.step()
        o_mine.step()
    for a, b in zip(mine, ref):
        assert float((a.detach().cpu().double() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max()) + 1e-8


def test_finetune_step_matches_oracle():
    """two steps of FinetuneTrainer (narrow ConvNeXt U-Net + Self-Query head, median rescale, SILog, clipping, AdamW, OneCycle) against
    the oracle's restatement of train_ft_SQLdepth.py:222-285 with torch.optim.AdamW on the host"""
    from finetune.train_ft_SQLdepth import FinetuneArgs, FinetuneTrainer, synthetic_batch
    from options import MonodepthOptions
    from oracle import finetune_ref as FR
    from oracle import torch_ref as O
    torch.manual_seed(0)
    opt = MonodepthOptions().parse(["--backbone", "convnext_large", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
                                    "--dec_channels", "64", "32", "16", "8", "--height", "64", "--width", "96", "--max_depth", "80.0",
                                    "--sqd_no_conv_tune", "--sqd_synthetic"])
    opt.sqd_convnext_depths, opt.sqd_convnext_dims = (1, 1, 2, 1), (16, 32, 64, 128)
    args = FinetuneArgs(bs=4, lr=3e-4, epochs=2)
    tr = FinetuneTrainer(opt, args, steps_per_epoch=4)
    for m in tr.model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0
    for n, p in tr.model.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.4)

    class RefModel(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = O.Unet(3, 16, (64, 32, 16, 8), depths=(1, 1, 2, 1), dims=(16, 32, 64, 128))
            self.depth_decoder = O.QueryTrDecoder(16, 16, 8, 4, 12, 24, min_val=0.001, max_val=80.0, dim_feedforward=1024, dropout=0.0)

        def forward(self, x):
            return self.depth_decoder(self.encoder(x))[("disp", 0)]
    ref = RefModel()
    ref.encoder.load_state_dict({k: v.detach().cpu() for k, v in tr.model.encoder.state_dict().items()})
    ref.depth_decoder.load_state_dict({k: v.detach().cpu() for k, v in tr.model.depth_decoder.state_dict().items()})
    ref.train()
    o_ref = torch.optim.AdamW([{"params": list(ref.encoder.parameters()), "lr": args.lr / 10},
                               {"params": list(ref.depth_decoder.parameters()), "lr": args.lr}], weight_decay=args.wd, lr=args.lr)
    s_ref = torch.optim.lr_scheduler.OneCycleLR(o_ref, args.lr, epochs=args.epochs, steps_per_epoch=4, cycle_momentum=True, base_momentum=0.85,
                                                max_momentum=0.95, div_factor=args.div_factor, final_div_factor=args.final_div_factor)
    for step in range(2):
        batch = synthetic_batch(4, 64, 96, 88, 120, seed=step)
        want, ratios = FR.finetune_step(ref, o_ref, s_ref, {k: v.clone() for k, v in batch.items()}, args)
        got, ratio = tr.train_step(batch)
        assert abs(float(got) - float(want)) <= 3e-4 * abs(float(want)), (step, float(got), float(want))
        r = ratio.cpu().numpy()
        assert np.allclose(r[:2], np.array(ratios, dtype=np.float32), rtol=2e-4) and (r[2:] == 1).all()
        for g_m, g_r in zip(tr.optimizer.param_groups, o_ref.param_groups):
            assert abs(g_m["lr"] - g_r["lr"]) <= 1e-12 and abs(g_m["betas"][0] - g_r["betas"][0]) <= 1e-12
    # weights after two clipped AdamW steps
    mine = tr.model.state_dict()
    for k, v in list(ref.encoder.state_dict().items())[:40] + list(ref.depth_decoder.state_dict().items())[:20]:
        kk = ("encoder." if k in ref.encoder.state_dict() and ("encoder." + k) in mine else "depth_decoder.") + k
        if v.dtype != torch.float32 or kk not in mine:
            continue
        d = (mine[kk].detach().cpu() - v).abs()
        assert float((d > 3e-5 + 1e-3 * v.abs()).float().mean()) < 2e-2, (kk, float(d.max()))


@pytest.mark.parametrize("crop", ["garg", "eigen", "eigen_nyu", None])
def test_metric_depth_eval(crop):
    """sqd_metric_depth_eval against the oracle's restatement of validate()'s per-image body (numpy float32, as the reference; its
    compute_errors is pinned to the reference's own by G22): the nine metrics within 2e-5 relative (numpy sums float32 pairwise, the
    kernel in float64), ratio and valid count exact."""
    from oracle import finetune_ref as FR
    from sqd import ops
    rs = np.random.RandomState(9)
    B, H, W = 4, 480, 640
    depth = rs.uniform(0.5, 90.0, (B, H, W)).astype(np.float32)
    depth[rs.uniform(size=depth.shape) > 0.25] = 0.0
    depth[2] = 0.0                                                   # an image without ground truth: NaN metrics, 0 valid pixels
    pred = (np.abs(depth + rs.normal(0, 3.0, depth.shape)) * 1.7 + 0.05 + (depth == 0) * rs.uniform(1, 50, depth.shape)).astype(np.float32)
    got = ops.metric_depth_eval(torch.from_numpy(pred).cuda(), torch.from_numpy(depth).cuda(), 1e-3, 80.0, crop).cpu().numpy()
    for i in range(B):
        want = FR.validate_image(pred[i], depth[i], 1e-3, 80.0, garg_crop=crop == "garg", eigen_crop=crop in ("eigen", "eigen_nyu"),
                                 dataset="nyu" if crop == "eigen_nyu" else "kitti")
        if want is None:
            assert got[i, 10] == 0 and np.isnan(got[i, :10]).all()
            continue
        e, ratio, n = want
        assert got[i, 10] == n and np.float32(got[i, 9]) == np.float32(ratio)
        for j, k in enumerate(ops.METRIC_DEPTH_NAMES):
            assert abs(got[i, j] - float(e[k])) <= 2e-5 * abs(float(e[k])) + 1e-7, (i, k, got[i, j], float(e[k]))
    if crop is None:
        # a prediction that is the ground truth at half scale: the median ratio undoes it — a1..a3 = 1, errors 0
        gt = rs.uniform(1.0, 70.0, (1, 50, 100)).astype(np.float32)
        t = ops.metric_depth_eval(torch.from_numpy(gt * np.float32(0.5)).cuda(), torch.from_numpy(gt).cuda(), 1e-3, 80.0, None).cpu().numpy()[0]
        assert t[0] == 1 and t[1] == 1 and t[2] == 1 and abs(t[3]) < 1e-6 and abs(t[9] - 2.0) < 1e-6 and t[10] == 5000


def test_finetune_validate_matches_oracle():
    """validate() of the finetune loop on a narrow ConvNeXt U-Net: mean metrics and mean SILog against the oracle's image-by-image numpy loop"""
    from finetune.train_ft_SQLdepth import FinetuneArgs, FinetuneTrainer, synthetic_batch, validate
    from finetune.loss import SILogLoss
    from options import MonodepthOptions
    from oracle import finetune_ref as FR
    from sqd import ops
    torch.manual_seed(1)
    opt = MonodepthOptions().parse(["--backbone", "convnext_large", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
                                    "--dec_channels", "64", "32", "16", "8", "--height", "64", "--width", "96", "--max_depth", "80.0",
                                    "--sqd_no_conv_tune", "--sqd_synthetic"])
    opt.sqd_convnext_depths, opt.sqd_convnext_dims = (1, 1, 2, 1), (16, 32, 64, 128)
    args = FinetuneArgs(bs=2, lr=1e-4, epochs=1)
    tr = FinetuneTrainer(opt, args, steps_per_epoch=2)
    batches = [synthetic_batch(2, 64, 96, 88, 120, seed=s) for s in (5, 6)]
    means, si = validate(tr, batches)
    tr.model.eval()
    want, losses = [], []
    with torch.no_grad():
        for b in batches:
            pred = ops.ResizeAlignCorners.apply(tr.model(b["image"].cuda()).contiguous(), 88, 120).cpu()
            for i in range(2):
                r = FR.validate_image(pred[i, 0].numpy(), b["depth"][i, 0].numpy(), args.min_depth_eval, args.max_depth_eval, garg_crop=True)
                want.append(r[0])
                losses.append(float(FR.SILogLoss()(pred[i:i + 1], b["depth"][i:i + 1], mask=b["depth"][i:i + 1] > args.min_depth, interpolate=False)))
    for k in ops.METRIC_DEPTH_NAMES:
        w = float(np.mean([float(e[k]) for e in want]))
        assert abs(means[k] - w) <= 5e-5 * abs(w) + 1e-6, (k, means[k], w)
    assert abs(si - float(np.mean(losses))) <= 1e-4 * abs(float(np.mean(losses)))


def test_metric_depth_eval_without_median_scaling():
    """evaluate_metric_depth.py's eval() body: unscaled, unclamped prediction (inf / nan replaced) against the oracle"""
    from oracle import finetune_ref as FR
    from sqd import ops
    rs = np.random.RandomState(11)
    B, H, W = 3, 352, 1216
    depth = rs.uniform(0.5, 90.0, (B, H, W)).astype(np.float32)
    depth[rs.uniform(size=depth.shape) > 0.2] = 0.0
    pred = (np.abs(depth + rs.normal(0, 2.0, depth.shape)) + 0.05 + (depth == 0) * rs.uniform(1, 50, depth.shape)).astype(np.float32)
    pred[0, 200, 300], pred[1, 210, 400] = np.inf, np.nan
    depth[0, 200, 300], depth[1, 210, 400] = 30.0, 20.0
    got = ops.metric_depth_eval(torch.from_numpy(pred).cuda(), torch.from_numpy(depth).cuda(), 1e-3, 80.0, "garg", median_scaling=False).cpu().numpy()
    for i in range(B):
        e = FR.eval_image(pred[i], depth[i], 1e-3, 80.0, garg_crop=True)
        assert got[i, 9] == 1.0
        for j, k in enumerate(ops.METRIC_DEPTH_NAMES):
            assert abs(got[i, j] - float(e[k])) <= 2e-5 * abs(float(e[k])) + 1e-7, (i, k, got[i, j], float(e[k]))


def test_predict_tta_and_evaluate():
    """flip test-time augmentation + evaluate() on a narrow model against the same composition on the host"""
    from finetune.evaluate_metric_depth import evaluate, predict_tta
    from finetune.train_ft_SQLdepth import FinetuneArgs, FinetuneTrainer, synthetic_batch
    from options import MonodepthOptions
    from oracle import finetune_ref as FR
    torch.manual_seed(2)
    opt = MonodepthOptions().parse(["--backbone", "convnext_large", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
                                    "--dec_channels", "64", "32", "16", "8", "--height", "64", "--width", "96", "--max_depth", "80.0",
                                    "--sqd_no_conv_tune", "--sqd_synthetic"])
    opt.sqd_convnext_depths, opt.sqd_convnext_dims = (1, 1, 2, 1), (16, 32, 64, 128)
    args = FinetuneArgs(bs=2)
    tr = FinetuneTrainer(opt, args, steps_per_epoch=2)
    tr.model.eval()
    b = synthetic_batch(2, 64, 96, 64, 96, seed=3)
    img = b["image"].cuda()
    with torch.no_grad():
        p0 = tr.model(img.contiguous(memory_format=torch.channels_last))
        p1 = torch.flip(tr.model(torch.flip(img, [3]).contiguous(memory_format=torch.channels_last)), [3])
        want = F.interpolate(0.5 * (p0 + p1), (64, 96), mode="bilinear", align_corners=True)
    got = predict_tta(tr.model, img)
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    metrics, invalid = evaluate(tr.model, [b], args)
    e = [FR.eval_image(want[i, 0].cpu().numpy(), b["depth"][i, 0].numpy(), args.min_depth, args.max_depth, garg_crop=True) for i in range(2)]
    assert invalid == 0
    for k, v in metrics.items():
        assert abs(v - round(float(np.mean([float(x[k]) for x in e])), 3)) <= 1.1e-3, k
