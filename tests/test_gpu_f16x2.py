"""Two-term fp16 convolution operands (csrc/conv.hip "f16x2", include/sqd.h section 10b) on the GPU: every plan of the arithmetic
against float64 next to the fp32 MFMA plan of the same geometry, magnitudes far from 1, and the max |.| bookkeeping the operand
scales come from (amax.hip, the producers' `amax` outputs) against torch's own maximum, bit for bit.
reference: the convolutions of networks/resnet_encoder.py:89-147, pose_cnn.py:14-29 multiply in fp32 on cuDNN; this arithmetic has to
stay as close to float64 as the fp32 MFMA chain does."""
import ctypes

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GEOMS = [  # N, C, H, W, K, R, stride, pad
    (2, 64, 24, 40, 128, 3, 1, 1),       # input-patch and implicit-GEMM plans
    (2, 128, 12, 20, 64, 1, 1, 0),       # 1x1
    (2, 64, 24, 40, 64, 3, 2, 1),        # stride 2: stride classes in the data gradient
    (1, 256, 10, 12, 32, 3, 1, 1),       # 32 filters, ragged tiles
    (2, 32, 33, 48, 32, 5, 2, 2),        # 5x5 stride 2, odd height
    (1, 512, 6, 20, 512, 3, 1, 1),       # few pixels, long reduction: split-K
]


def _ref64(x, w, dy, stride, pad):
    xr, wr = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    gx, gw = torch.autograd.grad(yr, (xr, wr), dy.double().cpu())
    return yr.detach(), gx, gw


def _inputs(N, C, H, W, K, R, stride, pad, dist, seed=0):
    torch.manual_seed(seed)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = torch.randn(N, C, H, W, device="cuda")
    dy = torch.randn(N, K, Ho, Wo, device="cuda")
    w = torch.randn(K, C, R, R, device="cuda") * (2.0 / (C * R * R)) ** 0.5
    if dist == "heavy":                   # per-pixel magnitudes over ~8 decades, gradients around 1e-7
        x = x * torch.exp(3 * torch.randn(N, 1, H, W, device="cuda"))
        dy = dy * torch.exp(3 * torch.randn(N, 1, Ho, Wo, device="cuda")) * 1e-7
    elif dist == "relu":
        x = F.relu(x)
    cl = torch.channels_last
    return x.contiguous(memory_format=cl), w.contiguous(memory_format=cl), dy.contiguous(memory_format=cl), (N, H, W, C, K, R, R, stride, pad, Ho, Wo)


def _run(x, w, dy, stride, pad):
    from sqd import nnkernels
    conv = nn.Conv2d(w.shape[1], w.shape[0], w.shape[2], stride, pad, bias=False).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        conv.weight.copy_(w)
    xg = x.clone().requires_grad_(True)
    nnkernels.begin_step()
    y = nnkernels.conv2d_native(xg, conv, None)
    y.backward(dy)
    return y.detach(), xg.grad, conv.weight.grad


def _err(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("dist", ["normal", "relu", "heavy"])
@pytest.mark.parametrize("geom", GEOMS)
def test_f16x2_plans_against_float64(geom, dist):
    """every two-term plan the library accepts for the geometry: forward, data gradient and weight gradient (through the autograd node,
    i.e. with the operand scales taken from the tensors' tags or a standalone pass) no further from float64 than 1.25x the three-term
    bf16 plan of the same tile / split — or than the cost-model fp32 MFMA plan, where that one is further off (the heavy-tailed
    weight gradients: a tensor-wide scale gives the small pixels fewer bits than a three-term split does, 2.0e-7 against 1.0e-7
    with the fp32 chain at 3.2e-7) —, and within 4x the fp32 plan unless the three-term plan of the same tile is further off still (long
    reductions of non-negative inputs: the summation order of the tile, not the operand arithmetic, sets the distance)"""
    from sqd import lib, nnkernels
    L = lib.lib()
    N, C, H, W, K, R, stride, pad = geom
    x, w, dy, g = _inputs(*geom, dist)
    yr, gxr, gwr = _ref64(x, w, dy, stride, pad)
    nnkernels.reset_plans()
    nnkernels.amax_enable(True)
    try:
        y0, gx0, gw0 = _run(x, w, dy, stride, pad)                    # cost-model plans: fp32 MFMA
        e32 = (_err(y0, yr), _err(gx0, gxr), _err(gw0, gwr))
        tried = 0
        for flags in (32 + 1024 + 4096, 32 + 1024 + 4096 + 256, 32 + 1024 + 4096 + 2048, 32 + 1024 + 4096 + 2048 + 256):
            for bm, bn in ((128, 128), (128, 64), (64, 128), (64, 64), (128, 32), (64, 32)):
                for z in (1, 2, 4):
                    # (a tile may suit one pass only: 32 filters take the 32-column tiles forward, the 256 input channels the wide ones backward)
                    ok = [L.sqd_conv_set_plan(mode, *g, bm, bn, z, flags) == 0 for mode in (0, 1)]
                    for mode in (0, 1):
                        L.sqd_conv_set_plan(mode, *g, 0, 0, 0, 16)
                    if not any(ok):
                        continue
                    for mode in (0, 1):
                        if ok[mode]:
                            nnkernels._register_conv_plan(mode, g, (bm, bn, z, flags))
                    y, gx, _ = _run(x, w, dy, stride, pad)
                    # the yardstick: the three-term bf16 plan of the SAME tile and split (the summation order — tile, slice width, split-K —
                    # moves an fp32 result by more than the arithmetic does: tools/diag_f16x2_err.py), and the cost-model fp32 plan
                    for mode in (0, 1):
                        if ok[mode]:
                            nnkernels._register_conv_plan(mode, g, (bm, bn, z, flags - 4096))
                    y3, gx3, _ = _run(x, w, dy, stride, pad)
                    for mode in (0, 1):
                        nnkernels._register_conv_plan(mode, g, (0, 0, 0, 16))
                    tried += 1
                    ey, ex = _err(y, yr), _err(gx, gxr)
                    ey3, ex3 = _err(y3, yr), _err(gx3, gxr)
                    if ok[0]:
                        assert ey <= max(1.25 * ey3, e32[0]) + 5e-8 and (ey <= 4.0 * e32[0] + 2e-7 or ey <= ey3), ("fwd", bm, bn, z, flags, ey, ey3, e32[0])
                    if ok[1]:
                        assert ex <= max(1.25 * ex3, e32[1]) + 5e-8 and (ex <= 4.0 * e32[1] + 2e-7 or ex <= ex3), ("dgrad", bm, bn, z, flags, ex, ex3, e32[1])
        assert tried >= 2
        nnkernels.reset_plans()
        # weight gradient: impl 7, every register tile that divides
        Ho, Wo = g[9], g[10]
        tried_w = 0
        for v in range(8):
            for sp in (1, 3):
                if L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, 7 | (v << 4), sp) != 0:
                    continue
                nnkernels._register_wgrad_plan((N, Ho, Wo, C, K, R, R), (7 | (v << 4), sp))
                _, _, gw = _run(x, w, dy, stride, pad)
                nnkernels._register_wgrad_plan((N, Ho, Wo, C, K, R, R), (6 | (v << 4), sp))
                _, _, gw3 = _run(x, w, dy, stride, pad)
                tried_w += 1
                ew, ew3 = _err(gw, gwr), _err(gw3, gwr)
                assert ew <= max(1.25 * ew3, e32[2]) + 5e-8 and (ew <= 4.0 * e32[2] + 2e-7 or ew <= ew3), ("wgrad", v, sp, ew, ew3, e32[2])
        if C % 64 == 0 and K % 64 == 0 and (Wo % 2 == 0 or R == 1):
            assert tried_w >= 2
    finally:
        nnkernels.reset_plans()


@pytest.mark.parametrize("sx,sdy,sw", [(1e-30, 1e-12, 1e-3), (1e25, 1e-30, 1.0), (3e-41, 1.0, 1.0), (0.0, 1.0, 1.0)])
def test_f16x2_magnitudes_far_from_one(sx, sdy, sw):
    """operands of 1e-30 or 1e25, fp32 subnormals and an all-zero tensor: the power-of-two scale keeps every fp16 term finite, the
    results stay within 1e-6 of float64 (relative to their own largest entry) and nothing turns into inf / NaN"""
    from sqd import nnkernels
    geom = (2, 64, 24, 40, 128, 3, 1, 1)
    x, w, dy, g = _inputs(*geom, "normal", seed=3)
    x, w, dy = x * sx, w * sw, dy * sdy
    yr, gxr, gwr = _ref64(x, w, dy, 1, 1)
    nnkernels.reset_plans()
    nnkernels.amax_enable(True)
    try:
        for mode in (0, 1):
            nnkernels._register_conv_plan(mode, g, (128, 64, 1, 32 + 1024 + 4096))
        nnkernels._register_wgrad_plan((2, 24, 40, 64, 128, 3, 3), (7, 2))
        y, gx, gw = _run(x, w, dy, 1, 1)
    finally:
        nnkernels.reset_plans()
    for name, a, ref in (("y", y, yr), ("dx", gx, gxr), ("dw", gw, gwr)):
        assert bool(torch.isfinite(a).all()), name
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(a.abs().max()) == 0.0, name
        elif scale > 1e-37:               # (below that the float64 reference itself is an fp32 subnormal: only finiteness is asked)
            assert float((a.double().cpu() - ref).abs().max()) <= 1e-6 * scale, (name, scale)


def test_amax_kernels_match_torch_bit_for_bit():
    from sqd import lib, nnkernels
    from sqd.ops import _ptr, _stream
    L = lib.lib()
    torch.manual_seed(0)
    for n in (4, 1000, 12 * 48 * 160 * 64 + 3):
        t = torch.randn(n + 4, device="cuda")[:n] * 37.0 if n % 4 else torch.randn(n, device="cuda") * 1e-9
        a = torch.full((nnkernels.AMAX_REC,), 7.0, device="cuda")
        lib.check(L.sqd_amax(_ptr(t), n, _ptr(a), _stream()), "amax")
        assert nnkernels.amax_value(a) == float(t.abs().max()), n
    # the optimiser-table form: a few tensors of different sizes
    ws = [torch.randn(s, device="cuda") * (i + 1) for i, s in enumerate((5, 4096, 4097, 70000))]
    chunk = L.sqd_adam_chunk_elems()
    recs = torch.tensor([[w_.data_ptr(), 0, 0, w_.numel()] for w_ in ws], dtype=torch.int64, device="cuda")
    chunks = torch.tensor([[i, c] for i, w_ in enumerate(ws) for c in range((w_.numel() + chunk - 1) // chunk)], dtype=torch.int32, device="cuda")
    out = torch.full((len(ws) * nnkernels.AMAX_REC,), 3.0, device="cuda")
    lib.check(L.sqd_amax_multi(_ptr(recs), _ptr(chunks), chunks.shape[0], len(ws), _ptr(out), _stream()), "amax_multi")
    assert [nnkernels.amax_value(r) for r in out.view(len(ws), -1)] == [float(w_.abs().max()) for w_ in ws]


def test_producers_record_the_maximum_of_what_they_write():
    """BatchNorm forward / backward, convolution epilogues (plain, split-K, input-patch), activation gradient, up-sampling + concat and the
    frame staging tag their outputs with max |output| — equal to torch's maximum of the tensor, bit for bit"""
    from sqd import nnkernels, nnops
    nnkernels.reset_plans()
    nnkernels.amax_enable(True)
    torch.manual_seed(1)
    cl = torch.channels_last

    def tag_value(t):
        a = nnkernels._amax_get(t)
        assert a is not None, "no tag"
        return nnkernels.amax_value(a)
    nnkernels.begin_step()
    conv = nn.Conv2d(64, 128, 3, 1, 1, bias=True).cuda().to(memory_format=cl)
    bn = nn.BatchNorm2d(128).cuda().train()
    x = (torch.randn(2, 64, 24, 40, device="cuda") * 3).contiguous(memory_format=cl).requires_grad_(True)
    y = nnkernels.conv2d_native(x, conv, "relu")
    assert tag_value(y) == float(y.abs().max())
    z = nnkernels.batch_norm_act(y, bn, "relu")
    assert tag_value(z) == float(z.abs().max())
    seen = {}

    def grab(name):
        def hook(g):
            seen[name] = (None if nnkernels._amax_get(g) is None else nnkernels.amax_value(nnkernels._amax_get(g)), float(g.abs().max()))
        return hook
    y.register_hook(grab("bn dx"))
    x.register_hook(grab("conv dx"))
    z.backward(torch.randn_like(z) * 1e-6)
    assert seen["bn dx"][0] == seen["bn dx"][1] and seen["conv dx"][0] == seen["conv dx"][1], seen
    # split-K and input-patch plans record it in the sum over the splits / in the patch epilogue
    g = (2, 24, 40, 64, 128, 3, 3, 1, 1, 24, 40)
    try:
        for plan in ((64, 64, 4, 32 + 1024), (128, 64, 1, 32 + 1024 + 2048), (64, 64, 2, 32 + 1024 + 4096)):
            nnkernels._register_conv_plan(0, g, plan)
            nnkernels.begin_step()
            y2 = nnkernels.conv2d_native(x.detach(), conv, None)
            assert tag_value(y2) == float(y2.abs().max()), plan
    finally:
        nnkernels.reset_plans()
    nnkernels.begin_step()
    lo = torch.randn(2, 32, 6, 10, device="cuda").contiguous(memory_format=cl)
    sk = (torch.randn(2, 16, 12, 20, device="cuda") * 5).contiguous(memory_format=cl)
    u = nnops.upsample_concat(lo, sk)
    assert tag_value(u) == float(u.abs().max())
    stem = nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    frames = torch.rand(2, 3, 32, 48, device="cuda")
    ys = nnkernels.conv2d_stem_s2d_planar([(frames, None)], stem, None, None, (0.45, 0.225))
    assert tag_value(ys) == float(ys.abs().max())


def test_first_step_plan_timing_offers_the_two_term_plans():
    """with plan timing on, a config-B layer ends up on a two-term plan in at least one pass and the autograd node feeds it the scales
    (the tags of its producers: no standalone pass for a BatchNorm -> convolution -> BatchNorm chain after the first use)"""
    from sqd import nnkernels
    nnkernels.reset_plans()
    nnkernels.amax_enable(True)
    torch.manual_seed(2)
    cl = torch.channels_last
    conv1 = nn.Conv2d(64, 64, 3, 1, 1, bias=False).cuda().to(memory_format=cl)
    conv2 = nn.Conv2d(64, 256, 1, 1, 0, bias=False).cuda().to(memory_format=cl)
    bn0, bn1, bn2 = nn.BatchNorm2d(64).cuda().train(), nn.BatchNorm2d(64).cuda().train(), nn.BatchNorm2d(256).cuda().train()
    x = torch.randn(12, 64, 48, 160, device="cuda").contiguous(memory_format=cl)
    old_tune = nnkernels.TUNE_CONV
    nnkernels.TUNE_CONV = True
    try:
        for step in range(2):
            nnkernels.begin_step()
            before = dict(nnkernels.AMAX_STATS["sites"])
            a = nnkernels.batch_norm_act(x.clone().requires_grad_(True), bn0, "relu")
            b = nnkernels.batch_norm_act(nnkernels.conv2d_native(a, conv1, None), bn1, "relu")
            c = nnkernels.batch_norm_act(nnkernels.conv2d_native(b, conv2, None), bn2, None)
            c.backward(torch.randn_like(c) * 1e-6)
            new = {k: v - before.get(k, 0) for k, v in nnkernels.AMAX_STATS["sites"].items() if v != before.get(k, 0)}
        mix = nnkernels.plan_mix()
        assert any("f16x2" in name for p in mix.values() for name in p), mix
        # second step: the only standalone pass left is the incoming gradient of the last BatchNorm's consumer-less output (none here)
        assert set(new) <= {"conv backward: output gradient"} and sum(new.values()) <= 1, new
    finally:
        nnkernels.TUNE_CONV = old_tune
        nnkernels.reset_plans()


def test_tagged_gradient_summed_in_place_is_not_trusted():
    """a Conv2d output with TWO consumers, one of which returns a gradient tagged with its max |.| (the data gradient of the
    next convolution).  Autograd's input buffer adds the second consumer's gradient IN PLACE into that tensor: the Python tag survives, the
    maximum it names does not — a 64x louder second gradient would overflow the two-term fp16 high part to inf if the stale tag were used.
    The tag carries the tensor's version counter; the summed gradient gets a stand-alone pass.  Gradients against float64."""
    from sqd import nnkernels
    torch.manual_seed(0)
    N, C, H, W = 2, 64, 24, 40
    cl = torch.channels_last
    c1 = nn.Conv2d(C, C, 3, 1, 1, bias=False).cuda().to(memory_format=cl)
    c2 = nn.Conv2d(C, C, 3, 1, 1, bias=False).cuda().to(memory_format=cl)
    x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=cl).requires_grad_(True)
    dy2 = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=cl)
    w_side = 64.0 * torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=cl)          # the second consumer: y * w_side
    nnkernels.begin_step()
    y = nnkernels.conv2d_native(x, c1, None)
    z = nnkernels.conv2d_native(y, c2, None)                      # consumer 1: its backward hands back a TAGGED d y
    loss = (z * dy2).sum() + (y * w_side).sum()                   # consumer 2: an untagged, much louder gradient w.r.t. y
    loss.backward()
    xr = x.detach().double().cpu().requires_grad_(True)
    w1, w2 = c1.weight.detach().double().cpu().requires_grad_(True), c2.weight.detach().double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, w1, None, 1, 1)
    lr = (F.conv2d(yr, w2, None, 1, 1) * dy2.double().cpu()).sum() + (yr * w_side.double().cpu()).sum()
    lr.backward()
    for name, got, want in (("dx", x.grad, xr.grad), ("dw1", c1.weight.grad, w1.grad), ("dw2", c2.weight.grad, w2.grad)):
        assert torch.isfinite(got).all(), name
        e = _err(got, want)
        print("two consumers, %s: error against float64 %.2e of max" % (name, e))
        assert e <= 5e-6, (name, e)
