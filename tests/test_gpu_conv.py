"""GPU parity of the native implicit-GEMM convolution (csrc/conv.hip through the C ABI: forward, input
gradient, weight + bias gradient) against torch's fp32 convolution on the same inputs."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # N, C, H, W, K, R, stride, pad, bias, act
    (2, 64, 12, 20, 256, 1, 1, 0, False, None),        # bottleneck 1x1
    (2, 64, 24, 40, 64, 3, 1, 1, False, None),         # 3x3 stride 1
    (2, 128, 24, 40, 128, 3, 2, 1, False, None),       # 3x3 stride 2 (ResNet v1.5)
    (2, 256, 12, 20, 512, 1, 2, 0, False, None),       # downsample 1x1 stride 2
    (2, 2048, 3, 5, 256, 1, 1, 1, True, None),         # DecoderBN.conv2: 1x1 with padding 1
    (2, 32, 32, 48, 32, 16, 16, 0, True, None),        # patch embedding 16x16 / 16
    (2, 16, 33, 47, 32, 5, 2, 2, True, "relu"),        # PoseCNN 5x5 stride 2 + ReLU, odd sizes
    (1, 1280, 6, 10, 128, 3, 1, 1, True, None),        # UpSampleBN first conv (concat input)
    (2, 32, 20, 28, 16, 3, 1, 1, True, None),          # K = 16
    (3, 48, 9, 7, 80, 3, 1, 1, True, None),            # channel counts that are not powers of two
    (12, 64, 48, 160, 64, 3, 1, 1, False, None),       # config-B layer1 shape
    (2, 32, 13, 21, 64, 3, 2, 1, True, None),          # stride 2 on odd sizes (ragged stride classes in dgrad)
    (1, 512, 6, 20, 512, 3, 1, 1, False, None),        # few pixels, long reduction: split-K path
    (4, 32, 24, 40, 64, 3, 1, 1, True, None),          # direct-operand wgrad with the fused bias gradient
    (2, 16, 64, 96, 32, 5, 2, 2, True, "relu"),        # same, 5x5 stride 2 + ReLU, K = 32
    # EfficientNet-b5 widths: 24 / 40 channels are multiples of 4, not of 16 — the last reduction slice is partial
    (2, 24, 20, 28, 144, 1, 1, 0, False, None),        # MBConv expansion 24 -> 144 (forward reduction over 24)
    (2, 144, 20, 28, 24, 1, 1, 0, False, None),        # projection 144 -> 24 (data gradient reduces over 24)
    (2, 40, 10, 14, 240, 1, 1, 0, False, None),
    (2, 240, 10, 14, 40, 1, 1, 0, True, None),
    (1, 88, 12, 20, 40, 3, 1, 1, True, None),          # decoder: concat widths such as 64 + 24 = 88
]


@pytest.mark.parametrize("N,C,H,W,K,R,stride,pad,bias,act", CASES)
def test_conv_fwd_bwd(N, C, H, W, K, R, stride, pad, bias, act):
    from sqd import nnkernels
    torch.manual_seed(C + K + R)
    conv = nn.Conv2d(C, K, R, stride, pad, bias=bias)
    x = torch.randn(N, C, H, W)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    if act == "relu":
        yr = F.relu(yr)
    wgt = torch.randn_like(yr)
    (yr * wgt).sum().backward()
    conv_g = nn.Conv2d(C, K, R, stride, pad, bias=bias).cuda()
    conv_g.load_state_dict(conv.state_dict())
    conv_g = conv_g.to(memory_format=torch.channels_last)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = nnkernels.conv2d_native(xg, conv_g, act)
    (y * wgt.cuda()).sum().backward()

    def close(a, b, name, rtol=2e-4):
        a, b = a.detach().cpu().float(), b.detach().float()
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        assert err <= rtol * scale + 1e-6, (name, err, scale)
    close(y, yr, "y")
    close(xg.grad, xr.grad, "dx")
    close(conv_g.weight.grad, conv.weight.grad, "dw", 5e-4)
    if bias:
        close(conv_g.bias.grad, conv.bias.grad, "db", 5e-4)


@pytest.mark.parametrize("N,C,H,W,K,R,stride,pad", [(2, 64, 24, 40, 128, 3, 2, 1), (2, 128, 12, 20, 64, 1, 1, 0), (1, 256, 9, 11, 32, 3, 1, 1),
                                                     (2, 64, 12, 20, 128, 1, 2, 0)])
def test_every_registered_plan_is_exact(N, C, H, W, K, R, stride, pad):
    """Every tile / split-K / slice-width plan sqd_conv_set_plan accepts must give the same convolution (the first-step
    tuner picks among them by time alone).  The node runs with its pass-through output (skip=True), so the data gradient
    also has to add the second gradient in its epilogue — including the stride classes that are a pure fill."""
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(7)
    conv = nn.Conv2d(C, K, R, stride, pad, bias=True).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    geom = (N, H, W, C, K, R, R, stride, pad, Ho, Wo)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    gy = torch.randn_like(yr)
    gxr, = torch.autograd.grad(yr, xr, gy)
    gskip = torch.randn_like(x)
    gxr = gxr + gskip
    tried = 0
    try:
        for bm, bn in nnkernels._TUNE_TILES:
            for bk in (16, 32, 272, 288, 528, 544, 576, 1056, 1312):  # + 256: 8-wave workgroups, + 512: single-buffered LDS (also 64-wide slices), 32 + 1024: three-term bf16 operands
                for z in nnkernels._TUNE_Z:
                    ok = [L.sqd_conv_set_plan(mode, *geom, bm, bn, z, bk) == 0 for mode in (0, 1)]
                    if not any(ok):
                        continue
                    for mode in (0, 1):
                        if not ok[mode]:
                            L.sqd_conv_set_plan(mode, *geom, 0, 0, 0, 16)
                    nnkernels._PLAN_CACHE.clear()
                    xg = x.clone().requires_grad_(True)
                    y, xs = nnkernels.conv2d_native(xg, conv, None, True)
                    assert xs.data_ptr() == xg.data_ptr()
                    gx, = torch.autograd.grad((y, xs), xg, (gy, gskip))
                    tried += 1
                    assert torch.allclose(y, yr, rtol=1e-4, atol=1e-4 * float(yr.abs().max())), (bm, bn, z, bk)
                    assert torch.allclose(gx, gxr, rtol=1e-4, atol=1e-4 * float(gxr.abs().max())), (bm, bn, z, bk)
    finally:
        for mode in (0, 1):
            L.sqd_conv_set_plan(mode, *geom, 0, 0, 0, 16)
        nnkernels._PLAN_CACHE.clear()
    assert tried >= 8


@pytest.mark.parametrize("N,C,H,W,K,bias,act", [(2, 3, 16, 24, 64, False, None), (2, 6, 32, 20, 16, True, "relu"), (12, 3, 192, 640, 64, False, None)])
def test_stem_space_to_depth(N, C, H, W, K, bias, act):
    """7x7/2 stem as a 4x4/1 convolution on the space-to-depth image: output and filter / bias gradients against ATen."""
    from sqd import nnkernels
    torch.manual_seed(C + K)
    conv = nn.Conv2d(C, K, 7, 2, 3, bias=bias)
    x = torch.randn(N, C, H, W)
    yr = conv(x)
    if act == "relu":
        yr = F.relu(yr)
    gy = torch.randn_like(yr)
    yr.backward(gy)
    conv_g = nn.Conv2d(C, K, 7, 2, 3, bias=bias).cuda()
    conv_g.load_state_dict(conv.state_dict())
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    assert nnkernels.stem_s2d_supported(conv_g, xg)
    y = nnkernels.conv2d_stem_s2d(xg, conv_g, act)
    y.backward(gy.cuda())
    assert y.shape == yr.shape
    assert torch.allclose(y.cpu(), yr.detach(), rtol=1e-4, atol=1e-4 * float(yr.abs().max()))
    gw, gwr = conv_g.weight.grad.cpu(), conv.weight.grad
    assert float((gw - gwr).abs().max()) <= 2e-4 * float(gwr.abs().max()), float((gw - gwr).abs().max()) / float(gwr.abs().max())
    if bias:
        assert torch.allclose(conv_g.bias.grad.cpu(), conv.bias.grad, rtol=2e-4, atol=2e-4 * float(conv.bias.grad.abs().max()))


@pytest.mark.parametrize("N,C,H,W,K,R,stride,pad,bias", [(6, 32, 45, 61, 64, 3, 1, 1, True), (2, 64, 24, 40, 256, 1, 1, 0, False),
                                                          (4, 3, 128, 192, 64, 7, 2, 3, False)])
def test_conv_epilogue_feeds_batchnorm_statistics(N, C, H, W, K, R, stride, pad, bias):
    """conv -> BatchNorm(train) -> ReLU through nnops.conv_bn_act with the statistics partials taken from the convolution's
    epilogue: output, running statistics and all gradients against the torch composite."""
    from sqd import nnkernels, nnops
    torch.manual_seed(K + R)
    conv, bn = nn.Conv2d(C, K, R, stride, pad, bias=bias), nn.BatchNorm2d(K)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(N, C, H, W)
    conv_g, bn_g = nn.Conv2d(C, K, R, stride, pad, bias=bias).cuda(), nn.BatchNorm2d(K).cuda()
    conv_g.load_state_dict(conv.state_dict())
    bn_g.load_state_dict(bn.state_dict())
    conv_g = conv_g.to(memory_format=torch.channels_last)
    conv, bn = conv.double(), bn.double()                     # fp64 reference: the comparison then measures OUR rounding only
    xr = x.double().requires_grad_(C > 3)
    yr = F.relu(bn(conv(xr)))
    gy = torch.randn_like(yr)
    yr.backward(gy)
    gy = gy.float()
    try:
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(C > 3)
        y = nnops.conv_bn_act(xg, conv_g, bn_g, "relu")
        geom = nnkernels.conv_out_geom(xg, conv_g, s2d=(C == 3))
        assert nnkernels.conv_stats_rows(geom) > 0, "this geometry must take the fused-statistics path"
        y.backward(gy.cuda())
    finally:
        pass
    assert torch.allclose(y.cpu().double(), yr.detach(), rtol=1e-4, atol=2e-4)
    assert torch.allclose(bn_g.running_mean.cpu().double(), bn.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn_g.running_var.cpu().double(), bn.running_var, rtol=1e-4, atol=1e-5)
    # gradients through BatchNorm are differences of large sums (the filter gradient of a conv followed by BN sums to ~0 per
    # output channel): tolerance relative to the largest entry, fp32 accumulation over N*Ho*Wo pixels
    for name, a, b in (("dw", conv_g.weight.grad, conv.weight.grad), ("dgamma", bn_g.weight.grad, bn.weight.grad),
                       ("dbeta", bn_g.bias.grad, bn.bias.grad)):
        assert float((a.cpu().double() - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-6, name
    if C > 3:
        assert float((xg.grad.cpu().double() - xr.grad).abs().max()) <= 1e-3 * float(xr.grad.abs().max())


@pytest.mark.parametrize("M,C,K,bias", [(5120, 768, 3072, True), (1284, 3072, 768, True), (600, 264, 2000, False)])
def test_weight_gradient_as_forward_gemm_on_transposed_operands(M, C, K, bias):
    """Plan impl 5 (nnkernels._wgrad_transposed): dW = dY^T X through sqd_transpose2d x 2 + sqd_conv_fwd, the bias gradient from the column
    sums the transpose takes on the way — against float64 and against the direct fp32 kernel (same error level), row counts that are
    no multiples of the 64 x 64 transpose tiles included."""
    from sqd import nnkernels
    torch.manual_seed(M + K)
    x = torch.randn(M, C, 1, 1)
    w = torch.randn(K, C, 1, 1) * 0.05
    b = torch.randn(K) if bias else None
    g = torch.randn(M, K, 1, 1)
    xr, wr = x.double(), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    F.conv2d(xr, wr, br).backward(g.double())
    geom = (M, 1, 1, C, K, 1, 1, 1, 0, 1, 1)
    assert nnkernels.wgrad_transposed_applies(geom)
    got = {}
    nnkernels.reset_plans()
    try:
        direct = 1 if C % 16 == 0 and K % 16 == 0 else 0          # (the direct-operand kernel wants multiples of 16: else the LDS-tiled one)
        for impl in (nnkernels.WGRAD_TRANSPOSED, 1):
            nnkernels._register_wgrad_plan((M, 1, 1, C, K, 1, 1), (impl, 0) if impl == nnkernels.WGRAD_TRANSPOSED else (direct, 8))
            xg = x.cuda().contiguous(memory_format=torch.channels_last)
            wg = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            bg = b.cuda().requires_grad_(True) if bias else None
            y = nnkernels.Conv2d.apply(xg, wg, bg, 1, 0, None, False, None, None)
            y.backward(g.cuda())
            got[impl] = (wg.grad.cpu().double(), bg.grad.cpu().double() if bias else None)
    finally:
        nnkernels.reset_plans()
    for i, (name, ref) in enumerate((("dw", wr.grad), ("db", br.grad if bias else None))):
        if ref is None:
            continue
        scale = float(ref.abs().max())
        et = float((got[nnkernels.WGRAD_TRANSPOSED][i] - ref).abs().max()) / scale
        ed = float((got[1][i] - ref).abs().max()) / scale
        assert et <= 2e-5 and et <= 4.0 * ed + 1e-6, (name, "transposed", et, "direct", ed)


@pytest.mark.parametrize("N,C,H,W,K,R,z,bk", [(2, 512, 6, 20, 512, 3, 4, 16), (3, 256, 9, 13, 72, 1, 2, 16), (2, 512, 12, 20, 320, 1, 2, 32 + 1024)])
def test_split_plan_takes_the_batchnorm_statistics_in_its_sum(N, C, H, W, K, R, z, bk):
    """A forward plan that splits the reduction: the sum over its splits (gemm_reduce_stats_kernel) writes the BatchNorm partials —
    ceil(M / 64) rows — and conv -> BatchNorm -> ReLU gives what the unsplit plan gives (same fp32 level against float64)."""
    from sqd import lib, nnkernels, nnops
    L = lib.lib()
    torch.manual_seed(K + z)
    conv, bn = nn.Conv2d(C, K, R, 1, R // 2, bias=False), nn.BatchNorm2d(K)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(N, C, H, W)
    cr, br = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    xr = x.double().requires_grad_(True)
    yr = F.relu(br(cr(xr)))
    gy = torch.randn_like(yr)
    yr.backward(gy)
    geom = (N, H, W, C, K, R, R, 1, R // 2, H, W)
    nnkernels.reset_plans()
    got = {}
    try:
        for zz in (z, 1):
            nnkernels._PLAN_CACHE.clear()
            assert L.sqd_conv_set_plan(0, *geom, 64, 64, zz, bk) == 0, L.sqd_last_error()
            cg, bg = copy.deepcopy(conv).cuda().to(memory_format=torch.channels_last), copy.deepcopy(bn).cuda()
            xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = nnops.conv_bn_act(xg, cg, bg, "relu")
            rows = nnkernels.conv_stats_rows(geom)
            assert rows == ((N * H * W + 63) // 64 if zz > 1 else (N * H * W + 63) // 64), rows
            y.backward(gy.float().cuda())
            got[zz] = (y.detach().cpu().double(), bg.running_mean.cpu().double(), bg.running_var.cpu().double(), xg.grad.cpu().double(),
                       bg.weight.grad.cpu().double(), bg.bias.grad.cpu().double())
    finally:
        L.sqd_conv_set_plan(0, *geom, 0, 0, 0, 16)
        nnkernels.reset_plans()
    refs = (yr.detach(), br.running_mean, br.running_var, xr.grad, br.weight.grad, br.bias.grad)
    for i, name in enumerate(("y", "running_mean", "running_var", "dx", "dgamma", "dbeta")):
        scale = float(refs[i].abs().max())
        es, eu = float((got[z][i] - refs[i]).abs().max()) / scale, float((got[1][i] - refs[i]).abs().max()) / scale
        assert es <= 2e-4 and es <= 4.0 * eu + 2e-6, (name, "split", es, "unsplit", eu)


@pytest.mark.parametrize("N,C,H,W,K,R,stride,pad,bias,act", [CASES[1], CASES[2], CASES[5], CASES[7], CASES[11], CASES[12]])
def test_split_precision_variant(N, C, H, W, K, R, stride, pad, bias, act):
    """opt-in arithmetic of forward / data gradient (sqd_conv_set_precision(1)): every fp32 operand as three bf16 terms on the
    bf16 matrix cores, 6 partial products, fp32 accumulation.  Must meet the same bar as the fp32 MFMA path — and, being an
    exact decomposition up to 2^-24, land within a few fp32 roundings of it."""
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(C + K + R)
    conv = nn.Conv2d(C, K, R, stride, pad, bias=bias).double()
    x = torch.randn(N, C, H, W)
    xr = x.double().requires_grad_(True)
    yr = conv(xr)
    wgt = torch.randn(yr.shape)
    (yr * wgt.double()).sum().backward()
    res = {}
    try:
        for prec in (0, 1):
            nnkernels.set_conv_precision(prec)
            assert L.sqd_conv_precision() == prec
            conv_g = nn.Conv2d(C, K, R, stride, pad, bias=bias).cuda()
            conv_g.load_state_dict({k: v.float() for k, v in conv.state_dict().items()})
            conv_g = conv_g.to(memory_format=torch.channels_last)
            xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = nnkernels.conv2d_native(xg, conv_g, None)
            (y * wgt.cuda()).sum().backward()
            res[prec] = (y.detach().cpu().double(), xg.grad.cpu().double())
    finally:
        nnkernels.set_conv_precision(0)
    for name, i, ref in (("y", 0, yr.detach()), ("dx", 1, xr.grad)):
        scale = float(ref.abs().max())
        e32, ebf = float((res[0][i] - ref).abs().max()), float((res[1][i] - ref).abs().max())
        assert ebf <= 1e-4 * scale, (name, ebf, scale)
        assert ebf <= 4.0 * e32 + 1e-7 * scale, (name, "split precision", ebf, "fp32 MFMA", e32)


@pytest.mark.parametrize("kt,ct", [(2, 4), (4, 2), (2, 2), (1, 1), (4, 4)])
def test_wgrad_register_tile_shapes(kt, ct):
    """every register-tile shape of the direct-operand weight-gradient kernel the tuner may register (impl 1 + 16 kt + 256 ct)"""
    from sqd import lib, nnkernels
    L = lib.lib()
    N, C, H, W, K, R, stride, pad = 4, 64, 24, 40, 128, 3, 1, 1
    torch.manual_seed(kt * 10 + ct)
    conv = nn.Conv2d(C, K, R, stride, pad, bias=True)
    x = torch.randn(N, C, H, W)
    xr = x.double().requires_grad_(True)
    conv64 = nn.Conv2d(C, K, R, stride, pad, bias=True).double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    yr = conv64(xr)
    wgt = torch.randn(yr.shape)
    (yr * wgt.double()).sum().backward()
    geom = (N, H, W, C, K, R, R, stride, pad, H, W)
    try:
        assert L.sqd_conv_wgrad_set_plan(N, H, W, C, K, R, R, 1 | (kt << 4) | (ct << 8), 6) == 0
        nnkernels._PLAN_CACHE.clear()
        conv_g = nn.Conv2d(C, K, R, stride, pad, bias=True).cuda()
        conv_g.load_state_dict(conv.state_dict())
        conv_g = conv_g.to(memory_format=torch.channels_last)
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = nnkernels.conv2d_native(xg, conv_g, None)
        (y * wgt.cuda()).sum().backward()
    finally:
        L.sqd_conv_wgrad_set_plan(N, H, W, C, K, R, R, -1, 0)
        nnkernels._PLAN_CACHE.clear()
    for name, a, b in (("dw", conv_g.weight.grad, conv64.weight.grad), ("db", conv_g.bias.grad, conv64.bias.grad)):
        err, scale = float((a.cpu().double() - b).abs().max()), float(b.abs().max())
        assert err <= 1e-4 * scale, (name, err, scale)
    assert L.sqd_conv_wgrad_set_plan(N, H, W, C, 48, R, R, 1 | (4 << 4) | (4 << 8), 6) != 0       # 64-filter tile on K = 48: refused


@pytest.mark.parametrize("N,C,H,W,K,R,stride,pad", [(2, 64, 24, 40, 64, 3, 1, 1), (2, 144, 20, 28, 24, 1, 1, 0), (2, 128, 24, 40, 128, 3, 2, 1),
                                                     (1, 512, 6, 20, 512, 3, 1, 1)])
def test_conv_bf16_operand_mode(N, C, H, W, K, R, stride, pad):
    """sqd_conv_set_precision(2) — operands rounded to nearest-even bf16 when staged, fp32 accumulation (BASELINE.json configs[3]):
    forward and data gradient equal an fp32 convolution of the bf16-rounded operands to accumulation-order rounding; the weight
    gradient stays fp32"""
    from sqd import lib, nnkernels
    torch.manual_seed(C + K)
    conv = nn.Conv2d(C, K, R, stride, pad, bias=False)
    x = torch.randn(N, C, H, W)
    rb = lambda t: t.bfloat16().float()
    xr = rb(x).requires_grad_(True)
    wr = rb(conv.weight.detach()).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    g = torch.randn_like(yr)
    # the data gradient multiplies bf16(dy) with bf16(w)
    (dx_ref,) = torch.autograd.grad(F.conv2d(xr, wr, None, stride, pad), xr, rb(g))
    conv_g = nn.Conv2d(C, K, R, stride, pad, bias=False).cuda()
    conv_g.load_state_dict(conv.state_dict())
    conv_g = conv_g.to(memory_format=torch.channels_last)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    L = lib.lib()
    nnkernels.set_conv_precision(2)
    try:
        y = nnkernels.conv2d_native(xg, conv_g, None)
        (y * g.cuda()).sum().backward()
    finally:
        nnkernels.set_conv_precision(0)

    def close(a, b, name, rtol):
        a, b = a.detach().cpu().float(), b.detach().float()
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        assert err <= rtol * scale + 1e-6, (name, err, scale)
    close(y, yr, "y", 2e-5)
    close(xg.grad, dx_ref, "dx", 2e-5)
    # weight gradient: full fp32 of the UNROUNDED operands
    x0 = x.clone().requires_grad_(True)
    (F.conv2d(x0, conv.weight, None, stride, pad) * g).sum().backward()
    close(conv_g.weight.grad, conv.weight.grad, "dw", 5e-4)


@pytest.mark.parametrize("variant,N,C,H,W,K,R,stride,pad,bias,splits", [
    (0, 2, 128, 24, 40, 256, 3, 1, 1, False, 5), (0, 3, 256, 13, 21, 128, 1, 1, 0, True, 3), (1, 2, 128, 24, 40, 64, 3, 2, 1, False, 4),
    (2, 2, 64, 24, 40, 128, 3, 1, 1, True, 7), (3, 2, 64, 23, 37, 64, 5, 2, 2, False, 2), (3, 1, 192, 9, 11, 64, 3, 1, 1, False, 64)])
@pytest.mark.parametrize("impl", [2, 3])
def test_wgrad_shared_operand_kernel(impl, variant, N, C, H, W, K, R, stride, pad, bias, splits):
    """the shared-operand weight-gradient kernels (impl 2: fp32 MFMA, dy / x rows of 32 pixels staged once per workgroup; impl 3: the
    three-term bf16 kernel on the same blocks; + 16 * variant) against float64 — ragged pixel ranges, strides, padding, more
    splits than pixels allow."""
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(variant + K)
    conv = nn.Conv2d(C, K, R, stride, pad, bias=bias)
    x = torch.randn(N, C, H, W)
    conv64 = nn.Conv2d(C, K, R, stride, pad, bias=bias).double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    yr = conv64(x.double())
    wgt = torch.randn(yr.shape)
    (yr * wgt.double()).sum().backward()
    Ho, Wo = yr.shape[2:]
    try:
        assert L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, impl | (variant << 4), splits) == 0
        nnkernels._PLAN_CACHE.clear()
        conv_g = nn.Conv2d(C, K, R, stride, pad, bias=bias).cuda()
        conv_g.load_state_dict(conv.state_dict())
        conv_g = conv_g.to(memory_format=torch.channels_last)
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = nnkernels.conv2d_native(xg, conv_g, None)
        (y * wgt.cuda()).sum().backward()
    finally:
        L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, -1, 0)
        nnkernels._PLAN_CACHE.clear()
    pairs = [("dw", conv_g.weight.grad, conv64.weight.grad)] + ([("db", conv_g.bias.grad, conv64.bias.grad)] if bias else [])
    for name, a, b in pairs:
        err, scale = float((a.cpu().double() - b).abs().max()), float(b.abs().max())
        assert err <= 1e-4 * scale, (name, err, scale)
    assert L.sqd_conv_wgrad_set_plan(N, Ho, Wo, 48, K, R, R, impl, 4) != 0         # 128-channel block on C = 48: refused


@pytest.mark.parametrize("variant,N,C,H,W,K,R,stride,pad,bias,splits", [
    (0, 2, 128, 24, 40, 256, 3, 1, 1, False, 5), (0, 3, 256, 13, 21, 128, 1, 1, 0, True, 3), (0, 2, 128, 24, 40, 64, 3, 2, 1, False, 4),
    (1, 2, 32, 24, 40, 128, 3, 1, 1, True, 7), (2, 2, 64, 24, 36, 32, 5, 2, 2, False, 2), (0, 1, 192, 9, 12, 64, 3, 1, 1, False, 64),
    (0, 2, 64, 12, 40, 128, 1, 2, 0, False, 2), (2, 1, 64, 7, 9, 96, 1, 1, 0, False, 1),
    (3, 2, 64, 24, 40, 128, 3, 1, 1, False, 3), (4, 2, 128, 13, 21, 64, 1, 1, 0, True, 2), (5, 2, 128, 12, 20, 256, 3, 2, 1, False, 2),
    (5, 1, 256, 9, 11, 128, 1, 1, 0, False, 5), (6, 2, 64, 24, 40, 128, 3, 1, 1, False, 3), (6, 3, 128, 13, 21, 256, 1, 1, 0, True, 2),
    (7, 2, 64, 23, 36, 64, 3, 2, 1, False, 2), (7, 1, 64, 9, 11, 64, 1, 1, 0, False, 1)])
def test_wgrad_three_term_direct_kernel(variant, N, C, H, W, K, R, stride, pad, bias, splits):
    """impl 6 (round 4): three-term bf16 operands straight from memory — the wave's halves take the even / odd pixel of a pair, a lane
    its consecutive channels of one pixel as one load — against float64: 3x3 / 5x5 / 1x1, strides, padding (the pair that straddles the
    image border), a ragged last pixel range, an odd pixel count (plain 1x1), more splits than the pixels allow; variants 6 / 7: eight waves per workgroup."""
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(variant + K)
    conv = nn.Conv2d(C, K, R, stride, pad, bias=bias)
    x = torch.randn(N, C, H, W)
    conv64 = nn.Conv2d(C, K, R, stride, pad, bias=bias).double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    yr = conv64(x.double())
    wgt = torch.randn(yr.shape)
    (yr * wgt.double()).sum().backward()
    Ho, Wo = yr.shape[2:]
    try:
        assert L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, 6 | (variant << 4), splits) == 0
        nnkernels._PLAN_CACHE.clear()
        conv_g = nn.Conv2d(C, K, R, stride, pad, bias=bias).cuda()
        conv_g.load_state_dict(conv.state_dict())
        conv_g = conv_g.to(memory_format=torch.channels_last)
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = nnkernels.conv2d_native(xg, conv_g, None)
        (y * wgt.cuda()).sum().backward()
    finally:
        L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, -1, 0)
        nnkernels._PLAN_CACHE.clear()
    pairs = [("dw", conv_g.weight.grad, conv64.weight.grad)] + ([("db", conv_g.bias.grad, conv64.bias.grad)] if bias else [])
    for name, a, b in pairs:
        err, scale = float((a.cpu().double() - b).abs().max()), float(b.abs().max())
        assert err <= 1e-4 * scale, (name, err, scale)
    assert L.sqd_conv_wgrad_set_plan(N, Ho, Wo, 48, K, R, R, 6, 4) != 0            # 64-channel tile on C = 48: refused
    assert L.sqd_conv_wgrad_set_plan(N, 7, 9, C, K, 3, 3, 6 | (variant << 4), 4) != 0     # odd Wo under a 3x3: pairs would straddle rows, refused


def test_wgrad_plan_shared_by_output_geometry():
    """Two convolutions with the same (N, Ho, Wo, C, K, R, S) and different strides share one library plan: tuning the second must
    not leave the first with a stale workspace size (regression: out-of-bounds partial sums on the side-stream launch)."""
    import ctypes
    from sqd import lib, nnkernels
    L = lib.lib()
    N, C, K, R = 2, 128, 128, 3
    gA = (N, 24, 40, C, K, R, R, 1, 1, 24, 40)          # stride 1
    gB = (N, 48, 80, C, K, R, R, 2, 1, 24, 40)          # stride 2, same output
    try:
        for g in (gA, gB):
            nnkernels._TUNED.discard(nnkernels._wgrad_key(g))
        for g, H, W, st in ((gA, 24, 40, 1), (gB, 48, 80, 2)):
            x = torch.randn(N, H, W, C, device="cuda")
            dy = torch.randn(N, 24, 40, K, device="cuda")
            dw = torch.empty(K, R, R, C, device="cuda")
            P = lambda t: ctypes.c_void_p(t.data_ptr())
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            nnkernels._tune_wgrad(g, False, lambda part: L.sqd_conv_wgrad(P(dy), P(x), P(dw), None, P(part), *g, stream))
            sp, pf = ctypes.c_int(0), ctypes.c_int64(0)
            L.sqd_conv_wgrad_plan(N, 24, 40, C, K, R, R, ctypes.byref(sp), ctypes.byref(pf))
            for g2 in (gA, gB):                           # both see the library's current plan
                assert nnkernels._wgrad_part_floats(g2) == (pf.value, sp.value)
    finally:
        L.sqd_conv_wgrad_set_plan(N, 24, 40, C, K, R, R, -1, 0)
        nnkernels._PLAN_CACHE.clear()


@pytest.mark.parametrize("N,C,H,W,K", [(2, 64, 24, 40, 64), (1, 40, 9, 11, 96), (3, 72, 17, 33, 200), (2, 256, 6, 20, 128), (1, 8, 5, 50, 64), (2, 32, 20, 36, 16),
                                       (1, 16, 9, 40, 32), (2, 96, 13, 16, 24)])
def test_input_patch_plans(N, C, H, W, K):
    """The 3x3 / stride 1 / pad 1 kernel that stages the input patch once per channel chunk (bk = 32 + 1024 + 2048): every patch /
    channel-tile shape and channel split against float64 — forward with bias, the data gradient with the second gradient added in
    its epilogue, image sizes that leave partial patches, channel counts that leave partial chunks and partial channel tiles."""
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(C + K)
    conv = nn.Conv2d(C, K, 3, 1, 1, bias=True).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    geom = (N, H, W, C, K, 3, 3, 1, 1, H, W)
    conv64 = nn.Conv2d(C, K, 3, 1, 1, bias=True).cuda().double()
    conv64.load_state_dict(conv.state_dict())
    xr = x.double().requires_grad_(True)
    yr = conv64(xr)
    gy = torch.randn_like(yr)
    gskip = torch.randn_like(xr)
    gxr, = torch.autograd.grad(yr, xr, gy)
    gxr = gxr + gskip
    tried = 0
    try:
        for bm, bn, z, w8 in ((bm, bn, z, w8) for bm in (128, 64) for bn in (128, 64, 32) for z in (1, 2, 3) for w8 in (0, 256)):
            # (+ 256: the 8-wave workgroups of the same tiles)
            ok = [L.sqd_conv_set_plan(mode, *geom, bm, bn, z, 32 + 1024 + 2048 + w8) == 0 for mode in (0, 1)]
            if not any(ok):
                continue
            for mode in (0, 1):
                if not ok[mode]:
                    L.sqd_conv_set_plan(mode, *geom, 0, 0, 0, 16)
            nnkernels._PLAN_CACHE.clear()
            xg = x.clone().requires_grad_(True)
            y, xs = nnkernels.conv2d_native(xg, conv, None, True)
            gx, = torch.autograd.grad((y, xs), xg, (gy.float(), gskip.float()))
            tried += 1
            ey = float((y.detach().double() - yr.detach()).abs().max()) / float(yr.abs().max())
            ex = float((gx.double() - gxr).abs().max()) / float(gxr.abs().max())
            assert ey <= 4e-6 and ex <= 4e-6, (bm, bn, z, w8, ok, ey, ex)
    finally:
        for mode in (0, 1):
            L.sqd_conv_set_plan(mode, *geom, 0, 0, 0, 16)
        nnkernels._PLAN_CACHE.clear()
    assert tried >= 2
    assert L.sqd_conv_set_plan(0, N, H, W, C, K, 3, 3, 2, 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 128, 64, 1, 32 + 1024 + 2048) != 0    # stride 2: refused


@pytest.mark.parametrize("K,plans", [(96, ((128, 64), (64, 128), (128, 128))), (32, ((128, 32), (64, 32))), (16, ((128, 32), (64, 32)))])
def test_input_patch_plan_feeds_batchnorm_statistics(K, plans):
    """conv -> BatchNorm(train) -> ReLU with the per-patch statistics partials of the input-patch kernel (32-channel tiles: the waves
    that split the patch rows add their column sums through LDS)"""
    from sqd import lib, nnkernels, nnops
    L = lib.lib()
    N, C, H, W = 3, 64, 21, 37
    torch.manual_seed(5)
    conv, bn = nn.Conv2d(C, K, 3, 1, 1, bias=False), nn.BatchNorm2d(K)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(N, C, H, W)
    conv_g, bn_g = nn.Conv2d(C, K, 3, 1, 1, bias=False).cuda(), nn.BatchNorm2d(K).cuda()
    conv_g.load_state_dict(conv.state_dict())
    bn_g.load_state_dict(bn.state_dict())
    conv_g = conv_g.to(memory_format=torch.channels_last)
    conv, bn = conv.double(), bn.double()
    xr = x.double().requires_grad_(True)
    yr = F.relu(bn(conv(xr)))
    gy = torch.randn_like(yr)
    yr.backward(gy)
    geom = (N, H, W, C, K, 3, 3, 1, 1, H, W)
    try:
        for bm, bn_t in plans:
            assert L.sqd_conv_set_plan(0, *geom, bm, bn_t, 1, 32 + 1024 + 2048) == 0
            nnkernels._PLAN_CACHE.clear()
            bn_g.load_state_dict({k: v.float() for k, v in nn.BatchNorm2d(K).state_dict().items()} | {"weight": bn.weight.float(), "bias": bn.bias.float()})
            xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = nnops.conv_bn_act(xg, conv_g, bn_g, "relu")
            assert nnkernels.conv_stats_rows(nnkernels.conv_out_geom(xg, conv_g)) == N * ((H + bm // 16 - 1) // (bm // 16)) * ((W + 15) // 16)
            conv_g.weight.grad = None
            y.backward(gy.float().cuda())
            assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=1e-4, atol=2e-4), (bm, bn_t, float((y.detach().cpu().double() - yr.detach()).abs().max()))
            assert torch.allclose(bn_g.running_var.cpu().double(), bn.running_var, rtol=1e-4, atol=1e-5), (bm, bn_t)
            ex = float((xg.grad.cpu().double() - xr.grad).abs().max()) / float(xr.grad.abs().max())
            assert ex <= 1e-3, (bm, bn_t, ex)
    finally:
        L.sqd_conv_set_plan(0, *geom, 0, 0, 0, 16)
        nnkernels._PLAN_CACHE.clear()


def test_planar_frame_staging_matches_the_copies():
    """sqd_space_to_depth2_planar (frame pairs, normalisation and layout conversion inside the stem's space-to-depth pass) against the
    staged path it replaces — torch.cat + channels-last copy + (x - a) / b + sqd_space_to_depth2: same bits, forward and filter gradient."""
    from sqd import nnkernels, nnops
    from networks.pose_cnn import PoseCNN
    torch.manual_seed(3)
    B, H, W = 3, 32, 48
    f = [torch.rand(B, 3, H, W, device="cuda") for _ in range(3)]
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    xn = ((f[0] - 0.45) / 0.225).contiguous(memory_format=torch.channels_last)
    y_ref = nnkernels.conv2d_stem_s2d(xn, conv, "relu")
    y = nnkernels.conv2d_stem_s2d_planar([(f[0], None)], conv, "relu", None, (0.45, 0.225))
    assert torch.equal(y, y_ref)
    g = torch.randn_like(y)
    gw_ref, = torch.autograd.grad(y_ref, conv.weight, g)
    gw, = torch.autograd.grad(y, conv.weight, g)
    assert torch.equal(gw, gw_ref)
    pose = PoseCNN(2).cuda().to(memory_format=torch.channels_last)
    pairs = [(f[1], f[0]), (f[0], f[2])]
    x = torch.stack([torch.cat(p, 1) for p in pairs], 1).reshape(B * 2, 6, H, W).contiguous(memory_format=torch.channels_last)
    try:
        a_ref, t_ref = pose(x)
        a, t = pose.forward_pairs(pairs)
    finally:
        pass
    assert torch.equal(a, a_ref) and torch.equal(t, t_ref)


@pytest.mark.parametrize("N,C,H,W,K", [(2, 3, 32, 48, 64), (3, 6, 30, 44, 16), (1, 3, 18, 34, 32)])
def test_input_patch_plans_on_the_stems(N, C, H, W, K):
    """The 7x7 / stride 2 stems run as 4x4 / stride 1 / pad 2 convolutions on the space-to-depth image; with the input-patch plans
    (16 taps read from one staged patch) against float64, bias + ReLU, every accepted patch / channel-tile shape."""
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(C * K)
    conv = nn.Conv2d(C, K, 7, 2, 3, bias=True).cuda()
    x = torch.randn(N, C, H, W, device="cuda")
    yr = F.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), 2, 3))
    Cp = (4 * C + 15) // 16 * 16
    geom = (N, H // 2, W // 2, Cp, K, 4, 4, 1, 2, H // 2, W // 2)
    tried = 0
    try:
        for bm in (128, 64):
            for bn in (64, 32):
                if L.sqd_conv_set_plan(0, *geom, bm, bn, 1, 32 + 1024 + 2048) != 0:
                    continue
                nnkernels._PLAN_CACHE.clear()
                for y in (nnkernels.conv2d_stem_s2d(x.contiguous(memory_format=torch.channels_last), conv, "relu"),
                          nnkernels.conv2d_stem_s2d_planar([(x, None)], conv, "relu")):
                    err = float((y.detach().double() - yr).abs().max()) / float(yr.abs().max())
                    assert err <= 4e-6, (bm, bn, err)
                tried += 1
    finally:
        L.sqd_conv_set_plan(0, *geom, 0, 0, 0, 16)
        nnkernels._PLAN_CACHE.clear()
    assert tried >= 2
    assert L.sqd_conv_set_plan(1, *geom, 64, 32, 1, 32 + 1024 + 2048) != 0         # no data gradient for the 4x4 form


@pytest.mark.parametrize("bm,bn_t", [(128, 64), (64, 64), (128, 32), (64, 32)])
def test_stem_with_input_patch_plan_feeds_batchnorm_statistics(bm, bn_t):
    """the encoder stem as the Trainer runs it: planar frame -> (x - 0.45) / 0.225 -> 7x7/2 convolution (4x4 input-patch plan on the
    space-to-depth image, waves splitting the patch rows) -> BatchNorm(train) from the epilogue's per-patch partials -> ReLU"""
    from sqd import lib, nnkernels, nnops
    L = lib.lib()
    N, H, W, K = 3, 44, 72, 64 if bn_t == 64 else 32
    torch.manual_seed(bm + bn_t)
    conv, bn = nn.Conv2d(3, K, 7, 2, 3, bias=False), nn.BatchNorm2d(K)
    x = torch.rand(N, 3, H, W)
    conv_g, bn_g = nn.Conv2d(3, K, 7, 2, 3, bias=False).cuda(), nn.BatchNorm2d(K).cuda()
    conv_g.load_state_dict(conv.state_dict())
    conv, bn = conv.double(), bn.double()
    yr = F.relu(bn(conv((x.double() - 0.45) / 0.225)))
    gy = torch.randn_like(yr)
    gwr, = torch.autograd.grad(yr, conv.weight, gy)
    geom = (N, H // 2, W // 2, 16, K, 4, 4, 1, 2, H // 2, W // 2)
    try:
        assert L.sqd_conv_set_plan(0, *geom, bm, bn_t, 1, 32 + 1024 + 2048) == 0
        nnkernels._PLAN_CACHE.clear()
        y = nnops.conv_bn_act(x.cuda(), conv_g, bn_g, "relu", input_affine=(0.45, 0.225))
        assert nnkernels.conv_stats_rows(geom) == N * (((H // 2) + bm // 16 - 1) // (bm // 16)) * (((W // 2) + 15) // 16)
        gw, = torch.autograd.grad(y, conv_g.weight, gy.float().cuda())
    finally:
        L.sqd_conv_set_plan(0, *geom, 0, 0, 0, 16)
        nnkernels._PLAN_CACHE.clear()
    assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=1e-4, atol=2e-4)
    assert torch.allclose(bn_g.running_var.cpu().double(), bn.running_var, rtol=1e-4, atol=1e-5)
    assert float((gw.cpu().double() - gwr).abs().max()) <= 1e-3 * float(gwr.abs().max())


@pytest.mark.parametrize("N,C,H,W,K,R,pad,bias,splits", [
    (2, 96, 24, 80, 16, 3, 1, True, 40), (1, 32, 17, 45, 16, 4, 2, False, 7), (2, 16, 33, 70, 64, 4, 2, True, 1000),
    (2, 32, 24, 40, 32, 3, 1, True, 16), (1, 16, 9, 31, 32, 3, 1, False, 3), (3, 16, 12, 64, 16, 3, 1, True, 5),
    (1, 16, 20, 33, 64, 3, 0, False, 9), (2, 32, 8, 100, 16, 3, 1, False, 1)])
def test_wgrad_row_window_kernel(N, C, H, W, K, R, pad, bias, splits):
    """the row-window weight-gradient kernel (impl 4: whole filter bank per workgroup, x rows in an LDS ring) against float64 —
    ragged column chunks, column runs that start mid-image, more workgroups than steps, padding 0 / 1 / 2, 3x3 and 4x4 taps."""
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(C + K + R)
    conv = nn.Conv2d(C, K, R, 1, pad, bias=bias)
    x = torch.randn(N, C, H, W)
    conv64 = nn.Conv2d(C, K, R, 1, pad, bias=bias).double()
    conv64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    yr = conv64(x.double())
    wgt = torch.randn(yr.shape)
    (yr * wgt.double()).sum().backward()
    Ho, Wo = yr.shape[2:]
    try:
        assert L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, 4, splits) == 0
        nnkernels._PLAN_CACHE.clear()
        conv_g = nn.Conv2d(C, K, R, 1, pad, bias=bias).cuda()
        conv_g.load_state_dict(conv.state_dict())
        conv_g = conv_g.to(memory_format=torch.channels_last)
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = nnkernels.conv2d_native(xg, conv_g, None)
        (y * wgt.cuda()).sum().backward()
        if R == 3:           # a strided convolution that shares the plan key runs the direct kernel inside the same workspace
            conv_s = nn.Conv2d(C, K, R, 2, pad, bias=bias).cuda().to(memory_format=torch.channels_last)
            conv_s.load_state_dict(conv.state_dict())
            xs = torch.randn(N, C, 2 * Ho + R - 2 * pad - 2, 2 * Wo + R - 2 * pad - 2)
            ys64 = F.conv2d(xs.double(), conv64.weight.detach(), None, 2, pad)
            assert tuple(ys64.shape[2:]) == (Ho, Wo)
            w64 = conv64.weight.detach().clone().requires_grad_(True)
            (F.conv2d(xs.double(), w64, None, 2, pad) * wgt.double()).sum().backward()
            ysg = nnkernels.conv2d_native(xs.cuda().contiguous(memory_format=torch.channels_last), conv_s, None)
            (ysg * wgt.cuda()).sum().backward()
            err, scale = float((conv_s.weight.grad.cpu().double() - w64.grad).abs().max()), float(w64.grad.abs().max())
            assert err <= 1e-4 * scale, ("strided dw", err, scale)
    finally:
        L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, R, -1, 0)
        nnkernels._PLAN_CACHE.clear()
    pairs = [("dw", conv_g.weight.grad, conv64.weight.grad)] + ([("db", conv_g.bias.grad, conv64.bias.grad)] if bias else [])
    for name, a, b in pairs:
        err, scale = float((a.cpu().double() - b).abs().max()), float(b.abs().max())
        assert err <= 1e-4 * scale, (name, err, scale)
    assert L.sqd_conv_wgrad_set_plan(N, Ho, Wo, 64, 64, 3, 3, 4, 4) != 0          # 64 x 64 x 9: not instantiated, refused
    assert L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, 5, 5, 4, 4) != 0            # 5x5 taps: refused


@pytest.mark.parametrize("N,C,H,W,K,R,stride,pad", [(4, 32, 24, 40, 64, 3, 1, 1), (12, 8, 48, 160, 16, 7, 2, 3), (2, 64, 24, 40, 128, 1, 1, 0),
                                                     (12, 256, 6, 20, 256, 3, 2, 1)])
def test_filter_and_bias_split_sums_in_one_launch_are_the_same_bits(N, C, H, W, K, R, stride, pad):
    """sqd_conv_wgrad sums the filter partials and (where the plan left bias partials behind them) the bias partials in ONE launch;
    sqd_conv_wgrad_partials + sqd_split_reduce is the two-launch form of the same arithmetic: equal bits, filter and bias."""
    import ctypes
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(N + K)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = torch.randn(N, H, W, C, device="cuda")
    dy = torch.randn(N, Ho, Wo, K, device="cuda")
    geom = (N, H, W, C, K, R, R, stride, pad, Ho, Wo)
    pf, splits = nnkernels._wgrad_part_floats(geom)                       # the workspace Conv2d.backward allocates: filter partials + bias rows
    nfl = pf + max((N * Ho * Wo + 1023) // 1024, splits) * K
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for partial in (False, True):
        part = torch.zeros(nfl, device="cuda")
        dw = torch.full((K, R, R, C), float("nan"), device="cuda")
        db = torch.full((K,), float("nan"), device="cuda")
        if partial:
            sp = ctypes.c_int(0)
            lib.check(L.sqd_conv_wgrad_partials(P(dy), P(x), P(dw), P(db), P(part), *geom, ctypes.byref(sp), st), "wgrad_partials")
            lib.check(L.sqd_split_reduce(P(part), P(dw), K * R * R * C, sp.value, st), "split_reduce")
        else:
            lib.check(L.sqd_conv_wgrad(P(dy), P(x), P(dw), P(db), P(part), *geom, st), "wgrad")
        torch.cuda.synchronize()
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (K, C, R, R), dy.permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    assert torch.allclose(outs[0][0], ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()))
    assert torch.allclose(outs[0][1], dy.sum((0, 1, 2)), rtol=1e-4, atol=1e-4 * float(dy.sum((0, 1, 2)).abs().max()))
