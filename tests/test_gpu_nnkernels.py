"""GPU parity of the network-operator kernels (BatchNorm+activation+residual, ...) against PyTorch fp32
reference implementations of the same operators (the oracle's networks are built from exactly these
torch modules)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act(y, act):
    return F.relu(y) if act == "relu" else F.leaky_relu(y, 0.01) if act == "leaky_relu" else y


@pytest.mark.parametrize("N,C,H,W,act,with_res", [(2, 64, 12, 20, "relu", False), (3, 256, 6, 10, "relu", True),
                                                  (2, 16, 24, 40, "leaky_relu", False), (1, 2048, 3, 5, None, True),
                                                  (2, 32, 17, 9, "leaky_relu", True), (12, 64, 96, 320, "relu", False)])
def test_batchnorm_act_train_and_eval(N, C, H, W, act, with_res):
    from sqd import nnkernels
    torch.manual_seed(C + H)
    x = (torch.randn(N, C, H, W) * 1.5 + 0.3)
    res = torch.randn(N, C, H, W) if with_res else None
    wgt = torch.randn(N, C, H, W)
    ref_bn = nn.BatchNorm2d(C)
    with torch.no_grad():
        ref_bn.weight.uniform_(0.5, 1.5); ref_bn.bias.uniform_(-0.2, 0.2)
        ref_bn.running_mean.uniform_(-0.1, 0.1); ref_bn.running_var.uniform_(0.5, 1.5)
    my_bn = nn.BatchNorm2d(C).cuda()
    my_bn.load_state_dict(ref_bn.state_dict())
    # ---- training step
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    y_ref = ref_bn(xr)
    if with_res:
        y_ref = y_ref + rr
    y_ref = _act(y_ref, act)
    (y_ref * wgt).sum().backward()
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = res.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    y = nnkernels.batch_norm_act(xg, my_bn, act, rg)
    (y * wgt.cuda()).sum().backward()

    def close(a, b, name, rtol=1e-4):
        a, b = a.detach().cpu().float(), b.detach().float()
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        assert err <= rtol * scale + 1e-5, (name, err, scale)
    close(y, y_ref, "y")
    close(xg.grad, xr.grad, "dx", 5e-4)
    close(my_bn.weight.grad, ref_bn.weight.grad, "dgamma", 5e-4)
    close(my_bn.bias.grad, ref_bn.bias.grad, "dbeta", 5e-4)
    if with_res:
        close(rg.grad, rr.grad, "dres")
    close(my_bn.running_mean, ref_bn.running_mean, "running_mean")
    close(my_bn.running_var, ref_bn.running_var, "running_var")
    assert int(my_bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    # ---- eval
    ref_bn.eval(); my_bn.eval()
    with torch.no_grad():
        ye_ref = ref_bn(x)
        if with_res:
            ye_ref = ye_ref + res
        ye_ref = _act(ye_ref, act)
        ye = nnkernels.batch_norm_act(x.cuda(), my_bn, act, res.cuda() if with_res else None)
    close(ye, ye_ref, "y_eval")


@pytest.mark.parametrize("N,Cx,Hi,Wi,Cs,Ho,Wo", [(2, 32, 8, 22, 64, 12, 40), (2, 128, 12, 40, 512, 24, 80), (1, 16, 5, 7, 8, 10, 14),
                                                 (3, 8, 6, 20, 4, 6, 20), (2, 4, 1, 1, 4, 3, 5), (2, 32, 48, 160, 64, 96, 320)])
def test_upsample_concat(N, Cx, Hi, Wi, Cs, Ho, Wo):
    from sqd import nnkernels
    torch.manual_seed(Cx + Ho)
    x, skip = torch.randn(N, Cx, Hi, Wi), torch.randn(N, Cs, Ho, Wo)
    wgt = torch.randn(N, Cx + Cs, Ho, Wo)
    xr, sr = x.clone().requires_grad_(True), skip.clone().requires_grad_(True)
    ref = torch.cat([F.interpolate(xr, size=[Ho, Wo], mode="bilinear", align_corners=True), sr], 1)
    (ref * wgt).sum().backward()
    xg, sg = x.cuda().requires_grad_(True), skip.cuda().requires_grad_(True)
    out = nnkernels.UpsampleConcat.apply(xg, sg)
    (out * wgt.cuda()).sum().backward()
    for a, b, n in ((out, ref, "out"), (xg.grad, xr.grad, "g_x"), (sg.grad, sr.grad, "g_skip")):
        a, b = a.detach().cpu(), b.detach()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-5, n


def test_fused_adam_matches_torch_adam():
    from sqd.optim import FusedAdam
    torch.manual_seed(3)
    shapes = [(64, 3, 7, 7), (5,), (1000, 33), (4096,), (4097,), (2, 3), (128, 64, 3, 3)]
    ref_p = [torch.randn(s).cuda().requires_grad_(True) for s in shapes]
    my_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref_p[6].data = ref_p[6].data.contiguous(memory_format=torch.channels_last)
    my_p[6].data = my_p[6].data.contiguous(memory_format=torch.channels_last)
    unused_ref, unused_my = torch.randn(7).cuda().requires_grad_(True), torch.randn(7).cuda().requires_grad_(True)
    ref = torch.optim.Adam(ref_p + [unused_ref], lr=1e-2)
    mine = FusedAdam(my_p + [unused_my], lr=1e-2)
    for step in range(4):
        grads = [torch.randn(s).cuda() * (0.1 + step) for s in shapes]
        for p, q, g in zip(ref_p, my_p, grads):
            p.grad = g.clone().contiguous(memory_format=torch.channels_last) if p.dim() == 4 and not p.is_contiguous() else g.clone()
            q.grad = p.grad.clone()
        if step == 2:
            for grp in ref.param_groups + mine.param_groups:
                grp["lr"] = 1e-3                    # StepLR acts through param_groups
        ref.step(); mine.step()
        for p, q in zip(ref_p, my_p):
            assert float((p - q).abs().max()) <= 2e-6 * float(p.abs().max()) + 1e-7, step
    sd_ref, sd_my = ref.state_dict(), mine.state_dict()
    assert sd_ref["param_groups"][0]["params"] == sd_my["param_groups"][0]["params"]
    assert set(sd_ref["state"][0].keys()) == set(sd_my["state"][0].keys())
    assert float(sd_my["state"][0]["step"]) == 4.0 and 7 not in sd_my["state"]
    np.testing.assert_allclose(sd_my["state"][3]["exp_avg_sq"].cpu().numpy(), sd_ref["state"][3]["exp_avg_sq"].cpu().numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 12, 20), (1, 16, 7, 9), (3, 8, 1, 5), (12, 64, 96, 320)])
def test_maxpool3x3s2_bit_exact(N, C, H, W):
    """MaxPool2d(3,2,1) forward and (gather-form) backward against ATen on the CPU — including ties (quantised input)."""
    from sqd import nnkernels
    g = torch.Generator().manual_seed(N + C + H)
    x = torch.round(torch.randn(N, C, H, W, generator=g) * 4) / 4        # many exact ties inside the windows
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = nnkernels.MaxPool3x3s2.apply(xg)
    y.backward(gy.cuda())
    assert torch.equal(y.cpu(), yr.detach())
    assert torch.equal(xg.grad.cpu(), xr.grad)


def test_maxpool3x3s2_skip_adds_second_consumer_gradient():
    """skip=True: the input's second consumer reads x'; its gradient is added inside the gather kernel"""
    from sqd import nnkernels
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 9, 14, generator=g)
    gy, g2 = torch.randn(2, 16, 5, 7, generator=g), torch.randn(2, 16, 9, 14, generator=g)
    xr = x.clone().requires_grad_(True)
    (F.max_pool2d(xr, 3, 2, 1) * gy).sum().backward()
    ref = xr.grad + g2
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y, x2 = nnkernels.MaxPool3x3s2.apply(xg, True)
    assert torch.equal(x2, xg)
    ((y * gy.cuda()).sum() + (x2 * g2.cuda()).sum()).backward()
    assert torch.allclose(xg.grad.cpu(), ref, rtol=0, atol=1e-6)
    # only the second consumer carries a gradient
    xg2 = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y, x2 = nnkernels.MaxPool3x3s2.apply(xg2, True)
    (x2 * g2.cuda()).sum().backward()
    assert torch.equal(xg2.grad.cpu(), g2)


def test_encoder_taps_carry_the_same_values_and_gradients():
    """the feature taps handed through the next stage's nodes (resnet_encoder.ResnetEncoder.forward) against the plain wiring:
    same features, same parameter gradients when both the trunk and a decoder-like consumer use every tap"""
    from networks.resnet_encoder import ResnetEncoder
    from sqd import nnops
    torch.manual_seed(3)
    enc = ResnetEncoder(18).cuda().train()
    try:
        x = torch.rand(2, 3, 64, 96, device="cuda")
        ws = [torch.randn(1, c, 1, 1, device="cuda") for c in enc.num_ch_enc]

        def loss(feats):
            return sum((f * w).mean() for f, w in zip(feats, ws))

        feats = enc(x)
        loss(feats).backward()
        got = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
        vals = [f.detach().clone() for f in feats]
        enc.zero_grad(set_to_none=True)
        e = enc.encoder                                             # plain wiring: every tap consumed twice, autograd adds
        f0 = nnops.conv_bn_act(x, e.conv1, e.bn1, "relu", input_affine=(0.45, 0.225))
        f1 = e.layer1(nnops.maxpool3x3s2(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        plain = [f0, f1, f2, f3, f4]
        loss(plain).backward()
        for a, b in zip(vals, plain):
            assert torch.equal(a, b.detach())
        for n, p in enc.named_parameters():
            if p.grad is not None:
                err = (got[n] - p.grad).abs().max().item()
                assert err <= 1e-5 * max(p.grad.abs().max().item(), 1e-6), (n, err)
    finally:
        pass


@pytest.mark.parametrize("rows,K,N,act", [(12, 2048, 1024, "leaky_relu"), (12, 1024, 256, "leaky_relu"), (12, 256, 64, None),
                                          (2, 3840, 1920, "leaky_relu"), (8, 4096, 2048, "leaky_relu"), (1, 192, 32, None)])
def test_linear_native(rows, K, N, act):
    """the bins regressor's nn.Linear (+ LeakyReLU) through the 1x1 implicit-GEMM kernels against fp64"""
    from sqd import nnops
    torch.manual_seed(rows + K)
    lin = nn.Linear(K, N)
    x = torch.randn(rows, K)
    g = torch.randn(rows, N)
    ref_lin = nn.Linear(K, N).double()
    ref_lin.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = _act(ref_lin(xr), act)
    yr.backward(g.double())
    lin = lin.cuda()
    xd = x.cuda().requires_grad_(True)
    y = nnops.linear(xd, lin, act)
    y.backward(g.cuda())
    assert "hip" in nnops.BACKEND["linear"]

    def close(a, b, what):
        a, b = a.detach().cpu().double(), b.detach()
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-7, (what, float((a - b).abs().max()), float(b.abs().max()))
    close(y, yr, "y")
    close(xd.grad, xr.grad, "dx")
    close(lin.weight.grad, ref_lin.weight.grad, "dW")
    close(lin.bias.grad, ref_lin.bias.grad, "db")


@pytest.mark.parametrize("B,C,h,w,J", [(24, 256, 2, 5, 6), (2, 256, 1, 2, 6), (3, 64, 4, 3, 12), (1, 512, 3, 10, 16)])
def test_pose_head(B, C, h, w, J):
    """0.01 * pose_conv(x).mean(3).mean(2) (reference networks/pose_cnn.py:40-45) against the fp64 composite"""
    from sqd import nnops
    torch.manual_seed(B + C + J)
    conv = nn.Conv2d(C, J, 1)
    x, g = torch.randn(B, C, h, w), torch.randn(B, J)
    ref = nn.Conv2d(C, J, 1).double()
    ref.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = 0.01 * ref(xr).mean(3).mean(2)
    yr.backward(g.double())
    conv = conv.cuda()
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = nnops.pose_head(xd, conv, 0.01)
    y.backward(g.cuda())
    for a, b, what in ((y, yr, "out"), (xd.grad, xr.grad, "dx"), (conv.weight.grad, ref.weight.grad, "dW"), (conv.bias.grad, ref.bias.grad, "db")):
        a, b = a.detach().cpu().double(), b.detach()
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9, (what, float((a - b).abs().max()))


@pytest.mark.parametrize("N,C,H,W,k,stride,pad", [(2, 48, 17, 23, 3, 1, "same"), (2, 144, 16, 24, 3, 2, "same"), (2, 240, 15, 21, 5, 2, "same"),
                                                  (1, 768, 9, 12, 5, 1, "same"), (2, 24, 12, 10, 3, 1, 1), (3, 64, 11, 13, 5, 2, 2)])
def test_depthwise_conv(N, C, H, W, k, stride, pad):
    """depthwise k x k convolution (TensorFlow SAME or symmetric padding) against torch's grouped convolution in fp64"""
    from sqd import nnkernels
    torch.manual_seed(C + k)
    x, w = torch.randn(N, C, H, W), torch.randn(C, 1, k, k) / k
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    if pad == "same":
        (Ho, pt), (Wo, pl) = nnkernels.tf_same_pad(H, k, stride), nnkernels.tf_same_pad(W, k, stride)
        tot_h, tot_w = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
        yr = F.conv2d(F.pad(xr, (pl, tot_w - pl, pt, tot_h - pt)), wr, None, stride, 0, 1, C)
    else:
        yr = F.conv2d(xr, wr, None, stride, pad, 1, C)
    g = torch.randn(yr.shape)
    yr.backward(g.double())
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    y = nnkernels.DepthwiseConv.apply(xd, wd, stride, pad)
    assert y.shape == yr.shape
    y.backward(g.cuda())
    for a, b, what in ((y, yr, "y"), (xd.grad, xr.grad, "dx"), (wd.grad, wr.grad, "dw")):
        a, b = a.detach().cpu().double(), b.detach()
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7, (what, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("B,C,H,W", [(2, 144, 40, 64), (3, 40, 7, 9), (1, 1056, 33, 35)])
def test_batchnorm_takes_the_gate_pool_on_the_way(B, C, H, W):
    """batch_norm_act(pool=True) (sqd_bn_train_fwd_pool): the same output as the plain element-wise pass, and the per-image channel sums
    it attaches are bit-identical to sqd_se_pool of that output — the squeeze-and-excite node then skips its own pooling pass."""
    import ctypes
    from sqd import lib, nnkernels
    L = lib.lib()
    torch.manual_seed(C)
    x = torch.randn(B, C, H, W).cuda().contiguous(memory_format=torch.channels_last)
    bn_a, bn_b = nn.BatchNorm2d(C).cuda(), nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5)
        bn_a.bias.uniform_(-0.5, 0.5)
    bn_b.load_state_dict(bn_a.state_dict())
    y_plain = nnkernels.batch_norm_act(x, bn_a, "swish")
    y_pool = nnkernels.batch_norm_act(x, bn_b, "swish", pool=True)
    assert torch.equal(y_plain, y_pool) and torch.equal(bn_a.running_var, bn_b.running_var)
    part = y_pool._sqd_pool_part
    want = torch.empty_like(part)
    lib.check(L.sqd_se_pool(ctypes.c_void_p(y_plain.data_ptr()), None, ctypes.c_void_p(want.data_ptr()), B, H * W, C,
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "se_pool")
    assert part.shape == (B, L.sqd_se_chunks(H * W), C) and torch.equal(part, want)
    assert getattr(y_plain, "_sqd_pool_part", None) is None


@pytest.mark.parametrize("B,C,H,W,R", [(2, 144, 12, 20, 6), (3, 48, 7, 9, 12), (1, 1056, 5, 8, 44), (2, 3072, 3, 4, 128), (2, 240, 40, 64, 10)])
def test_squeeze_excite(B, C, H, W, R):
    from sqd import nnkernels
    torch.manual_seed(C + R)
    x = torch.randn(B, C, H, W)
    w1, b1 = torch.randn(R, C, 1, 1) / C ** 0.5, 0.1 * torch.randn(R)
    w2, b2 = torch.randn(C, R, 1, 1) / R ** 0.5, 0.1 * torch.randn(C)
    g = torch.randn(B, C, H, W)
    ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    s = ref[0].mean((2, 3), keepdim=True)
    r = F.conv2d(s, ref[1], ref[2])
    r = r * torch.sigmoid(r)
    yr = ref[0] * torch.sigmoid(F.conv2d(r, ref[3], ref[4]))
    yr.backward(g.double())
    devs = [x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)] + [t.cuda().requires_grad_(True) for t in (w1, b1, w2, b2)]
    y = nnkernels.SqueezeExcite.apply(*devs)
    y.backward(g.cuda())
    for a, b, what in [(y, yr, "y")] + [(d.grad, q.grad, n) for d, q, n in zip(devs, ref, ("dx", "dw1", "db1", "dw2", "db2"))]:
        a, b = a.detach().cpu().double(), b.detach()
        assert float((a - b).abs().max()) <= 5e-5 * float(b.abs().max()) + 1e-7, (what, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("N,C,H,W", [(2, 48, 12, 20), (3, 144, 9, 7), (2, 2048, 3, 4), (1, 24, 16, 24)])
def test_batchnorm_swish(N, C, H, W):
    """BatchNorm2d (train) + swish / SiLU, the activation of the EfficientNet trunk: forward, input and parameter gradients vs fp64"""
    from sqd import nnkernels
    torch.manual_seed(C)
    x = torch.randn(N, C, H, W) * 1.3 + 0.2
    g = torch.randn(N, C, H, W)
    bn_ref = nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).double()
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-0.3, 0.3)
    bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.01).cuda()
    bn.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in bn_ref.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = F.silu(bn_ref(xr))
    yr.backward(g.double())
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = nnkernels.batch_norm_act(xd, bn, "swish")
    y.backward(g.cuda())
    for a, b, what in ((y, yr, "y"), (xd.grad, xr.grad, "dx"), (bn.weight.grad, bn_ref.weight.grad, "dgamma"), (bn.bias.grad, bn_ref.bias.grad, "dbeta"),
                       (bn.running_var, bn_ref.running_var, "running_var")):
        a, b = a.detach().cpu().double(), b.detach()
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-6, (what, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("rows,cols", [(2048, 32), (1920, 64), (1024, 128), (300, 32), (7, 256)])
def test_colsum_multi_row_counts(rows, cols):
    """sqd_colsum_multi over few and over thousands of partial rows (2 / 8 / 32 float4 columns per workgroup by row count): fixed-order sums against
    float64, run-to-run identical"""
    from sqd import nnkernels
    g = torch.Generator().manual_seed(rows + cols)
    part = torch.randn(rows, cols, generator=g).cuda()
    out = [torch.empty(cols, device="cuda") for _ in range(2)]
    for o in out:
        nnkernels._colsum_multi([(part, o, 0)])
    ref = part.double().sum(0)
    assert torch.equal(out[0], out[1])
    assert float((out[0].double() - ref).abs().max()) <= 1e-5 * float(part.double().abs().sum(0).max())


def test_adam_step_leaves_the_filter_records_behind():
    """FusedAdam's launch leaves max |w| of every registered convolution filter in its operand-scale record (sqd_adam_step_amax): the next
    step's begin_step runs no pass over the weights; a filter written through torch in between is noticed (version) and the pass runs again"""
    import torch.nn as nn
    from sqd import nnkernels, nnops, optim
    nnkernels.amax_enable(True)
    torch.manual_seed(3)
    convs = [nn.Conv2d(32, 64, 3, 1, 1).cuda().to(memory_format=torch.channels_last), nn.Conv2d(64, 32, 1).cuda().to(memory_format=torch.channels_last)]
    params = [p for c in convs for p in c.parameters()]
    opt = optim.FusedAdam(params, lr=1e-2)
    x = torch.randn(2, 32, 12, 20, device="cuda").contiguous(memory_format=torch.channels_last)

    passes = []
    orig = nnkernels._wam_refresh
    nnkernels._wam_refresh = lambda: (passes.append(1), orig())[1]
    for c in convs:
        nnkernels.amax_of_weight(c.weight)            # (what a two-term fp16 plan of these layers does on its first use: the filters join the table)
    try:
        for step in range(3):
            refreshed = nnkernels.begin_step()
            y = nnops.conv2d(nnops.conv2d(x, convs[0]), convs[1])
            (y ** 2).mean().backward()
            standalone = nnkernels.AMAX_STATS["sites"].get("filter (first use)", 0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            assert not nnkernels.filter_records_stale(), step
            for c in convs:
                ent = nnkernels._wam_slot(c.weight)
                rec = nnkernels._WAM["buf"][ent[0] * nnkernels.AMAX_REC:(ent[0] + 1) * nnkernels.AMAX_REC]
                got = rec.view(torch.int32).max().view(torch.float32)
                assert float(got) == float(c.weight.detach().abs().max()), (step, float(got), float(c.weight.detach().abs().max()))
            if step > 0:
                assert refreshed is False, step                        # the previous step's Adam launch had left the records valid
        n = len(passes)
        assert nnkernels.begin_step() is False and len(passes) == n
        with torch.no_grad():
            convs[0].weight.mul_(3.0)                                  # written through torch: the version moves
        assert nnkernels.filter_records_stale()
        assert nnkernels.begin_step() is True and len(passes) == n + 1
        ent = nnkernels._wam_slot(convs[0].weight)
        rec = nnkernels._WAM["buf"][ent[0] * nnkernels.AMAX_REC:(ent[0] + 1) * nnkernels.AMAX_REC]
        assert float(rec.view(torch.int32).max().view(torch.float32)) == float(convs[0].weight.detach().abs().max())
    finally:
        nnkernels._wam_refresh = orig
