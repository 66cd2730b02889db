"""BatchNorm-backward statistics from the consuming convolution's data-gradient epilogue (sqd_conv_dgrad_bn + sqd_bn_train_bwd_pre):
the fused path must be TAKEN where it applies, must agree with the unfused path and with float64, and must step aside when the
gradient of the BatchNorm output is a sum over several consumers."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _chain(conv1, bn, conv2, x, act, res=None, double=False):
    y = conv1(x)
    z = bn(y) if res is None else bn(y) + res
    z = F.relu(z) if act == "relu" else F.leaky_relu(z, 0.01) if act == "leaky_relu" else z
    return conv2(z)


@pytest.mark.parametrize("N,C,H,W,Cm,K,R2,stride2,act,with_res,plan2", [
    (3, 32, 20, 28, 64, 48, 1, 1, "relu", False, (64, 64, 1, 16)),            # 1x1 consumer
    (2, 32, 21, 37, 64, 32, 3, 1, "relu", False, (128, 64, 1, 16)),           # 3x3 consumer, ragged tiles
    (2, 32, 24, 40, 64, 64, 3, 1, "leaky_relu", False, (64, 64, 1, 32 + 1024 + 2048)),      # the input-patch data gradient
    (2, 64, 24, 40, 64, 128, 3, 2, "relu", True, (64, 64, 1, 16)),            # stride-2 consumer (4 stride classes), residual into the BatchNorm
    (2, 64, 24, 40, 64, 128, 3, 2, "relu", False, None),                      # whatever the cost model picks
    (2, 32, 16, 24, 64, 32, 1, 1, None, False, (64, 64, 1, 32 + 1024)),                      # no activation, three-term plan
    (2, 64, 12, 20, 256, 64, 1, 1, "relu", False, (128, 64, 2, 16)),                        # split reduction: the sum over the splits takes the partials
    (3, 64, 12, 20, 320, 64, 3, 1, "leaky_relu", True, (64, 64, 4, 32 + 1024)),             # ... 4 splits, three-term plan, residual, 320 channels = two channel chunks
    (2, 32, 9, 13, 72, 64, 3, 2, "relu", False, (64, 64, 2, 16)),                           # ... stride classes, ragged rows, 72 channels (18 threads per row)
])
def test_bn_backward_statistics_from_the_dgrad_epilogue(N, C, H, W, Cm, K, R2, stride2, act, with_res, plan2):
    from sqd import lib, nnkernels, nnops
    L = lib.lib()
    torch.manual_seed(C + K + H)
    conv1, bn, conv2 = nn.Conv2d(C, Cm, 1, bias=False), nn.BatchNorm2d(Cm), nn.Conv2d(Cm, K, R2, stride2, R2 // 2, bias=False)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(N, C, H, W)
    res = torch.randn(N, Cm, H, W) if with_res else None
    # float64 reference
    import copy
    c1, b1, c2 = copy.deepcopy(conv1).double(), copy.deepcopy(bn).double(), copy.deepcopy(conv2).double()
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    out_r = _chain(c1, b1, c2, xr, act, rr)
    g = torch.randn_like(out_r)
    out_r.backward(g)
    geom2 = (N, H, W, Cm, K, R2, R2, stride2, R2 // 2, out_r.shape[2], out_r.shape[3])
    nnkernels.reset_plans()
    results = {}
    try:
        for fused in (True, False):
            nnkernels.FUSE_BN_BWD_STATS = fused
            nnkernels._PLAN_CACHE.clear()
            if plan2 is not None:
                assert L.sqd_conv_set_plan(1, *geom2, *plan2) == 0, L.sqd_last_error()
            m1, mb, m2 = copy.deepcopy(conv1).cuda().to(memory_format=torch.channels_last), copy.deepcopy(bn).cuda(), \
                copy.deepcopy(conv2).cuda().to(memory_format=torch.channels_last)
            xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            rg = res.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
            z = nnops.conv_bn_act(xg, m1, mb, act, residual=rg)
            assert (getattr(z, "_sqd_bn_src", None) is not None) == fused
            out = nnops.conv2d(z, m2)
            rows = nnkernels.conv_dgrad_stats_rows(geom2)
            out.backward(g.float().cuda())
            results[fused] = (out.detach().cpu().double(), xg.grad.cpu().double(), m1.weight.grad.cpu().double(), mb.weight.grad.cpu().double(),
                              mb.bias.grad.cpu().double(), rg.grad.cpu().double() if with_res else None, rows)
    finally:
        nnkernels.FUSE_BN_BWD_STATS = True
        L.sqd_conv_set_plan(1, *geom2, 0, 0, 0, 16)
        nnkernels.reset_plans()
    if plan2 is not None:
        assert results[True][6] > 0, results[True][6]                               # the fused path ran (split plans: through gemm_reduce_stats_kernel)
        if plan2[2] > 1:
            assert results[True][6] == (N * H * W + 63) // 64
    refs = (out_r.detach(), xr.grad, c1.weight.grad, b1.weight.grad, b1.bias.grad, rr.grad if with_res else None)
    for name, i in (("out", 0), ("dx", 1), ("dW1", 2), ("dgamma", 3), ("dbeta", 4), ("dres", 5)):
        if refs[i] is None:
            continue
        scale = float(refs[i].abs().max())
        ef, eu = float((results[True][i] - refs[i]).abs().max()) / scale, float((results[False][i] - refs[i]).abs().max()) / scale
        # gradients through BatchNorm are differences of large sums: both paths sit at the same fp32 level against float64
        assert ef <= 2e-4 and ef <= 4.0 * eu + 2e-6, (name, "fused", ef, "unfused", eu)


@pytest.mark.parametrize("N,C,H,W,Cm,shared_res,expect_ds", [
    (2, 32, 12, 20, 64, False, True),        # few partial rows: finalize + element-wise pass in one launch (bn_fused_bwd_kernel)
    (4, 32, 48, 80, 128, False, True),       # many partial rows: the three-launch path (bn_apply_bwd_kernel takes the sums)
    (8, 32, 96, 160, 256, False, True),      # ... more elements than the 2048 workgroups of that pass take in one sweep
    (2, 32, 12, 20, 72, False, False),       # 18 channel groups: not a multiple of 64 channels, not a divisor of 256 threads — not served
    (2, 32, 12, 20, 64, True, False),        # the down-sample output has a SECOND consumer: its gradient is a sum, the ordinary path must run
])
def test_residual_branch_batchnorm_sums_ride_on_the_main_branch(N, C, H, W, Cm, shared_res, expect_ds):
    """a bottleneck's tail, out = relu(bn3(conv3(x)) + bn_ds(conv_ds(x))) -> conv: the backward of bn3 writes dres, which is the whole gradient
    of bn_ds — its two sums are taken in that same pass (sqd_bn_train_bwd_res), bn_ds runs no reduction of its own, and every gradient is
    what float64 and the unfused path give"""
    from sqd import nnkernels, nnops
    import copy
    torch.manual_seed(N * H + Cm)
    conv3, bn3, convd, bnd, conv4 = nn.Conv2d(C, Cm, 1, bias=False), nn.BatchNorm2d(Cm), nn.Conv2d(C, Cm, 1, bias=False), nn.BatchNorm2d(Cm), \
        nn.Conv2d(Cm, 32, 1, bias=False)
    with torch.no_grad():
        for b in (bn3, bnd):
            b.weight.uniform_(0.5, 1.5)
            b.bias.uniform_(-0.3, 0.3)
    x = torch.randn(N, C, H, W)
    mods = [conv3, bn3, convd, bnd, conv4]
    r = [copy.deepcopy(m).double() for m in mods]
    xr = x.double().requires_grad_(True)
    ds_r = r[3](r[2](xr))
    z_r = F.relu(r[1](r[0](xr)) + ds_r)
    out_r = r[4](z_r) + (0.25 * ds_r.mean() if shared_res else 0.0)
    g = torch.randn_like(out_r)
    out_r.backward(g)
    results, took = {}, {}
    orig = nnkernels.BatchNormAct.backward
    try:
        for fused in (True, False):
            nnkernels.FUSE_BN_BWD_STATS = fused
            m = [copy.deepcopy(q).cuda() for q in mods]
            for q in (m[0], m[2], m[4]):
                q.to(memory_format=torch.channels_last)
            xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            pre = []

            def backward(ctx, dy, _pre=pre):
                sh = getattr(ctx, "shared", None)
                _pre.append(bool(sh is not None and sh.get("dx") is dy and sh.get("rows", 0) > 0))
                return orig(ctx, dy)
            nnkernels.BatchNormAct.backward = staticmethod(backward)
            ds = nnops.conv_bn_act(xg, m[2], m[3], None)
            z = nnops.conv_bn_act(xg, m[0], m[1], "relu", residual=ds)
            out = nnops.conv2d(z, m[4])
            if shared_res:
                out = out + 0.25 * ds.mean()
            out.backward(g.float().cuda())
            took[fused] = list(pre)          # backward order: bn3 first (sums from conv4's epilogue when fused), then bn_ds
            results[fused] = [xg.grad] + [q.weight.grad for q in m[:4]] + [m[1].bias.grad, m[3].bias.grad]
    finally:
        nnkernels.FUSE_BN_BWD_STATS = True
        nnkernels.BatchNormAct.backward = orig
    assert len(took[True]) == 2 and took[True][0], took                      # bn3: from the data-gradient epilogue of conv4
    assert took[True][1] == expect_ds, took                                  # bn_ds: from bn3's pass, unless its gradient is a sum of two / the shape is not served
    assert took[False] == [False, False], took
    refs = [xr.grad] + [q.weight.grad for q in r[:4]] + [r[1].bias.grad, r[3].bias.grad]
    for name, a, b, ref in zip(("dx", "dW3", "dgamma3", "dWd", "dgamma_ds", "dbeta3", "dbeta_ds"), results[True], results[False], refs):
        scale = float(ref.abs().max())
        ef, eu = float((a.cpu().double() - ref).abs().max()) / scale, float((b.cpu().double() - ref).abs().max()) / scale
        assert ef <= 2e-4 and ef <= 4.0 * eu + 2e-6, (name, "fused", ef, "unfused", eu)


@pytest.mark.parametrize("N,H,W,C,act,direct_use", [
    (2, 24, 40, 64, "relu", False),            # the ResNet stem: bn1 + ReLU -> max-pool, the decoder's skip through the pool node
    (3, 17, 23, 32, "leaky_relu", False),      # odd sizes, LeakyReLU
    (2, 24, 40, 64, "relu", True),             # a consumer that bypasses the pool node: the gradient is a sum, the ordinary path must run
])
def test_stem_batchnorm_sums_ride_on_the_maxpool_backward(N, H, W, C, act, direct_use):
    """x -> conv -> bn + act -> max-pool(3, 2, 1) -> conv, with the pooled tensor's input also feeding a skip convolution (handed through the pool
    node): the pool's backward gather writes bn's whole gradient and takes its two sums on the way (sqd_maxpool3x3s2_bwd_bn)"""
    from sqd import nnkernels, nnops
    import copy
    torch.manual_seed(H * W + C)
    conv1, bn, conv2, convs = nn.Conv2d(16, C, 1, bias=False), nn.BatchNorm2d(C), nn.Conv2d(C, 32, 1, bias=False), nn.Conv2d(C, 8, 1, bias=False)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(N, 16, H, W)
    mods = [conv1, bn, conv2, convs]
    r = [copy.deepcopy(m).double() for m in mods]
    xr = x.double().requires_grad_(True)
    zr = r[1](r[0](xr))
    zr = F.relu(zr) if act == "relu" else F.leaky_relu(zr, 0.01)
    out_r = r[2](F.max_pool2d(zr, 3, 2, 1)).sum() * 0.0 + (r[2](F.max_pool2d(zr, 3, 2, 1)) ** 2).sum() + (r[3](zr) ** 2).sum() + (zr.sum() if direct_use else 0.0)
    out_r.backward()
    results, took = {}, {}
    orig = nnkernels.BatchNormAct.backward
    try:
        for fused in (True, False):
            nnkernels.FUSE_BN_BWD_STATS = fused
            m = [copy.deepcopy(q).cuda() for q in mods]
            for q in (m[0], m[2], m[3]):
                q.to(memory_format=torch.channels_last)
            xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            pre = []

            def backward(ctx, dy, _pre=pre):
                sh = getattr(ctx, "shared", None)
                _pre.append(bool(sh is not None and sh.get("dx") is dy and sh.get("rows", 0) > 0))
                return orig(ctx, dy)
            nnkernels.BatchNormAct.backward = staticmethod(backward)
            z = nnops.conv_bn_act(xg, m[0], m[1], act)
            p, zs = nnops.maxpool3x3s2(z, skip=True)
            out = (nnops.conv2d(p, m[2]) ** 2).sum() + (nnops.conv2d(zs, m[3]) ** 2).sum()
            if direct_use:
                out = out + z.sum()
            out.backward()
            took[fused] = list(pre)
            results[fused] = [xg.grad, m[0].weight.grad, m[1].weight.grad, m[1].bias.grad]
    finally:
        nnkernels.FUSE_BN_BWD_STATS = True
        nnkernels.BatchNormAct.backward = orig
    assert took[True] == [not direct_use] and took[False] == [False], took
    refs = [xr.grad, r[0].weight.grad, r[1].weight.grad, r[1].bias.grad]
    for name, a, b, ref in zip(("dx", "dW1", "dgamma", "dbeta"), results[True], results[False], refs):
        scale = float(ref.abs().max())
        ef, eu = float((a.cpu().double() - ref).abs().max()) / scale, float((b.cpu().double() - ref).abs().max()) / scale
        assert ef <= 2e-4 and ef <= 4.0 * eu + 2e-6, (name, "fused", ef, "unfused", eu)


@pytest.mark.parametrize("N,Hi,Wi,Ho,Wo,C,Cs,act", [
    (2, 12, 20, 24, 40, 64, 32, "leaky_relu"),       # a decoder stage: BatchNorm + LeakyReLU -> up-sample x2 + concat with the skip
    (3, 6, 10, 13, 21, 32, 16, "relu"),              # odd target size
    (2, 12, 20, 24, 40, 72, 32, "leaky_relu"),       # 18 channel groups do not divide 256: not served, ordinary path
])
def test_decoder_batchnorm_sums_ride_on_the_upsample_concat_backward(N, Hi, Wi, Ho, Wo, C, Cs, act):
    """x -> conv -> bn + act -> bilinear up-sample + concat(skip) -> conv: the adjoint of the up-sampling writes bn's whole gradient and takes its
    two sums on the way (sqd_upcat_bwd_bn)"""
    from sqd import nnkernels, nnops
    import copy
    torch.manual_seed(Hi * Wo + C)
    conv1, bn, conv2 = nn.Conv2d(16, C, 1, bias=False), nn.BatchNorm2d(C), nn.Conv2d(C + Cs, 32, 1, bias=False)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x, skip = torch.randn(N, 16, Hi, Wi), torch.randn(N, Cs, Ho, Wo)
    mods = [conv1, bn, conv2]
    r = [copy.deepcopy(m).double() for m in mods]
    xr, sr = x.double().requires_grad_(True), skip.double().requires_grad_(True)
    zr = r[1](r[0](xr))
    zr = F.relu(zr) if act == "relu" else F.leaky_relu(zr, 0.01)
    out_r = (r[2](torch.cat([F.interpolate(zr, size=(Ho, Wo), mode="bilinear", align_corners=True), sr], 1)) ** 2).sum()
    out_r.backward()
    results, took = {}, {}
    orig = nnkernels.BatchNormAct.backward
    try:
        for fused in (True, False):
            nnkernels.FUSE_BN_BWD_STATS = fused
            m = [copy.deepcopy(q).cuda() for q in mods]
            for q in (m[0], m[2]):
                q.to(memory_format=torch.channels_last)
            xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            sg = skip.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            pre = []

            def backward(ctx, dy, _pre=pre):
                sh = getattr(ctx, "shared", None)
                _pre.append(bool(sh is not None and sh.get("dx") is dy and sh.get("rows", 0) > 0))
                return orig(ctx, dy)
            nnkernels.BatchNormAct.backward = staticmethod(backward)
            z = nnops.conv_bn_act(xg, m[0], m[1], act)
            out = (nnops.conv2d(nnops.upsample_concat(z, sg), m[2]) ** 2).sum()
            out.backward()
            took[fused] = list(pre)
            results[fused] = [xg.grad, sg.grad, m[0].weight.grad, m[1].weight.grad, m[1].bias.grad]
    finally:
        nnkernels.FUSE_BN_BWD_STATS = True
        nnkernels.BatchNormAct.backward = orig
    assert took[True] == [256 % (C // 4) == 0] and took[False] == [False], took
    refs = [xr.grad, sr.grad, r[0].weight.grad, r[1].weight.grad, r[1].bias.grad]
    for name, a, b, ref in zip(("dx", "dskip", "dW1", "dgamma", "dbeta"), results[True], results[False], refs):
        scale = float(ref.abs().max())
        ef, eu = float((a.cpu().double() - ref).abs().max()) / scale, float((b.cpu().double() - ref).abs().max()) / scale
        assert ef <= 2e-4 and ef <= 4.0 * eu + 2e-6, (name, "fused", ef, "unfused", eu)


@pytest.mark.parametrize("N,C,H,W,K,impl,splits,expect", [
    (2, 64, 24, 40, 128, 2 | (2 << 4), 7, True),      # fp32 shared-operand weight gradient: leaves no bias partials -> the BatchNorm backward delivers dbias
    (8, 64, 48, 80, 64, 3 | (3 << 4), 8, True),       # three-term shared-operand plan; 240 partial rows: the three-launch BatchNorm path
    (2, 64, 24, 40, 32, 1, 3, False),                 # fp32 direct plan: leaves bias partials, nothing to take over
])
def test_bias_gradient_from_the_batchnorm_backward(N, C, H, W, K, impl, splits, expect):
    """x -> conv(bias) -> bn + act -> conv: the column sums of the BatchNorm backward's dx are the first convolution's bias gradient; where its
    weight-gradient plan leaves no bias partials behind, the BatchNorm's element-wise pass takes them (sqd_bn_train_bwd_res's dx_colsum_part)
    instead of two colsum launches over dx"""
    from sqd import lib, nnkernels, nnops
    import copy
    L = lib.lib()
    torch.manual_seed(C + K)
    conv1, bn, conv2 = nn.Conv2d(C, K, 3, 1, 1, bias=True), nn.BatchNorm2d(K), nn.Conv2d(K, 16, 1, bias=False)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(N, C, H, W)
    r = [copy.deepcopy(m).double() for m in (conv1, bn, conv2)]
    xr = x.double().requires_grad_(True)
    out_r = (r[2](F.leaky_relu(r[1](r[0](xr)), 0.01)) ** 2).sum()
    out_r.backward()
    tagged = []
    orig_tag = nnkernels._colsum_tag
    nnkernels.reset_plans()
    try:
        nnkernels._PLAN_CACHE.clear()
        assert L.sqd_conv_wgrad_set_plan(N, H, W, C, K, 3, 3, impl, splits) == 0, L.sqd_last_error()

        def tag(t, cs):
            tagged.append(tuple(t.shape))
            return orig_tag(t, cs)
        nnkernels._colsum_tag = tag
        m = [copy.deepcopy(q).cuda() for q in (conv1, bn, conv2)]
        for q in (m[0], m[2]):
            q.to(memory_format=torch.channels_last)
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for it in range(2):          # (the hand-over is decided from the weight gradient's SETTLED plan: from the geometry's second backward on)
            del tagged[:]
            xg.grad = None
            for q in m:
                q.zero_grad(set_to_none=True)
            z = nnops.conv_bn_act(xg, m[0], m[1], "leaky_relu")
            out = (nnops.conv2d(z, m[2]) ** 2).sum()
            out.backward()
            if it == 0:
                assert (N, K, H, W) not in tagged, tagged
    finally:
        nnkernels._colsum_tag = orig_tag
        L.sqd_conv_wgrad_set_plan(N, H, W, C, K, 3, 3, -1, 0)
        nnkernels.reset_plans()
    assert ((N, K, H, W) in tagged) == expect, tagged
    # the bias gradient of a convolution in front of a BatchNorm is a sum that cancels analytically: compare on the scale of the terms that cancel
    # (sum |dx| per channel), as for every fp32 path
    for name, got, ref in (("dx", xg.grad, xr.grad), ("dW1", m[0].weight.grad, r[0].weight.grad), ("dgamma", m[1].weight.grad, r[1].weight.grad)):
        err = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        assert err <= 2e-4, (name, err)
    zr = r[0](xr.detach()).detach().requires_grad_(True)
    (r[2](F.leaky_relu(r[1](zr), 0.01)) ** 2).sum().backward()
    scale = float(zr.grad.abs().sum((0, 2, 3)).max())
    assert float((m[0].bias.grad.cpu().double() - r[0].bias.grad).abs().max()) <= 2e-6 * scale, (m[0].bias.grad, r[0].bias.grad, scale)


def test_summed_gradient_takes_the_ordinary_path():
    """the BatchNorm output feeds the convolution AND a second consumer directly (no skip hand-over): autograd sums the two gradients,
    the tensor the BatchNorm node receives is not the data gradient's, and the node must run its own reduction — results as float64"""
    from sqd import nnkernels, nnops
    import copy
    torch.manual_seed(4)
    N, C, H, W, Cm = 2, 32, 16, 24, 32
    conv1, bn, conv2 = nn.Conv2d(C, Cm, 1, bias=False), nn.BatchNorm2d(Cm), nn.Conv2d(Cm, Cm, 3, 1, 1, bias=False)
    x = torch.randn(N, C, H, W)
    c1, b1, c2 = copy.deepcopy(conv1).double(), copy.deepcopy(bn).double(), copy.deepcopy(conv2).double()
    xr = x.double().requires_grad_(True)
    zr = F.relu(b1(c1(xr)))
    out_r = c2(zr) + 0.5 * zr
    g = torch.randn_like(out_r)
    out_r.backward(g)
    m1, mb, m2 = conv1.cuda().to(memory_format=torch.channels_last), bn.cuda(), conv2.cuda().to(memory_format=torch.channels_last)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    z = nnops.conv_bn_act(xg, m1, mb, "relu")
    out = nnops.conv2d(z, m2) + 0.5 * z
    out.backward(g.float().cuda())
    assert z._sqd_bn_src["dx"] is None and z._sqd_bn_src["rows"] == 0           # consumed / reset by the BatchNorm node
    for got, ref, name in ((xg.grad, xr.grad, "dx"), (mb.weight.grad, b1.weight.grad, "dgamma"), (mb.bias.grad, b1.bias.grad, "dbeta")):
        err = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max())
        assert err <= 2e-4, (name, err)


def test_deferred_wgrad_reduce_rides_on_the_batchnorm_backward():
    """DEFER_WGRAD_REDUCE: the sum of a weight gradient's pixel splits is carried by the finalize launch of the next BatchNorm backward
    (sqd_conv_wgrad_partials + sqd_bn_train_bwd_pre_red) — same bits as the launch of its own, for every filter of a conv/BN stack incl.
    the last convolution (no BatchNorm backward follows it: the end-of-pass callback sums it)."""
    from sqd import nnkernels, nnops
    torch.manual_seed(11)
    convs = [nn.Conv2d(16, 64, 3, 1, 1, bias=False), nn.Conv2d(64, 32, 1, 1, 0, bias=False), nn.Conv2d(32, 64, 3, 2, 1, bias=True)]
    bns = [nn.BatchNorm2d(64), nn.BatchNorm2d(32), nn.BatchNorm2d(64)]
    convs = [c.cuda().to(memory_format=torch.channels_last) for c in convs]
    bns = [b.cuda() for b in bns]
    x = torch.randn(3, 16, 40, 72, device="cuda").contiguous(memory_format=torch.channels_last)
    gy = torch.randn(3, 64, 20, 36, device="cuda").contiguous(memory_format=torch.channels_last)

    deferred = []
    real_set = nnkernels._set_pending_reduce

    def counting_set(part, dw, n, splits, w=None):
        deferred.append(dw)
        return real_set(part, dw, n, splits, w)

    def run(defer):
        nnkernels.begin_step()                  # per-step use counts of the filters: a filter is deferred only when used once
        nnkernels.DEFER_WGRAD_REDUCE = defer
        nnkernels._set_pending_reduce = counting_set
        del deferred[:]
        try:
            for m in convs + bns:
                m.zero_grad(set_to_none=True)
            for b in bns:
                b.reset_running_stats()
            h = x
            for c, b in zip(convs, bns):
                h = nnops.conv_bn_act(h, c, b, "relu")
            h.backward(gy)
            assert not nnkernels._PENDING_REDUCE
            if defer:
                # the deferred path ran: the second and third convolutions read a training-mode BatchNorm's output (the first reads the
                # plain input: no BatchNorm backward follows its weight gradient, it takes the launch of its own), and their filters
                # hold the very tensors handed over by the node
                assert len(deferred) == 2, len(deferred)
                assert all(any(c.weight.grad.data_ptr() == d.data_ptr() for d in deferred) for c in convs[1:])
            else:
                assert not deferred
            return [c.weight.grad.clone() for c in convs] + [convs[2].bias.grad.clone()] + [b.weight.grad.clone() for b in bns]
        finally:
            nnkernels.DEFER_WGRAD_REDUCE = False
            nnkernels._set_pending_reduce = real_set
    ref, got = run(False), run(True)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,C,H,W,Cm,act,with_res", [(12, 128, 6, 20, 512, "relu", True), (12, 64, 12, 40, 256, "relu", False),
                                                     (8, 64, 20, 64, 128, "leaky_relu", False), (2, 32, 9, 13, 64, None, False)])
def test_finalize_and_elementwise_pass_in_one_launch(N, C, H, W, Cm, act, with_res):
    """Round 4: with few partial rows from the producing convolution (<= 160: the 12x40 / 6x20 maps of layer3 / layer4, 20x64 at configs[2]) and a
    channel count that is a multiple of 64, BatchNorm's finalize and element-wise pass are ONE launch, forward and backward — every workgroup sums
    the partial rows of its own 64 channels.  Output, batch statistics, running statistics and every gradient against float64."""
    import copy
    from sqd import lib, nnkernels, nnops
    torch.manual_seed(N + Cm)
    conv1, bn, conv2 = nn.Conv2d(C, Cm, 1, bias=False), nn.BatchNorm2d(Cm), nn.Conv2d(Cm, 64, 1, bias=False)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x, res = torch.randn(N, C, H, W), (torch.randn(N, Cm, H, W) if with_res else None)
    c1, b1, c2 = copy.deepcopy(conv1).double(), copy.deepcopy(bn).double(), copy.deepcopy(conv2).double()
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    out_r = _chain(c1, b1, c2, xr, act, rr)
    g = torch.randn_like(out_r)
    out_r.backward(g)
    nnkernels.reset_plans()
    m1, mb, m2 = conv1.cuda().to(memory_format=torch.channels_last), bn.cuda(), conv2.cuda().to(memory_format=torch.channels_last)
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = res.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if with_res else None
    geom1 = (N, H, W, C, Cm, 1, 1, 1, 0, H, W)
    rows = nnkernels.conv_stats_rows(geom1)
    assert 0 < rows <= 160 and Cm % 64 == 0, rows                 # the shape takes the one-launch path (csrc/bn_act.hip FUSE_MAX_ROWS)
    z = nnops.conv_bn_act(xg, m1, mb, act, residual=rg)
    out = nnops.conv2d(z, m2)
    out.backward(g.float().cuda())

    def close(a, b, what, tol=2e-4):
        a, b = a.detach().cpu().double(), b.detach()
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-7, (what, float((a - b).abs().max()), float(b.abs().max()))
    close(out, out_r, "out")
    close(mb.running_mean, b1.running_mean, "running_mean", 1e-5)
    close(mb.running_var, b1.running_var, "running_var", 1e-5)
    close(xg.grad, xr.grad, "dx")
    close(mb.weight.grad, b1.weight.grad, "dgamma")
    close(mb.bias.grad, b1.bias.grad, "dbeta")
    close(m1.weight.grad, c1.weight.grad, "dW1")
    if with_res:
        close(rg.grad, rr.grad, "dres")
