"""GPU parity of the ConvNeXt-L / U-Net operators (csrc/convnext.hip, the 7x7 depthwise convolution of csrc/effnet.hip) and of the
assembled `networks.Unet` against torch composites in float64, the oracle (oracle/torch_ref.py) and the reference's own vectors (G20)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
# bars of the full-trunk step: set from the figures the test prints on MI355X (see below)
LOSS_RTOL, DISP_RTOL, WRONG_SIGN_FRACTION = 2e-4, 5e-4, 2e-2
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err, scale = float((a - b).abs().max()), float(b.abs().max())
    assert err <= tol * scale + 1e-7, (what, err, scale)


def cl(x):
    return x.cuda().contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("N,C,H,W,bias", [(2, 192, 12, 20, True), (1, 1536, 3, 5, False), (3, 48, 7, 9, True), (2, 4, 5, 3, False), (1, 384, 24, 40, True)])
def test_layer_norm_channels(N, C, H, W, bias):
    from sqd import nnops
    torch.manual_seed(C + H)
    x = torch.randn(N, C, H, W) * 2 + 0.5
    norm = nn.LayerNorm(C, eps=1e-6)
    norm.weight.data, norm.bias.data = torch.randn(C), torch.randn(C)
    pb = torch.randn(C) if bias else None
    wgt = torch.randn(N, C, H, W)
    xr, pr = x.double().requires_grad_(True), (pb.double().requires_grad_(True) if bias else None)
    nr = nn.LayerNorm(C, eps=1e-6).double()
    nr.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    yr = nr((xr + (pr.view(1, -1, 1, 1) if bias else 0)).permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    (yr * wgt.double()).sum().backward()
    ng = nn.LayerNorm(C, eps=1e-6).cuda()
    ng.load_state_dict(norm.state_dict())
    xg = cl(x).requires_grad_(True)
    pg = pb.cuda().requires_grad_(True) if bias else None
    y = nnops.layer_norm_channels(xg, ng, pg)
    (y * wgt.cuda()).sum().backward()
    close(y, yr, 2e-5, "y"); close(xg.grad, xr.grad, 1e-4, "dx")
    close(ng.weight.grad, nr.weight.grad, 1e-4, "dgamma"); close(ng.bias.grad, nr.bias.grad, 1e-4, "dbeta")
    if bias:
        close(pg.grad, pr.grad, 1e-4, "dprebias")


def test_gelu_and_scale_residual():
    from sqd import nnops
    torch.manual_seed(1)
    x = torch.randn(2, 64, 9, 11) * 3
    w = torch.randn_like(x)
    xr = x.double().requires_grad_(True)
    (F.gelu(xr) * w.double()).sum().backward()
    xg = cl(x).requires_grad_(True)
    y = nnops.gelu(xg)
    (y * w.cuda()).sum().backward()
    close(y, F.gelu(x.double()), 2e-6, "gelu"); close(xg.grad, xr.grad, 1e-5, "gelu'")
    res, z, gam = torch.randn(2, 64, 9, 11), torch.randn(2, 64, 9, 11), torch.randn(64)
    rr, zr, gr = (t.double().requires_grad_(True) for t in (res, z, gam))
    ((rr + zr * gr.view(1, -1, 1, 1)) * w.double()).sum().backward()
    rg, zg, gg = cl(res).requires_grad_(True), cl(z).requires_grad_(True), gam.cuda().requires_grad_(True)
    out = nnops.scale_residual(rg, zg, gg)
    (out * w.cuda()).sum().backward()
    close(out, res.double() + z.double() * gam.double().view(1, -1, 1, 1), 1e-6, "out")
    close(rg.grad, rr.grad, 1e-6, "dres"); close(zg.grad, zr.grad, 1e-6, "dz"); close(gg.grad, gr.grad, 1e-4, "dgamma")


@pytest.mark.parametrize("N,C,H,W", [(2, 16, 5, 7), (1, 128, 24, 80), (2, 8, 1, 1), (1, 4, 3, 2)])
def test_upsample2x(N, C, H, W):
    from sqd import nnops
    torch.manual_seed(H)
    x = torch.randn(N, C, H, W)
    w = torch.randn(N, C, 2 * H, 2 * W)
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2.0, mode="bilinear")
    (yr * w.double()).sum().backward()
    xg = cl(x).requires_grad_(True)
    y = nnops.upsample2x(xg)
    (y * w.cuda()).sum().backward()
    close(y, yr, 1e-6, "y"); close(xg.grad, xr.grad, 1e-5, "dx")


@pytest.mark.parametrize("N,C,H,W", [(2, 192, 12, 20), (1, 48, 9, 7), (2, 8, 24, 40)])
def test_depthwise_7x7(N, C, H, W):
    from sqd import nnops
    torch.manual_seed(C)
    conv = nn.Conv2d(C, C, 7, padding=3, groups=C)
    x, w = torch.randn(N, C, H, W), torch.randn(N, C, H, W)
    xr = x.double().requires_grad_(True)
    cr = nn.Conv2d(C, C, 7, padding=3, groups=C).double()
    cr.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    (F.conv2d(xr, cr.weight, None, 1, 3, 1, C) * w.double()).sum().backward()
    cg = nn.Conv2d(C, C, 7, padding=3, groups=C).cuda()
    cg.load_state_dict(conv.state_dict())
    xg = cl(x).requires_grad_(True)
    y = nnops.dw_conv(xg, cg)
    (y * w.cuda()).sum().backward()
    close(y, F.conv2d(x.double(), cr.weight, None, 1, 3, 1, C), 1e-5, "y"); close(xg.grad, xr.grad, 1e-5, "dx")
    close(cg.weight.grad, cr.weight.grad, 1e-4, "dw")


@pytest.mark.parametrize("C,K,s,H,W", [(3, 192, 4, 32, 48), (192, 384, 2, 12, 20), (16, 32, 2, 6, 10)])
def test_patchify_conv(C, K, s, H, W):
    from sqd import nnops
    torch.manual_seed(K)
    conv = nn.Conv2d(C, K, s, s)
    x = torch.randn(2, C, H, W)
    w = torch.randn(2, K, H // s, W // s)
    xr = x.double().requires_grad_(True)
    cr = nn.Conv2d(C, K, s, s).double()
    cr.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    (cr(xr) * w.double()).sum().backward()
    cg = nn.Conv2d(C, K, s, s).cuda()
    cg.load_state_dict(conv.state_dict())
    xg = x.cuda().requires_grad_(True)
    y = nnops.patchify_conv(xg, cg, s)
    (y * w.cuda()).sum().backward()
    close(y, cr(x.double()), 1e-4, "y"); close(xg.grad, xr.grad, 1e-4, "dx")
    close(cg.weight.grad, cr.weight.grad, 1e-4, "dw"); close(cg.bias.grad, cr.bias.grad, 1e-4, "db")


def test_unet_decoder_golden_g20(golden):
    """networks.UnetDecoder on the device against the reference's own output and gradients"""
    import networks
    from param_fill import fill_params
    g = golden("g20_unet_decoder")
    dec = networks.UnetDecoder([int(c) for c in g["enc_ch"]], tuple(int(c) for c in g["dec_ch"]), 4)
    assert sorted(dec.state_dict().keys()) == list(g["keys"])
    fill_params(dec, int(g["seeds"][0]))
    dec = dec.cuda().to(memory_format=torch.channels_last).train()
    rs = np.random.RandomState(int(g["seeds"][1]))
    sizes = [(3, 5), (7, 11), (14, 22), (28, 44)]
    feats = [torch.from_numpy((0.5 * rs.standard_normal((2, int(c), h, w))).astype(np.float32)) for c, (h, w) in zip(g["enc_ch"], sizes)]
    w = torch.from_numpy(np.random.RandomState(int(g["seeds"][2])).standard_normal(g["out"].shape).astype(np.float32))
    fr = [cl(f).requires_grad_(True) for f in feats]
    out = dec(fr)
    (out * w.cuda()).sum().backward()
    close(out, torch.from_numpy(g["out"]), 1e-4, "out")
    close(fr[0].grad, torch.from_numpy(g["grad_feat0"]), 2e-4, "grad head"); close(fr[3].grad, torch.from_numpy(g["grad_feat3"]), 2e-4, "grad skip")
    close(dec.final_conv.weight.grad, torch.from_numpy(g["grad_final_w"]), 2e-4, "dW final")
    close(dec.blocks[0].conv1.conv.weight.grad, torch.from_numpy(g["grad_b0c1"]), 5e-4, "dW block0")


def test_unet_matches_oracle():
    """a narrow ConvNeXt U-Net (same structure as convnext_large: 4 stages, LayerNorm2d stems / down-samplings, 7x7 depthwise blocks with
    layer scale, the reference's decoder) on the device against the oracle: output and a sample of parameter gradients"""
    import networks
    from oracle import torch_ref as O
    torch.manual_seed(0)
    kw = dict(depths=(1, 1, 2, 1), dims=(16, 32, 64, 128))
    ref = O.Unet(3, 16, (64, 32, 16, 8), **kw)
    for n, p in ref.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.5)                     # (the 1e-6 initial layer scale would hide the blocks from the comparison)
    ref.train()
    net = networks.Unet(num_classes=16, decoder_channels=(64, 32, 16, 8), **kw)
    assert sorted(net.state_dict().keys()) == sorted(ref.state_dict().keys())
    net.load_state_dict(ref.state_dict())
    net = net.cuda().to(memory_format=torch.channels_last).train()
    x = torch.rand(2, 3, 64, 96)
    w = torch.randn(2, 16, 32, 48)
    yr = ref(x)
    (yr * w).sum().backward()
    y = net(x.cuda())
    assert y.shape == (2, 16, 32, 48)
    (y * w.cuda()).sum().backward()
    close(y, yr, 2e-4, "out")
    grads = dict(net.named_parameters())
    for n, p in ref.named_parameters():
        if any(s in n for s in ("stem_0.weight", "stages_0.blocks.0.gamma", "stages_2.blocks.1.conv_dw.weight", "stages_1.downsample.1.weight",
                                "stages_3.blocks.0.mlp.fc1.weight", "stages_2.blocks.0.norm.weight", "decoder.blocks.3.conv1.conv.weight",
                                "decoder.final_conv.bias", "stages_0.blocks.0.conv_dw.bias", "stages_3.blocks.0.mlp.fc1.bias",
                                "stages_1.blocks.0.mlp.fc2.bias", "stages_2.blocks.1.mlp.fc2.bias")):
            close(grads[n].grad, p.grad, 2e-3, n)


@pytest.mark.parametrize("H,W,B,patch,Q,dout,tune", [(64, 128, 2, 8, 16, 32, False), (320, 1024, 1, 32, 64, 64, True)])
def test_convnext_large_train_step_matches_oracle(H, W, B, patch, Q, dout, tune):
    """one optimisation step of the Trainer with --backbone convnext_large (the full ConvNeXt-L U-Net, 238 M encoder parameters,
    + Depth_Decoder_QueryTr + PoseCNN; BASELINE.json configs[4]'s trunk) against the oracle: at a small image size under the default
    plans, and at the workload shape (320x1024, patch 32, 64 queries, dim_out 64 as bench.py runs it; batch 1 for the oracle's CPU step) with the plans
    measured in the step — the benchmarked arithmetic, the transposed-GEMM weight gradient of the stage 3 / 4 MLPs included"""
    from oracle import torch_ref as O
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    from sqd import nnkernels
    nnkernels.reset_plans()
    args = ["--backbone", "convnext_large", "--model_dim", "32", "--patch_size", str(patch), "--query_nums", str(Q), "--dim_out", str(dout),
            "--height", str(H), "--width", str(W), "--batch_size", str(B), "--num_workers", "0", "--sqd_synthetic",
            "--log_dir", "/tmp/sqd_convnext_test", "--max_depth", "80.0", "--sqd_no_graph"] + ([] if tune else ["--sqd_no_conv_tune"])
    torch.manual_seed(0)
    tr = Trainer(MonodepthOptions().parse(args))
    tr.set_train()
    for m in tr.models.values():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
    for n, p in tr.models["encoder"].named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.3)                     # (at the 1e-6 initial layer scale the blocks would not show in the loss)
    enc = O.Unet(3, 32, (1024, 512, 256, 128))
    dep = O.QueryTrDecoder(32, 32, patch, 4, Q, dout, min_val=0.001, max_val=80.0, dim_feedforward=1024, dropout=0.0)
    pose = O.PoseCNN(2)
    for ref, mine in ((enc, tr.models["encoder"]), (dep, tr.models["depth"]), (pose, tr.models["pose"])):
        ref.load_state_dict({k: v.detach().cpu() for k, v in mine.state_dict().items()})
        ref.train()
    cpu_inputs = synthetic_batch(B, H, W)
    noise = torch.randn(B, 2, H, W)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = O.RefTrainStep(enc, dep, pose, (0, -1, 1), H, W)
    ref_out, ref_losses = ref.step(dict(cpu_inputs), noise)
    inputs = {k: v.cuda() for k, v in cpu_inputs.items()}
    inputs[("noise", 0)] = noise.cuda()
    try:
        outputs, losses = tr.train_step(inputs)
        torch.cuda.synchronize()
        mix = nnkernels.plan_mix()
    finally:
        nnkernels.reset_plans()
    if tune:
        assert mix.get("wgrad", {}).get("transposed forward-gemm (its own fwd plan)", 0) >= 1, mix          # the stage 3 / 4 MLPs took plan 5
    got, want = float(losses["loss"]), float(ref_losses["loss"])
    d, dr = outputs[("disp", 0)].detach().cpu(), ref_out[("disp", 0)].detach()
    d_err = float((d - dr).abs().max()) / float(dr.abs().max())
    mine = tr.models["encoder"].state_dict()
    off = {}
    for k, v in enc.state_dict().items():
        if k.endswith(("stem_0.weight", "stages_2.blocks.5.mlp.fc2.weight", "stages_3.blocks.0.conv_dw.weight", "decoder.blocks.1.conv2.conv.weight")):
            off[k] = float(((mine[k].detach().cpu() - v).abs() > 1.2e-4).float().mean())
    print("convnext_large %dx%d: loss %.7f oracle %.7f (rel %.2e); disparity max err %.2e of max; fraction of weights updated the other way %s"
          % (H, W, got, want, abs(got - want) / abs(want), d_err, {k.split(".", 2)[-1]: "%.1e" % v for k, v in off.items()}))
    assert abs(got - want) <= LOSS_RTOL * abs(want), (got, want)
    assert d_err <= DISP_RTOL
    # updated weights (Adam step of 1e-4: a wrong-signed gradient moves a weight by 2e-4; only gradients within rounding of zero may)
    for k, v in off.items():
        assert v < WRONG_SIGN_FRACTION, (k, v)


def test_block_operators_record_the_maximum_of_what_they_write():
    """LayerNorm over channels and GELU tag their outputs, GELU's and the layer scale's backward their gradients, with max |tensor| — equal to
    torch's maximum bit for bit — so that the block's two Linear layers take their two-term fp16 operand scales without a pass of their own"""
    from sqd import nnkernels, nnops
    was = nnkernels.AMAX_ON
    nnkernels.amax_enable(True)
    try:
        torch.manual_seed(3)
        cl = torch.channels_last

        def tag_value(t):
            a = nnkernels._amax_get(t)
            assert a is not None, "no tag"
            return nnkernels.amax_value(a)
        nnkernels.begin_step()
        for N, C, H, W in ((2, 192, 20, 64), (1, 1536, 5, 8), (3, 96, 7, 9)):
            norm = nn.LayerNorm(C, eps=1e-6).cuda()
            with torch.no_grad():
                norm.weight.uniform_(0.5, 2.0); norm.bias.uniform_(-1.0, 1.0)
            x = (torch.randn(N, C, H, W, device="cuda") * 4).contiguous(memory_format=cl).requires_grad_(True)
            y = nnops.layer_norm_channels(x, norm, torch.randn(C, device="cuda"))
            assert tag_value(y) == float(y.abs().max())
            u = (torch.randn(N, C, H, W, device="cuda") * 3).contiguous(memory_format=cl).requires_grad_(True)
            v = nnops.gelu(u)
            assert tag_value(v) == float(v.abs().max())
            z = (torch.randn(N, C, H, W, device="cuda")).contiguous(memory_format=cl).requires_grad_(True)
            gamma = (torch.rand(C, device="cuda") * 1e-3).requires_grad_(True)
            out = nnops.scale_residual(x.detach(), z, gamma)
            seen = {}

            def grab(name):
                def hook(g):
                    seen[name] = (None if nnkernels._amax_get(g) is None else nnkernels.amax_value(nnkernels._amax_get(g)), float(g.abs().max()))
                    # ... and the per-block column sums of what they wrote (the bias gradient of the Linear layer in front of them)
                    cs = nnkernels._colsum_get(g)               # (valid for this content of the tensor only: it carries the version counter)
                    assert cs is not None and cs.shape == (C,), name
                    want = g.double().sum((0, 2, 3))
                    assert torch.allclose(cs.double(), want, rtol=1e-4, atol=1e-4 * float(want.abs().max())), name
                return hook
            u.register_hook(grab("gelu dx"))
            z.register_hook(grab("scale dz"))
            gv = torch.randn_like(v) * 1e-5
            v.backward(gv)
            out.backward(torch.randn_like(out) * 1e-5)
            assert seen["gelu dx"][0] == seen["gelu dx"][1] and seen["scale dz"][0] == seen["scale dz"][1], seen
            ur = u.detach().clone().requires_grad_(True)
            F.gelu(ur).backward(gv)
            assert torch.allclose(u.grad, ur.grad, rtol=1e-5, atol=1e-5 * float(ur.grad.abs().max()))
    finally:
        nnkernels.amax_enable(was)
