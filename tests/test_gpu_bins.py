"""GPU parity of the fused adaptive-bins head (csrc/bins.hip) against the torch-fp32 composite the reference
runs (networks/depth_decoder_QTR.py:61-70): conv1x1 -> softmax(dim=1) -> sum(out * centers)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, Q, D, h, w
    (2, 64, 64, 12, 40),       # config B sizes (tile-aligned)
    (2, 64, 64, 7, 9),         # ragged pixel count (63: not a multiple of 4)
    (3, 120, 100, 6, 20),      # args_files: query_nums 120, dim_out 100 (padded rows / planes)
    (1, 128, 128, 8, 16),      # maximum sizes
    (2, 16, 8, 5, 13),         # tiny Q / D
    (12, 64, 64, 96, 320),     # full config-B head
    (2, 128, 128, 160, 512),   # configs[2]'s head (Q = dim_out = 128) at its map size, two images
]


def composite(energy, weight, bias, centers):
    out = torch.softmax(F.conv2d(energy, weight, bias), dim=1)
    return torch.sum(out * centers.view(centers.shape[0], -1, 1, 1), dim=1, keepdim=True)


@pytest.fixture(params=[1, 0], ids=["f16x2", "fp32mfma"])
def arith(request):
    """both arithmetics of csrc/bins.hip (sqd_bins_set_arith): two-term fp16 operands (the default) and the fp32 matrix instruction"""
    from sqd import lib
    lib.check(lib.lib().sqd_bins_set_arith(request.param), "bins_set_arith")
    yield request.param
    lib.check(lib.lib().sqd_bins_set_arith(1), "bins_set_arith")


@pytest.mark.parametrize("B,Q,D,h,w", CASES)
def test_bins_head_fwd_bwd(B, Q, D, h, w, arith):
    from sqd import ops
    g = torch.Generator().manual_seed(B * 1000 + Q + D + h)
    energy = 2.0 * torch.randn(B, Q, h, w, generator=g)
    weight = 0.3 * torch.randn(D, Q, 1, 1, generator=g)
    bias = 0.1 * torch.randn(D, generator=g)
    centers = torch.sort(torch.rand(B, D, generator=g) * 80.0, dim=1).values
    gout = torch.randn(B, 1, h, w, generator=g)
    big = B * h * w > 100000
    dt = torch.float32 if big else torch.float64          # fp64 oracle for the small cases, fp32 CPU composite for the full size
    ref_in = [t.detach().clone().to(dt).requires_grad_(True) for t in (energy, weight, bias, centers)]
    ref = composite(*ref_in)
    ref.backward(gout.to(dt))
    dev_in = [t.detach().clone().cuda().requires_grad_(True) for t in (energy, weight, bias, centers)]
    out = ops.BinsHead.apply(*dev_in)
    out.backward(gout.cuda())
    torch.cuda.synchronize()
    assert out.shape == (B, 1, h, w)
    # tolerance: 1e-4 relative (north_star), fp32 accumulation over Q terms + exp
    assert torch.allclose(out.cpu().double(), ref.detach().double(), rtol=1e-4, atol=1e-4), float((out.cpu().double() - ref.detach().double()).abs().max())
    for name, a, r in zip(("energy", "weight", "bias", "centers"), dev_in, ref_in):
        ga, gr = a.grad.cpu().double(), r.grad.double()
        scale = float(gr.abs().max()) + 1e-12
        err = float((ga - gr).abs().max()) / scale
        # weight / bias / centers gradients are sums over B*h*w pixels: fp32 summation-order noise grows with the count
        tol = 2e-4 if not big else 2e-3
        assert err < tol, (name, err)


def test_bins_head_rejects():
    from sqd import ops
    with pytest.raises(RuntimeError):
        ops.BinsHead.apply(torch.randn(1, 8, 4, 4), torch.randn(8, 8, 1, 1), torch.randn(8), torch.rand(1, 8))      # host tensors
    e = torch.randn(1, 130, 4, 4, device="cuda")
    with pytest.raises(RuntimeError):
        ops.BinsHead.apply(e, torch.randn(8, 130, 1, 1, device="cuda"), torch.randn(8, device="cuda"), torch.rand(1, 8, device="cuda"))


def test_bins_head_deterministic():
    from sqd import ops
    torch.manual_seed(3)
    args = [torch.randn(4, 64, 24, 80, device="cuda"), 0.3 * torch.randn(64, 64, 1, 1, device="cuda"), torch.randn(64, device="cuda"),
            torch.rand(4, 64, device="cuda") * 80]
    gout = torch.randn(4, 1, 24, 80, device="cuda")
    res = []
    for _ in range(2):
        a = [t.clone().requires_grad_(True) for t in args]
        ops.BinsHead.apply(*a).backward(gout)
        res.append([t.grad.clone() for t in a])
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,D", [(12, 64), (2, 100), (3, 128), (1, 1), (5, 24)])
def test_bin_centers_linear_norm(B, D):
    """relu + 0.1 -> normalise -> widths -> pad / cumsum -> mid-points (reference depth_decoder_QTR.py:56-66) as one kernel"""
    from sqd import ops
    g = torch.Generator().manual_seed(B * 7 + D)
    y = torch.randn(B, D, generator=g)
    gout = torch.randn(B, D, generator=g)
    vmin, vmax = 0.001, 80.0
    yr = y.double().requires_grad_(True)
    v = torch.relu(yr) + 0.1
    v = v / v.sum(dim=1, keepdim=True)
    widths = F.pad((vmax - vmin) * v, (1, 0), mode="constant", value=vmin)
    edges = torch.cumsum(widths, dim=1)
    ref = 0.5 * (edges[:, :-1] + edges[:, 1:])
    ref.backward(gout.double())
    yd = y.cuda().requires_grad_(True)
    out = ops.BinCenters.apply(yd, vmin, vmax)
    out.backward(gout.cuda())
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-5 * vmax
    gs = float(yr.grad.abs().max())
    assert float((yd.grad.cpu().double() - yr.grad).abs().max()) <= 1e-4 * gs + 1e-9


def test_bins_head_f16x2_wide_dynamic_range():
    """two-term fp16 operands with per-pixel scales: energies whose magnitude varies by 2^28 across the pixels (and by 2^12 inside a pixel's
    Q values): the logits must come out at fp32 accuracy for every pixel, not only for the loudest ones"""
    from sqd import lib, ops
    lib.check(lib.lib().sqd_bins_set_arith(1), "bins_set_arith")
    g = torch.Generator().manual_seed(77)
    B, Q, D, h, w = 2, 64, 64, 9, 32
    mag = torch.exp2(torch.randint(-24, 5, (B, 1, h, w), generator=g).float())
    energy = torch.randn(B, Q, h, w, generator=g) * mag * torch.exp2(torch.randint(-12, 1, (B, Q, h, w), generator=g).float())
    weight = 0.05 * torch.randn(D, Q, 1, 1, generator=g)
    bias = 0.1 * torch.randn(D, generator=g)
    centers = torch.sort(torch.rand(B, D, generator=g) * 80.0, dim=1).values
    ref = composite(energy.double(), weight.double(), bias.double(), centers.double())
    out = ops.BinsHead.apply(energy.cuda(), weight.cuda(), bias.cuda(), centers.cuda())
    lib.check(lib.lib().sqd_bins_set_arith(0), "bins_set_arith")
    out32 = ops.BinsHead.apply(energy.cuda(), weight.cuda(), bias.cuda(), centers.cuda())
    lib.check(lib.lib().sqd_bins_set_arith(1), "bins_set_arith")
    e16 = float((out.cpu().double() - ref).abs().max())
    e32 = float((out32.cpu().double() - ref).abs().max())
    print("max |pred - float64|: f16x2 %.3e, fp32 MFMA %.3e" % (e16, e32))
    assert e16 <= max(4 * e32, 1e-4), (e16, e32)


def test_bins_head_f16x2_gradients_wide_range_along_the_image():
    """gradients under a wide dynamic range ALONG the image: the upstream gradient grows by 2^24 and the energies by 2^6 from the first to the
    last pixel, so that every per-pixel operand scale of the dE product differs and the weight gradient sums terms 2^30 apart — all four
    gradients must match float64 at the fp32 instruction's accuracy (the test that held round 5's running-scale fp16 weight-gradient
    product, tools/experiments/bins_bwd_dw_f16_running_scale_r05.diff: parity-green, no faster, not kept)"""
    from sqd import lib, ops
    g = torch.Generator().manual_seed(91)
    B, Q, D, h, w = 2, 64, 64, 48, 64
    ramp = torch.exp2(torch.linspace(0, 1, h * w).view(1, 1, h, w) * 24.0)
    energy = torch.randn(B, Q, h, w, generator=g) * torch.exp2(torch.linspace(-3, 3, h * w).view(1, 1, h, w))
    weight = 0.05 * torch.randn(D, Q, 1, 1, generator=g)
    bias = 0.1 * torch.randn(D, generator=g)
    centers = torch.sort(torch.rand(B, D, generator=g) * 80.0, dim=1).values
    gout = torch.randn(B, 1, h, w, generator=g) * ramp * 2.0 ** -20
    ref_in = [t.detach().clone().double().requires_grad_(True) for t in (energy, weight, bias, centers)]
    composite(*ref_in).backward(gout.double())
    errs = {}
    for arith in (1, 0):
        lib.check(lib.lib().sqd_bins_set_arith(arith), "bins_set_arith")
        dev_in = [t.detach().clone().cuda().requires_grad_(True) for t in (energy, weight, bias, centers)]
        ops.BinsHead.apply(*dev_in).backward(gout.cuda())
        for name, a, r in zip(("energy", "weight", "bias", "centers"), dev_in, ref_in):
            errs[(arith, name)] = float((a.grad.cpu().double() - r.grad).abs().max()) / (float(r.grad.abs().max()) + 1e-30)
    lib.check(lib.lib().sqd_bins_set_arith(1), "bins_set_arith")
    print({k: "%.2e" % v for k, v in errs.items()})
    for name in ("energy", "weight", "bias", "centers"):
        assert errs[(1, name)] <= max(4 * errs[(0, name)], 2e-5), (name, errs[(1, name)], errs[(0, name)])

