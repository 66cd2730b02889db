"""GPU parity tests of the hand-written gfx950 photometric chain (libsqd.so through its C ABI, via
sqd.ops) against the oracle (oracle/torch_ref.py, oracle/warp_chain.c) and the golden vectors frozen
from the imported reference.  Tolerances: integer taps bit-exact; fp32 tensors within 1e-4 rel
(BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from conftest import tt
from param_fill import chain_inputs

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def dev(a):
    return (tt(a) if isinstance(a, np.ndarray) else a).cuda().contiguous()


def pose_grad_close(got, want, tol=5e-3):
    """Pose gradients are sums over every pixel with heavy cancellation; measured fp32-vs-fp64 noise of
    the oracle itself is up to 1.6e-3 of max|grad| (B=12 192x640), so compare relative to the max."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert np.abs(got - want).max() < tol * np.abs(want).max(), (got, want)


def ties_only(sel_got, want, name, tie_tol=2.5e-5, avg=False):
    """identity_selection is an index output: it may differ from the oracle's only where the two best candidates of
    trainer.py:526's min tie to within fp32 rounding of the SSIM window sums.  SSIM = n/d is ill-conditioned in flat
    regions (d = (mu_x^2+mu_y^2+C1)(sigma_x+sigma_y+C2) ~ 1e-3 from variances that are differences of O(0.3) sums), so two
    exact-division fp32 implementations with different summation orders differ by ~1e-5 there: measured worst gap among
    the flipped pixels 1.25e-5 (29 of 1 474 560 pixels at config B, 0.002 %); bound = 2x that."""
    # (--avg_reprojection: the one reprojection candidate is the mean over the sources, trainer.py:508-509)
    comb = torch.cat((want["identity"], want["reproj"].mean(1, keepdim=True) if avg else want["reproj"]), 1).detach()
    top2 = torch.topk(comb, 2, dim=1, largest=False).values
    gap = (top2[:, 1] - top2[:, 0])
    # a flip of identity_selection needs the best identity and the best reprojection candidate to tie
    S = want["identity"].shape[1]
    gap_ir = (comb[:, :S].min(1).values - comb[:, S:].min(1).values).abs()
    mism = sel_got.detach().cpu() != want["identity_selection/0"]
    n = int(mism.sum())
    worst = float(gap_ir[mism].max()) if n else 0.0
    print("identity_selection[%s]: %d of %d pixels differ (%.4f %%), largest candidate gap among them %.2e"
          % (name, n, mism.numel(), 100.0 * n / mism.numel(), worst))
    assert worst < tie_tol, (n, worst)
    assert n <= 5e-4 * mism.numel(), n


def grad_close(got, want, frac=7.5e-3, tol=1e-3, mean_tol=1e-3):
    """Per-pixel gradient maps: the per-pixel min / auto-mask and sign() make a handful of pixels flip
    discretely when two candidates tie to within rounding.  Bounds = 2x the noise floor of the oracle
    itself: run in fp32 vs fp64 it differs by mean 2.5e-4..4.7e-4 of mean|grad| with 0.02..0.36 % of
    pixels beyond 1e-3*max (measured on the golden shapes and on B=12 192x640)."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    diff, scale = np.abs(got - want), np.abs(want).max()
    assert (diff > tol * scale).mean() < frac, ((diff > tol * scale).mean(), diff.max(), scale)
    assert diff.mean() < mean_tol * np.abs(want).mean() + 1e-12, (diff.mean(), np.abs(want).mean())


def close(a, b, rtol=RTOL, atol=1e-6):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def ops():
    from sqd import ops as o
    return o


@pytest.fixture(scope="module")
def O():
    from oracle import torch_ref
    return torch_ref


def oracle_P(O, d, depth, frame_ids=(-1, 1)):
    """P = (K @ T)[:, :3] for each source, computed by the oracle on the CPU."""
    mid = (1 / depth).mean(3, True).mean(2, True)
    Ps, Ts = [], []
    for i, f in enumerate(frame_ids):
        T = O.transformation_from_parameters(tt(d["axisangle_s%d" % i])[:, 0], tt(d["translation_s%d" % i])[:, 0] * mid[:, 0], f < 0)
        Ts.append(T)
        Ps.append(torch.matmul(tt(d["K"]), T)[:, :3, :])
    return torch.stack(Ps, 1).contiguous(), torch.stack(Ts, 1).contiguous(), mid


@pytest.mark.parametrize("B,H,W", [(2, 24, 80), (1, 48, 160), (3, 30, 70)])
def test_depth_up_bit_exact(ops, B, H, W):
    from oracle import c_chain
    d = chain_inputs(11, B, H, W)
    depth, part = ops.depth_up_fwd(dev(d["disp"]), H, W)
    want = c_chain.depth_up(d["disp"], H, W)
    assert np.array_equal(depth.cpu().numpy(), want)
    close(part[..., 0].sum(1), (1.0 / want.astype(np.float64)).sum((1, 2, 3)), rtol=1e-5)
    close(part[..., 1].sum(1), want.astype(np.float64).sum((1, 2, 3)), rtol=1e-5)


@pytest.mark.parametrize("B,H,W,rows", [(2, 24, 80, 8), (2, 48, 160, 15), (1, 64, 200, 22), (2, 40, 58, 8), (1, 23, 117, 15)])
def test_warp_taps_bit_exact(ops, O, B, H, W, rows):
    """Integer grid_sample taps + grid from the HIP kernel == C oracle, bit for bit, same P."""
    from oracle import c_chain
    d = chain_inputs(21, B, H, W)
    depth_np = c_chain.depth_up(d["disp"], H, W)
    P, _, _ = oracle_P(O, d, tt(depth_np))
    srcs = [dev(d["color_s0"]), dev(d["color_s1"])]
    ident = torch.full((B, 2, H, W), 10.0, device="cuda")
    out = ops.photo_fwd(dev(depth_np), dev(d["inv_K"]), dev(P), dev(d["color0"]), srcs, ident, training=False,
                        want_taps=True, rows_per_task=rows)
    for s in range(2):
        grid, x0, y0, warped = c_chain.warp(depth_np, d["inv_K"], P[:, s].numpy(), d["color_s%d" % s])
        got = out["x0y0"][s].cpu().numpy()
        assert np.array_equal(got[..., 0], x0) and np.array_equal(got[..., 1], y0)
        assert np.array_equal(out["sample"][s].cpu().numpy(), grid)
        close(out["warped"][s], warped, rtol=1e-6, atol=1e-7)


def _full_size_P(B, S, seed):
    """plausible KITTI-scale motions for the full-size tap tests: small rotations, translations up to ~0.5 m"""
    g = torch.Generator().manual_seed(seed)
    aa = 0.02 * torch.randn(B, S, 3, generator=g)
    tr = 0.3 * torch.randn(B, S, 3, generator=g)
    return aa, tr


@pytest.mark.parametrize("B,H,W,S", [(12, 192, 640, 2), (8, 320, 1024, 2), (4, 192, 640, 3)])
def test_warp_taps_bit_exact_full_size(ops, O, B, H, W, S):
    """BASELINE.json image sizes with the launch's DEFAULT tile cut (configs[1]: the balanced 1024-tile cut, 11 column strips x 24..28 rows
    on 8-wave workgroups; configs[2]: 18 strips, uniform 28-row tiles; S = 3: the stereo pair passes (0,1) + (2,2)): sampling grid and
    integer taps equal the C chain's bit for bit, for the kernel that dumps taps AND for the production call (the lean kernel of the
    default loss options, which stores no taps: its grid is compared bit for bit, and its warped colours equal the tap-dumping call's
    bit for bit — the same taps were read)."""
    from oracle import c_chain
    d = chain_inputs(23, B, H, W, S=S)
    depth_np = c_chain.depth_up(d["disp"], H, W)
    aa, tr = _full_size_P(B, S, 5)
    mid = (1.0 / tt(depth_np)).mean(3, True).mean(2, True)
    Ps = []
    for s in range(S):
        T = O.transformation_from_parameters(aa[:, s:s + 1], tr[:, s:s + 1] * mid[:, 0], s == 0)
        Ps.append(torch.matmul(tt(d["K"]), T)[:, :3, :])
    P = torch.stack(Ps, 1).contiguous()
    srcs = [dev(d["color_s%d" % s]) for s in range(S)]
    ident = ops.identity_fwd(dev(d["color0"]), srcs, dev(d["noise"]), 0)
    full = ops.photo_fwd(dev(depth_np), dev(d["inv_K"]), dev(P), dev(d["color0"]), srcs, ident, want_taps=True)
    prod = ops.photo_fwd(dev(depth_np), dev(d["inv_K"]), dev(P), dev(d["color0"]), srcs, ident)
    for s in range(S):
        grid, x0, y0, warped = c_chain.warp(depth_np, d["inv_K"], P[:, s].numpy(), d["color_s%d" % s])
        got = full["x0y0"][s].cpu().numpy()
        assert np.array_equal(got[..., 0], x0) and np.array_equal(got[..., 1], y0)
        assert np.array_equal(full["sample"][s].cpu().numpy(), grid)
        assert np.array_equal(prod["sample"][s].cpu().numpy(), grid)
        assert torch.equal(prod["warped"][s], full["warped"][s])
        close(full["warped"][s], warped, rtol=1e-6, atol=1e-7)
    assert torch.equal(prod["idx"], full["idx"]) and torch.equal(prod["sel"], full["sel"])


@pytest.mark.parametrize("B,H,W", [(12, 192, 640), (3, 96, 320), (2, 320, 1024)])
def test_forward_kernel_variants_agree(ops, B, H, W):
    """every kernel sqd_photo_set_fwd_variant can select (0 lean = the default, 6 lean with its late rows warped behind the barrier, 1 round 5's,
    2 colour-serial, 4 wide, 5 dynamic wave roles; 0x40: two rows of a wave in flight in phase 1, 0x80: resident workgroups) writes the same sampling grid, warped colours, identity_selection and argmin, bit for bit; the
    loss partials sum to the same loss (their partition differs: per wave / per row pair)."""
    from sqd import lib as _l
    d = chain_inputs(29, B, H, W)
    depth, part = ops.depth_up_fwd(dev(d["disp"]), H, W)
    aa, tr = _full_size_P(B, 2, 7)
    mid, T, P = ops.pose_mats_fwd(aa.cuda(), tr.cuda(), [1, 0], dev(d["K"]), part, H * W)
    srcs = [dev(d["color_s0"]), dev(d["color_s1"])]
    ident = ops.identity_fwd(dev(d["color0"]), srcs, dev(d["noise"]), 0)
    L = _l.lib()
    outs = {}
    try:
        for v in (1, 0, 6, 2, 4, 5, 0x40, 0x80):
            _l.check(L.sqd_photo_set_fwd_variant(v), "variant")
            outs[v] = ops.photo_fwd(depth, dev(d["inv_K"]), P, dev(d["color0"]), srcs, ident)
    finally:
        _l.check(L.sqd_photo_set_fwd_variant(0), "variant")
    ref = outs[1]
    for v, o in outs.items():
        assert torch.equal(o["sel"], ref["sel"]) and torch.equal(o["idx"], ref["idx"]), v
        for k in ("sample", "warped"):
            assert all(torch.equal(x, y) for x, y in zip(o[k], ref[k])), (v, k)
        a, b = o["loss_part"].double().sum().item(), ref["loss_part"].double().sum().item()
        assert abs(a - b) <= 1e-6 * abs(b), (v, a, b)


@pytest.mark.parametrize("B,H,W", [(12, 192, 640), (3, 96, 320), (2, 320, 1024), (1, 64, 64)])
def test_channels_last_sources_same_bits(ops, B, H, W):
    """source frames in [B,H,W,3] memory (SQD_SOURCES_HWC: ops recognise channels_last tensors by their strides; pack_pixels makes them):
    identity maps, every output of the fused forward and both gradients of the backward are those of the planar frames, bit for bit —
    the 2 x 2 taps are fetched as 16 + 8 bytes per row instead of three pairs, every product and sum is the same."""
    d = chain_inputs(53, B, H, W)
    depth, part = ops.depth_up_fwd(dev(d["disp"]), H, W)
    aa, tr = _full_size_P(B, 2, 11)
    mid, T, P = ops.pose_mats_fwd(aa.cuda(), tr.cuda(), [1, 0], dev(d["K"]), part, H * W)
    tgt, srcs = dev(d["color0"]), [dev(d["color_s0"]), dev(d["color_s1"])]
    px = ops.pack_pixels(srcs)
    for p, f in zip(px, srcs):
        assert p.shape == f.shape and p.is_contiguous(memory_format=torch.channels_last) and torch.equal(p, f)
    noise = dev(d["noise"])
    ident = ops.identity_fwd(tgt, srcs, noise, 0)
    assert torch.equal(ops.identity_fwd(tgt, px, noise, 0), ident)
    a = ops.photo_fwd(depth, dev(d["inv_K"]), P, tgt, srcs, ident)
    b = ops.photo_fwd(depth, dev(d["inv_K"]), P, tgt, px, ident)
    assert torch.equal(a["sel"], b["sel"]) and torch.equal(a["idx"], b["idx"]) and torch.equal(a["loss_part"], b["loss_part"])
    for k in ("sample", "warped"):
        assert all(torch.equal(x, y) for x, y in zip(a[k], b[k])), k
    ga = ops.photo_bwd(depth, dev(d["inv_K"]), P, tgt, srcs, a["sample"], a["warped"], a["idx"], 1.0 / (B * H * W))
    gb = ops.photo_bwd(depth, dev(d["inv_K"]), P, tgt, px, a["sample"], a["warped"], a["idx"], 1.0 / (B * H * W))
    assert torch.equal(ga[0], gb[0]) and torch.equal(ga[1], gb[1])


def test_channels_last_sources_refused_where_no_kernel_reads_them(ops):
    """launches only the planar kernels serve (loss options, tap dumps, three sources, narrow images, mixed layouts) raise instead of
    reading [B,H,W,3] memory as planes."""
    from sqd import lib as _l
    B, H, W = 2, 64, 96
    d = chain_inputs(59, B, H, W)
    depth, part = ops.depth_up_fwd(dev(d["disp"]), H, W)
    aa, tr = _full_size_P(B, 2, 13)
    mid, T, P = ops.pose_mats_fwd(aa.cuda(), tr.cuda(), [1, 0], dev(d["K"]), part, H * W)
    tgt, srcs = dev(d["color0"]), [dev(d["color_s0"]), dev(d["color_s1"])]
    px = ops.pack_pixels(srcs)
    ident = ops.identity_fwd(tgt, px, dev(d["noise"]), 0)
    with pytest.raises(RuntimeError, match="SQD_SOURCES_HWC"):
        ops.photo_fwd(depth, dev(d["inv_K"]), P, tgt, px, ident, want_taps=True)
    with pytest.raises(RuntimeError, match="SQD_SOURCES_HWC"):
        ops.photo_fwd(depth, dev(d["inv_K"]), P, tgt, px, ident, loss_flags=_l.LOSS_NO_SSIM)
    with pytest.raises(RuntimeError, match="SQD_SOURCES_HWC"):
        ops.identity_fwd(tgt, px + px[:1], None, 0)
    with pytest.raises(RuntimeError, match="share a memory layout"):
        ops.photo_fwd(depth, dev(d["inv_K"]), P, tgt, [px[0], srcs[1]], ident)
    small = [torch.rand(1, 3, 16, 32, device="cuda") for _ in range(3)]
    with pytest.raises(RuntimeError, match="SQD_SOURCES_HWC"):
        ops.identity_fwd(small[0], ops.pack_pixels(small[1:]), None, 0)


def test_pose_mats(ops, O):
    B, H, W = 3, 24, 80
    d = chain_inputs(31, B, H, W)
    depth, part = ops.depth_up_fwd(dev(d["disp"]), H, W)
    aa = torch.stack([tt(d["axisangle_s0"])[:, 0, 0], tt(d["axisangle_s1"])[:, 0, 0]], 1)
    tr = torch.stack([tt(d["translation_s0"])[:, 0, 0], tt(d["translation_s1"])[:, 0, 0]], 1)
    mid, T, P = ops.pose_mats_fwd(dev(aa), dev(tr), [1, 0], dev(d["K"]), part, H * W)
    Pw, Tw, midw = oracle_P(O, d, depth.cpu())
    close(mid, midw.flatten(), rtol=1e-5)
    close(T, Tw, rtol=1e-5, atol=1e-7)
    close(P, Pw, rtol=1e-5, atol=1e-5)
    # un-scaled cam_T_cam (trainer.py:336-337)
    _, T1, _ = ops.pose_mats_fwd(dev(aa), dev(tr), [1, 0], dev(d["K"]))
    want = torch.stack([O.transformation_from_parameters(aa[:, 0:1], tr[:, 0:1], True),
                        O.transformation_from_parameters(aa[:, 1:2], tr[:, 1:2], False)], 1)
    close(T1, want, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,H,W,rows", [(2, 24, 80, 8), (2, 48, 160, 15), (1, 33, 61, 8)])
def test_identity_losses(ops, O, B, H, W, rows):
    d = chain_inputs(41, B, H, W)
    tgt = tt(d["color0"])
    want = torch.cat([O.reprojection_loss(tt(d["color_s0"]), tgt), O.reprojection_loss(tt(d["color_s1"]), tgt)], 1)
    want = want + tt(d["noise"]) * 0.00001
    got = ops.identity_fwd(dev(tgt), [dev(d["color_s0"]), dev(d["color_s1"])], dev(d["noise"]), rows)
    close(got, want, rtol=RTOL, atol=2e-6)


def run_chain(ops, d, H, W, rows=0, grad=True):
    disp = dev(d["disp"]).requires_grad_(grad)
    aa = torch.stack([tt(d["axisangle_s0"])[:, 0, 0], tt(d["axisangle_s1"])[:, 0, 0]], 1).cuda().requires_grad_(grad)
    tr = torch.stack([tt(d["translation_s0"])[:, 0, 0], tt(d["translation_s1"])[:, 0, 0]], 1).cuda().requires_grad_(grad)
    tgt, srcs = dev(d["color0"]), [dev(d["color_s0"]), dev(d["color_s1"])]
    ident = ops.identity_fwd(tgt, srcs, dev(d["noise"]), rows)
    meta = dict(H=H, W=W, invert=[1, 0], smooth_weight=1e-3, rows_per_task=rows)
    outs = ops.PhotometricChain.apply(disp, aa, tr, dev(d["K"]), dev(d["inv_K"]), tgt, ident, meta, *srcs)
    return outs, disp, aa, tr


def oracle_chain(O, d, H, W):
    disp = tt(d["disp"]).requires_grad_(True)
    poses = {f: (tt(d["axisangle_s%d" % i]).requires_grad_(True), tt(d["translation_s%d" % i]).requires_grad_(True))
             for i, f in enumerate((-1, 1))}
    colors = {0: tt(d["color0"]), -1: tt(d["color_s0"]), 1: tt(d["color_s1"])}
    out = O.photometric_chain(disp, poses, tt(d["K"]), tt(d["inv_K"]), colors, [0, -1, 1], tt(d["noise"]), H, W)
    out["loss"].backward()
    return out, disp, poses


@pytest.mark.parametrize("B,H,W,rows,seed", [(2, 24, 80, 8, 51), (2, 48, 160, 15, 52), (1, 64, 122, 22, 53), (2, 31, 59, 8, 54)])
def test_full_chain_vs_oracle(ops, O, B, H, W, rows, seed):
    d = chain_inputs(seed, B, H, W)
    outs, disp, aa, tr = run_chain(ops, d, H, W, rows)
    total, photo, smooth, depth, sel, T = outs[:6]
    samples, warped = outs[6:8], outs[8:10]
    want, wdisp, wposes = oracle_chain(O, d, H, W)
    close(depth, want[("depth", 0, 0)], rtol=1e-6)
    for s, f in enumerate((-1, 1)):
        close(samples[s], want[("sample", f, 0)], rtol=RTOL, atol=2e-6)
        close(warped[s], want[("color", f, 0)], rtol=RTOL, atol=2e-5)
    ties_only(sel, want, "%dx%dx%d" % (B, H, W))
    close(smooth, want["smooth"], rtol=RTOL)
    close(total, want["loss"], rtol=RTOL)
    total.backward()
    grad_close(disp.grad, wdisp.grad)
    for s, f in enumerate((-1, 1)):
        pose_grad_close(aa.grad[:, s], wposes[f][0].grad[:, 0, 0])
        pose_grad_close(tr.grad[:, s], wposes[f][1].grad[:, 0, 0])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_golden_g07_g08(ops, golden, tag):
    """The HIP path against vectors frozen from the imported reference itself."""
    g7, g8 = golden("g07_generate_images_pred_" + tag), golden("g08_compute_losses_" + tag)
    B, H, W = int(g7["B"]), int(g7["H"]), int(g7["W"])
    d = chain_inputs(int(g7["seed"]), B, H, W)
    outs, disp, aa, tr = run_chain(ops, d, H, W)
    total, photo, smooth, depth, sel, T = outs[:6]
    samples, warped = outs[6:8], outs[8:10]
    close(depth, g7["depth"], rtol=1e-6)
    for s, n in enumerate(("m1", "p1")):
        close(samples[s], g7["sample_" + n], rtol=RTOL, atol=2e-6)
        close(warped[s], g7["color_" + n], rtol=RTOL, atol=2e-5)
    close(total, g8["loss"], rtol=RTOL)
    nm = int((sel.cpu().numpy() != g8["identity_selection"]).sum())
    print("identity_selection vs golden %s: %d of %d differ" % (tag, nm, sel.numel()))
    assert nm <= 5e-4 * sel.numel(), nm
    total.backward()
    grad_close(disp.grad, g8["grad_disp"])
    for s, n in enumerate(("m1", "p1")):
        wa, wt = g8["grad_axisangle_" + n][:, 0, 0], g8["grad_translation_" + n][:, 0, 0]
        pose_grad_close(aa.grad[:, s], wa)
        pose_grad_close(tr.grad[:, s], wt)


def test_smooth_fwd_bwd(ops, O):
    B, H, W = 2, 40, 72
    d = chain_inputs(61, B, H, W)
    depth, part = ops.depth_up_fwd(dev(d["disp"]), H, W)
    sm = ops.smooth_fwd(depth, dev(d["color0"]), part)
    got = sm[..., 0].sum() / (B * H * (W - 1)) + sm[..., 1].sum() / (B * (H - 1) * W)
    dd = depth.cpu().clone().requires_grad_(True)
    norm = dd / (dd.mean(2, True).mean(3, True) + 1e-7)
    want = O.smooth_loss(norm, tt(d["color0"]))
    close(got, want, rtol=RTOL)
    want.backward()
    planes = ops.smooth_bwd(depth, dev(d["color0"]), part, sm, 1.0)
    wd = dd.grad.numpy()
    assert np.abs(planes.cpu().numpy() - wd).max() < 1e-4 * np.abs(wd).max() + 1e-12


def test_full_size_config_b(ops, O):
    """BASELINE.json configs[1] shape (B=12, 192x640): whole chain vs the oracle on the host CPU, plus
    size-independent properties (partial sums == sum of per-pixel minima; backward linear in the
    upstream gradient)."""
    B, H, W = 12, 192, 640
    d = chain_inputs(71, B, H, W)
    outs, disp, aa, tr = run_chain(ops, d, H, W)
    total, photo = outs[0], outs[1]
    want, wdisp, wposes = oracle_chain(O, d, H, W)
    close(total, want["loss"], rtol=RTOL)
    close(photo, want["to_optimise"].mean(), rtol=RTOL)
    ties_only(outs[4], want, "config B 12x192x640")
    (2.5 * total).backward()
    grad_close(disp.grad / 2.5, wdisp.grad)
    for s, f in enumerate((-1, 1)):
        pose_grad_close(aa.grad[:, s] / 2.5, wposes[f][0].grad[:, 0, 0])
        pose_grad_close(tr.grad[:, s] / 2.5, wposes[f][1].grad[:, 0, 0])


def test_cpu_tensors_fail_loudly(ops):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.depth_up_fwd(torch.zeros(1, 1, 4, 4), 8, 8)


def _run_chain_stereo(ops, d, H, W, stereo_T, rows=0):
    disp = dev(d["disp"]).requires_grad_(True)
    aa = torch.stack([tt(d["axisangle_s0"])[:, 0, 0], tt(d["axisangle_s1"])[:, 0, 0]], 1).cuda().requires_grad_(True)
    tr = torch.stack([tt(d["translation_s0"])[:, 0, 0], tt(d["translation_s1"])[:, 0, 0]], 1).cuda().requires_grad_(True)
    tgt, srcs = dev(d["color0"]), [dev(d["color_s0"]), dev(d["color_s1"]), dev(d["color_s2"])]
    ident = ops.identity_fwd(tgt, srcs, dev(d["noise"]), rows)
    meta = dict(H=H, W=W, invert=[1, 0], smooth_weight=1e-3, rows_per_task=rows, use_stereo=True, stereo_T=dev(stereo_T))
    outs = ops.PhotometricChain.apply(disp, aa, tr, dev(d["K"]), dev(d["inv_K"]), tgt, ident, meta, *srcs)
    return outs, disp, aa, tr


def test_golden_g17_stereo(ops, golden):
    """three source frames (--use_stereo, reference trainer.py:52-53,405-421) against the reference's own vectors: the pair
    passes (0,1) + (2,2) of the tile kernel, the running minimum between them, the stereo source through stereo_T"""
    g = golden("g17_stereo_chain")
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    d = chain_inputs(int(g["seed"]), B, H, W, S=3)
    outs, disp, aa, tr = _run_chain_stereo(ops, d, H, W, g["stereo_T"])
    total, photo, smooth, depth, sel, T = outs[:6]
    samples, warped = outs[6:9], outs[9:12]
    close(depth, g["depth"], rtol=1e-6)
    for s, n in enumerate(("m1", "p1", "s")):
        close(samples[s], g["sample_" + n], rtol=RTOL, atol=2e-6)
        close(warped[s], g["color_" + n], rtol=RTOL, atol=2e-5)
    close(total, g["loss"], rtol=RTOL)
    nm = int((sel.cpu().numpy() != g["identity_selection"]).sum())
    print("identity_selection vs golden g17: %d of %d differ" % (nm, sel.numel()))
    assert nm <= 5e-4 * sel.numel(), nm
    total.backward()
    grad_close(disp.grad, g["grad_disp"])
    for s, n in enumerate(("m1", "p1")):
        pose_grad_close(aa.grad[:, s], g["grad_axisangle_" + n][:, 0, 0])
        pose_grad_close(tr.grad[:, s], g["grad_translation_" + n][:, 0, 0])


@pytest.mark.parametrize("B,H,W,rows,seed", [(2, 48, 160, 0, 81), (1, 40, 70, 8, 82)])
def test_stereo_chain_vs_oracle(ops, O, B, H, W, rows, seed):
    d = chain_inputs(seed, B, H, W, S=3)
    stereo_T = torch.eye(4).repeat(B, 1, 1)
    stereo_T[:, 0, 3] = -0.1
    outs, disp, aa, tr = _run_chain_stereo(ops, d, H, W, stereo_T, rows)
    wdisp = tt(d["disp"]).requires_grad_(True)
    poses = {f: (tt(d["axisangle_s%d" % i]).requires_grad_(True), tt(d["translation_s%d" % i]).requires_grad_(True))
             for i, f in enumerate((-1, 1))}
    colors = {0: tt(d["color0"]), -1: tt(d["color_s0"]), 1: tt(d["color_s1"]), "s": tt(d["color_s2"])}
    want = O.photometric_chain(wdisp, poses, tt(d["K"]), tt(d["inv_K"]), colors, [0, -1, 1, "s"], tt(d["noise"]), H, W,
                               stereo_T=stereo_T, use_stereo=True)
    want["loss"].backward()
    close(outs[0], want["loss"], rtol=RTOL)
    for s, f in enumerate((-1, 1, "s")):
        close(outs[9 + s], want[("color", f, 0)], rtol=RTOL, atol=2e-5)
    ties_only(outs[4], want, "stereo %dx%dx%d" % (B, H, W))
    outs[0].backward()
    grad_close(disp.grad, wdisp.grad)
    for s, f in enumerate((-1, 1)):
        pose_grad_close(aa.grad[:, s], poses[f][0].grad[:, 0, 0])
        pose_grad_close(tr.grad[:, s], poses[f][1].grad[:, 0, 0])


# ---------------------------------------------------------------------------------------------------
# the reference's loss options: --no_ssim / --avg_reprojection / --disable_automasking (trainer.py:447-451, 480-524)
LOSS_OPTION_SETS = {"no_ssim": dict(no_ssim=True), "avg": dict(avg_reprojection=True), "no_automask": dict(disable_automasking=True),
                    "avg_no_automask": dict(avg_reprojection=True, disable_automasking=True),
                    "no_ssim_avg": dict(no_ssim=True, avg_reprojection=True),
                    "all": dict(no_ssim=True, avg_reprojection=True, disable_automasking=True)}


def run_chain_options(ops, d, H, W, options, rows=0):
    from sqd import lib
    flags = lib.loss_flags(**options)
    disp = dev(d["disp"]).requires_grad_(True)
    aa = torch.stack([tt(d["axisangle_s0"])[:, 0, 0], tt(d["axisangle_s1"])[:, 0, 0]], 1).cuda().requires_grad_(True)
    tr = torch.stack([tt(d["translation_s0"])[:, 0, 0], tt(d["translation_s1"])[:, 0, 0]], 1).cuda().requires_grad_(True)
    tgt, srcs = dev(d["color0"]), [dev(d["color_s0"]), dev(d["color_s1"])]
    noise = dev(d["noise"][:, :1] if options.get("avg_reprojection") else d["noise"])
    ident = None if options.get("disable_automasking") else ops.identity_fwd(tgt, srcs, noise, rows, loss_flags=flags)
    meta = dict(H=H, W=W, invert=[1, 0], smooth_weight=1e-3, rows_per_task=rows, loss_flags=flags)
    outs = ops.PhotometricChain.apply(disp, aa, tr, dev(d["K"]), dev(d["inv_K"]), tgt, ident, meta, *srcs)
    return outs, disp, aa, tr, ident


def oracle_chain_options(O, d, H, W, options):
    disp = tt(d["disp"]).requires_grad_(True)
    poses = {f: (tt(d["axisangle_s%d" % i]).requires_grad_(True), tt(d["translation_s%d" % i]).requires_grad_(True))
             for i, f in enumerate((-1, 1))}
    colors = {0: tt(d["color0"]), -1: tt(d["color_s0"]), 1: tt(d["color_s1"])}
    noise = tt(d["noise"][:, :1] if options.get("avg_reprojection") else d["noise"])
    out = O.photometric_chain(disp, poses, tt(d["K"]), tt(d["inv_K"]), colors, [0, -1, 1], noise, H, W, **options)
    out["loss"].backward()
    return out, disp, poses


@pytest.mark.parametrize("name", sorted(LOSS_OPTION_SETS))
@pytest.mark.parametrize("B,H,W,rows,seed", [(2, 24, 80, 8, 61), (2, 48, 160, 0, 62), (1, 31, 59, 8, 63)])
def test_loss_options_vs_oracle(ops, O, name, B, H, W, rows, seed):
    """The fused HIP chain under each set of the reference's loss options against the oracle's compute_losses on the same inputs:
    identity maps, loss, identity_selection (ties only) and the gradients of disparity and poses."""
    options = LOSS_OPTION_SETS[name]
    d = chain_inputs(seed, B, H, W)
    outs, disp, aa, tr, ident = run_chain_options(ops, d, H, W, options, rows)
    total, photo, smooth, depth, sel, T = outs[:6]
    want, wdisp, wposes = oracle_chain_options(O, d, H, W, options)
    if ident is not None:
        close(ident, want["identity"], rtol=RTOL, atol=2e-6)
        ties_only(sel, want, "%s %dx%dx%d" % (name, B, H, W), avg=bool(options.get("avg_reprojection")))
    close(smooth, want["smooth"], rtol=RTOL)
    close(total, want["loss"], rtol=RTOL)
    total.backward()
    grad_close(disp.grad, wdisp.grad)
    for s, f in enumerate((-1, 1)):
        pose_grad_close(aa.grad[:, s], wposes[f][0].grad[:, 0, 0])
        pose_grad_close(tr.grad[:, s], wposes[f][1].grad[:, 0, 0])


@pytest.mark.parametrize("name", ["avg", "avg_no_automask", "no_ssim_avg"])
@pytest.mark.parametrize("B,H,W,rows,seed", [(2, 24, 80, 8, 71), (1, 48, 160, 0, 72)])
def test_avg_reprojection_three_sources_vs_oracle(ops, O, name, B, H, W, rows, seed):
    """--avg_reprojection with --use_stereo (three source frames: reference trainer.py:52-53 + 489-490, 508-509 take any S): the mean over
    three identity / reprojection maps travels through the pair passes (0,1) + (2,2); loss, selection and every gradient against the oracle."""
    from sqd import lib
    options = LOSS_OPTION_SETS[name]
    flags = lib.loss_flags(**options)
    d = chain_inputs(seed, B, H, W, S=3)
    stereo_T = torch.eye(4).repeat(B, 1, 1)
    stereo_T[:, 0, 3] = -0.1
    disp = dev(d["disp"]).requires_grad_(True)
    aa = torch.stack([tt(d["axisangle_s0"])[:, 0, 0], tt(d["axisangle_s1"])[:, 0, 0]], 1).cuda().requires_grad_(True)
    tr = torch.stack([tt(d["translation_s0"])[:, 0, 0], tt(d["translation_s1"])[:, 0, 0]], 1).cuda().requires_grad_(True)
    tgt, srcs = dev(d["color0"]), [dev(d["color_s0"]), dev(d["color_s1"]), dev(d["color_s2"])]
    noise1 = d["noise"][:, :1]
    ident = None if options.get("disable_automasking") else ops.identity_fwd(tgt, srcs, dev(noise1), rows, loss_flags=flags)
    meta = dict(H=H, W=W, invert=[1, 0], smooth_weight=1e-3, rows_per_task=rows, use_stereo=True, stereo_T=dev(stereo_T), loss_flags=flags)
    outs = ops.PhotometricChain.apply(disp, aa, tr, dev(d["K"]), dev(d["inv_K"]), tgt, ident, meta, *srcs)
    total, sel = outs[0], outs[4]
    wdisp = tt(d["disp"]).requires_grad_(True)
    poses = {f: (tt(d["axisangle_s%d" % i]).requires_grad_(True), tt(d["translation_s%d" % i]).requires_grad_(True)) for i, f in enumerate((-1, 1))}
    colors = {0: tt(d["color0"]), -1: tt(d["color_s0"]), 1: tt(d["color_s1"]), "s": tt(d["color_s2"])}
    want = O.photometric_chain(wdisp, poses, tt(d["K"]), tt(d["inv_K"]), colors, [0, -1, 1, "s"], tt(noise1), H, W, stereo_T=stereo_T, use_stereo=True,
                               **options)
    want["loss"].backward()
    if ident is not None:
        close(ident, want["identity"], rtol=RTOL, atol=2e-6)
        ties_only(sel, want, "%s S=3 %dx%dx%d" % (name, B, H, W), avg=True)
    close(total, want["loss"], rtol=RTOL)
    total.backward()
    grad_close(disp.grad, wdisp.grad)
    for s_, f in enumerate((-1, 1)):
        pose_grad_close(aa.grad[:, s_], poses[f][0].grad[:, 0, 0])
        pose_grad_close(tr.grad[:, s_], poses[f][1].grad[:, 0, 0])


@pytest.mark.parametrize("tag", ["no_ssim", "avg", "no_automask", "avg_no_automask", "no_ssim_avg"])
def test_golden_g23_loss_options(ops, golden, tag):
    """... and against vectors frozen from the imported reference's own compute_losses under those options."""
    g = golden("g23_loss_options_" + tag)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    fl = [bool(v) for v in g["flags"]]
    options = dict(no_ssim=fl[0], avg_reprojection=fl[1], disable_automasking=fl[2])
    d = chain_inputs(int(g["seed"]), B, H, W)
    outs, disp, aa, tr, _ = run_chain_options(ops, d, H, W, options)
    total, sel = outs[0], outs[4]
    close(total, g["loss"], rtol=RTOL)
    if not fl[2]:
        nm = int((sel.cpu().numpy() != g["identity_selection"]).sum())
        print("identity_selection vs golden %s: %d of %d differ" % (tag, nm, sel.numel()))
        assert nm <= 5e-4 * sel.numel(), nm
    total.backward()
    grad_close(disp.grad, g["grad_disp"])
    for s, n in enumerate(("m1", "p1")):
        pose_grad_close(aa.grad[:, s], g["grad_axisangle_" + n][:, 0, 0])
        pose_grad_close(tr.grad[:, s], g["grad_translation_" + n][:, 0, 0])


def test_loss_options_argument_errors(ops):
    from sqd import lib
    t = torch.zeros(1, 3, 16, 64, device="cuda")
    with pytest.raises(RuntimeError, match="at least two source frames"):
        ops.identity_fwd(t, [t], None, loss_flags=lib.LOSS_AVG_REPROJECTION)
    with pytest.raises(RuntimeError, match="unknown loss_flags"):
        ops.identity_fwd(t, [t, t], None, loss_flags=8)
