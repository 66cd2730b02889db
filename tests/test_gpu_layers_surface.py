"""GPU parity of the stand-alone layer classes (the reference's layers.py surface served by libsqd.so)
against the golden vectors frozen from the reference (G1, G2, G3, G5, G9)."""
import numpy as np
import pytest
import torch

from conftest import tt
from param_fill import chain_inputs

pytestmark = pytest.mark.gpu


def test_pose_algebra_g01(golden):
    import layers
    g = golden("g01_pose")
    aa, tr = tt(g["axisangle"]).cuda(), tt(g["translation"]).cuda()
    np.testing.assert_allclose(layers.rot_from_axisangle(aa).cpu().numpy(), g["R"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(layers.transformation_from_parameters(aa, tr, False).cpu().numpy(), g["M"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(layers.transformation_from_parameters(aa, tr, True).cpu().numpy(), g["M_inv"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(layers.get_translation_matrix(tr).cpu().numpy()[:, :3, 3], g["translation"][:, 0], rtol=0, atol=0)


def test_backproject_project3d_g02_g03(golden):
    import layers
    g2, g3 = golden("g02_backproject"), golden("g03_project3d")
    B, H, W = int(g2["B"]), int(g2["H"]), int(g2["W"])
    d = chain_inputs(int(g2["seed"]), B, H, W)
    cam = layers.BackprojectDepth(B, H, W)(tt(g2["depth"]).cuda(), tt(d["inv_K"]).cuda())
    assert np.array_equal(cam.cpu().numpy(), g2["cam_points"])                   # bit-exact (FMA-chain order)
    grid = layers.Project3D(B, H, W)(cam, tt(d["K"]).cuda(), tt(g3["T"]).cuda())
    np.testing.assert_allclose(grid.cpu().numpy(), g3["grid"], rtol=1e-5, atol=2e-6)


def test_ssim_g05_and_smooth_g09(golden):
    import layers
    g = golden("g05_ssim")
    s = layers.SSIM()(tt(g["x"]).cuda(), tt(g["y"]).cuda())
    np.testing.assert_allclose(s.cpu().numpy(), g["ssim"], rtol=1e-4, atol=2e-6)
    g = golden("g09_smooth")
    loss = layers.get_smooth_loss(tt(g["disp"]).cuda(), tt(g["img"]).cuda())
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-5)


def test_depth_errors_g14(golden):
    import layers
    from param_fill import sparse_gt
    g = golden("g14_depth_errors")
    rs = np.random.RandomState(1414)     # the generator drew (B,1,24,80) pred first, then the plain-error pred
    gt = sparse_gt(int(g["gt_seed"]), 2)
    pred_small = rs.uniform(2, 60, (2, 1, 24, 80)).astype(np.float32)
    pred = rs.uniform(1, 80, int((gt > 0).sum())).astype(np.float32)
    errs = layers.compute_depth_errors(tt(gt[gt > 0]).cuda(), tt(pred).cuda())
    np.testing.assert_allclose([float(e) for e in errs], g["plain_errors"], rtol=1e-4)


def test_standalone_layers_gradients_or_refusal():
    """The reference's stand-alone layers are autograd modules (layers.py:13-46,75-92,186-258,267-280).  So are these: get_smooth_loss,
    transformation_from_parameters, SSIM, BackprojectDepth, Project3D and Trainer.compute_reprojection_loss against the oracle's autograd
    (float64); arguments that are data in the reference's graph must REFUSE a tensor that requires a gradient — never detach it silently."""
    import layers
    from oracle import torch_ref as O
    torch.manual_seed(0)
    B, H, W = 2, 24, 40
    # get_smooth_loss w.r.t. the disparity
    disp = (torch.rand(B, 1, H, W) + 0.5)
    img = torch.rand(B, 3, H, W)
    d_dev = disp.cuda().requires_grad_(True)
    loss = layers.get_smooth_loss(d_dev, img.cuda())
    (3.0 * loss).backward()
    d_ref = disp.clone().requires_grad_(True)
    (3.0 * O.smooth_loss(d_ref, img)).backward()
    np.testing.assert_allclose(float(loss), float(O.smooth_loss(disp, img)), rtol=1e-5)
    np.testing.assert_allclose(d_dev.grad.cpu().numpy(), d_ref.grad.numpy(), rtol=1e-4, atol=1e-8)
    with pytest.raises(RuntimeError, match="no gradient is computed"):
        layers.get_smooth_loss(d_dev, img.cuda().requires_grad_(True))
    # transformation_from_parameters w.r.t. both pose vectors, both directions
    for invert in (False, True):
        aa, tr = 0.3 * torch.randn(B, 1, 3), torch.randn(B, 1, 3)
        w = torch.randn(B, 4, 4)
        a_dev, t_dev = aa.cuda().requires_grad_(True), tr.cuda().requires_grad_(True)
        (layers.transformation_from_parameters(a_dev, t_dev, invert) * w.cuda()).sum().backward()
        a_ref, t_ref = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        (O.transformation_from_parameters(a_ref, t_ref, invert) * w).sum().backward()
        np.testing.assert_allclose(a_dev.grad.cpu().numpy(), a_ref.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(t_dev.grad.cpu().numpy(), t_ref.grad.numpy(), rtol=1e-4, atol=1e-5)
    # SSIM w.r.t. both images (interior and reflected borders), and Trainer.compute_reprojection_loss through it
    x, y = torch.rand(B, 3, H, W), torch.rand(B, 3, H, W)
    x[:, :, :6, :8] *= 0.02                       # a dark flat corner: the ill-conditioned end of the quotient
    y[:, :, :6, :8] *= 0.02
    gup = torch.randn(B, 3, H, W)
    xd, yd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    (layers.SSIM()(xd, yd) * gup.cuda()).sum().backward()
    xr, yr = x.double().requires_grad_(True), y.double().requires_grad_(True)
    (O.ssim(xr, yr) * gup.double()).sum().backward()
    for got, want, name in ((xd.grad, xr.grad, "d ssim / d x"), (yd.grad, yr.grad, "d ssim / d y")):
        err = float((got.double().cpu() - want).abs().max() / want.abs().max())
        print("%s: max error %.2e of max |gradient| against float64" % (name, err))
        assert err <= 2e-4, (name, err)
    from trainer import Trainer
    xd2 = x.cuda().requires_grad_(True)
    Trainer.compute_reprojection_loss(None, xd2, y.cuda()).sum().backward()
    xr2 = x.double().requires_grad_(True)
    (0.85 * O.ssim(xr2, y.double()).mean(1, True) + 0.15 * (y.double() - xr2).abs().mean(1, True)).sum().backward()
    assert float((xd2.grad.double().cpu() - xr2.grad).abs().max() / xr2.grad.abs().max()) <= 2e-4
    # BackprojectDepth w.r.t. the depth, Project3D w.r.t. the points and T
    d = chain_inputs(41, B, H, W)
    K, invK = tt(d["K"]), tt(d["inv_K"])
    depth = torch.rand(B, 1, H, W) * 10 + 2
    Tm = O.transformation_from_parameters(0.05 * torch.randn(B, 1, 3), 0.3 * torch.randn(B, 1, 3), False)
    gg = torch.randn(B, H, W, 2)
    dd, Td = depth.cuda().requires_grad_(True), Tm.cuda().requires_grad_(True)
    pts = layers.BackprojectDepth(B, H, W)(dd, invK.cuda())
    (layers.Project3D(B, H, W)(pts, K.cuda(), Td) * gg.cuda()).sum().backward()
    dr, Tr = depth.double().requires_grad_(True), Tm.double().requires_grad_(True)
    (O.project_3d(O.backproject_depth(dr, invK.double()), K.double(), Tr, H, W) * gg.double()).sum().backward()
    for got, want, name in ((dd.grad, dr.grad, "d grid / d depth"), (Td.grad, Tr.grad, "d grid / d T")):
        err = float((got.double().cpu() - want).abs().max() / want.abs().max())
        print("%s: max error %.2e of max |gradient| against float64" % (name, err))
        assert err <= 2e-4, (name, err)
    # arguments that are data in the reference's graph, and the entry without an adjoint
    with pytest.raises(RuntimeError, match="no gradient is computed"):
        layers.BackprojectDepth(B, H, W)(depth.cuda(), invK.cuda().requires_grad_(True))
    with pytest.raises(RuntimeError, match="no gradient is computed"):
        layers.Project3D(B, H, W)(pts.detach(), K.cuda().requires_grad_(True), Tm.cuda())
    from sqd import ops as sqd_ops
    with pytest.raises(RuntimeError, match="no gradient is computed"):
        sqd_ops.grid_sample_border(x.cuda().requires_grad_(True), torch.zeros(B, H, W, 2).cuda())
