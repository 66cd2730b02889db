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
