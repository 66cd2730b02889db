"""Geometry + loss primitives of the MI355X build (surface of the reference's layers.py).

On the training path these are not called one by one: `Trainer.generate_images_pred` /
`compute_losses` run them fused in sqd.ops.PhotometricChain (depth upsample -> BackprojectDepth ->
Project3D -> grid_sample -> SSIM + L1 -> min/auto-mask -> smoothness).  The stand-alone names below
keep the reference's call signatures for scripts that use them directly; each is served by a kernel
of libsqd.so.  Device tensors only — there is no CPU fallback.  Like the reference's modules (layers.py:13-46,75-92,186-258,267-280)
they are differentiable: SSIM w.r.t. both images, BackprojectDepth w.r.t. the depth, Project3D w.r.t. the points and T,
get_smooth_loss w.r.t. the disparity, the pose-matrix functions w.r.t. both vectors — each through an adjoint kernel of
libsqd.so.  Arguments that are data in the reference's training graph (K, inv_K, the smoothness term's image) RAISE when
they require a gradient; nothing detaches silently."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from sqd import ops


def disp_to_depth(disp, min_depth, max_depth):
    """reference layers.py:51-60 (unused by the SQLdepth head, whose output already is depth)."""
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


def _pose(axisangle, translation, invert):
    return ops.pose_matrix(axisangle, translation, invert)


def transformation_from_parameters(axisangle, translation, invert=False):
    """(axisangle [B,1,3], translation [B,1,3]) -> 4x4 (reference layers.py:75-92).  One kernel launch
    instead of ~40 ATen ops, differentiable w.r.t. both vectors (sqd_pose_mats_bwd)."""
    return _pose(axisangle, translation, invert)


def rot_from_axisangle(vec):
    return _pose(vec, torch.zeros_like(vec), False)


def get_translation_matrix(translation_vector):
    return _pose(torch.zeros_like(translation_vector), translation_vector, False)


def compute_depth_errors(gt, pred):
    """Error metrics between predicted and ground-truth depths (reference layers.py:282-300);
    evaluated at log steps only."""
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


class SSIM(nn.Module):
    """SSIM loss map between two images, 7x7 window over a 3-px reflection pad (reference
    layers.py:13-46).  Stand-alone entry (sqd_ssim_fwd / sqd_ssim_bwd)."""

    def forward(self, x, y):
        return ops.ssim_map(x, y)


class BackprojectDepth(nn.Module):
    """depth image -> homogeneous point cloud [B,4,HW] (reference layers.py:186-215)."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        return ops.backproject(depth, inv_K)


class Project3D(nn.Module):
    """3D points -> normalised sampling grid [B,H,W,2] (reference layers.py:236-258)."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        return ops.project3d(points, K, T, self.height, self.width)


def get_smooth_loss(disp, img):
    """Edge-aware smoothness of `disp` (reference layers.py:267-280) — caller normalises disp."""
    return ops.smooth_loss_plain(disp, img)
