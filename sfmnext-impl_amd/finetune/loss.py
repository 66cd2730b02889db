"""SILogLoss of the reference (finetune/loss.py:24-42) as one fused kernel pair (sqd_silog_fwd / sqd_silog_bwd)."""
import torch
import torch.nn as nn

from sqd import ops


class SILogLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.name = "SILog"

    def forward(self, input, target, mask_min_depth, scale=None, interpolate=True):
        """input [B,1,h,w] prediction, target [B,1,H,W] depth; the mask of the reference's call sites is `target > min_depth`
        (train_ft_SQLdepth.py:268-271), given here as the threshold; scale [B]: constant per-sample factors on the prediction
        (the median ratios the reference multiplies in before the loss, :264)."""
        if interpolate and input.shape[-2:] != target.shape[-2:]:
            input = ops.ResizeAlignCorners.apply(input, target.shape[-2], target.shape[-1])
        if scale is None:
            scale = torch.ones(input.shape[0], device=input.device, dtype=torch.float32)
        return ops.SILog.apply(input, target, scale, float(mask_min_depth))
