"""TEST INFRASTRUCTURE — CPU restatement (torch fp32 + numpy, as the reference) of the metric-depth finetune step of
`/root/reference/finetune/train_ft_SQLdepth.py:219-285` and of `SILogLoss` (`finetune/loss.py:24-42`).  Only tests import this module.
Pinned: SILogLoss against golden G21, frozen from the imported reference class."""
import numpy as np
import torch
import torch.nn as nn


class SILogLoss(nn.Module):
    """finetune/loss.py:24-42"""

    def forward(self, input, target, mask=None, interpolate=True):
        if interpolate:
            input = nn.functional.interpolate(input, target.shape[-2:], mode="bilinear", align_corners=True)
        if mask is not None:
            input, target = input[mask], target[mask]
        g = torch.log(input) - torch.log(target)
        Dg = torch.var(g) + 0.15 * torch.pow(torch.mean(g), 2)
        return 10 * torch.sqrt(Dg)


def finetune_step(model, optimizer, scheduler, batch, args):
    """train_ft_SQLdepth.py:222-285 for one batch (model(img) returns the depth map, as SQLdepth.forward does) -> (loss, ratios)"""
    optimizer.zero_grad()
    img, depth = batch["image"], batch["depth"]
    pred = model(img)
    pred = nn.functional.interpolate(pred, depth.shape[-2:], mode="bilinear", align_corners=True)          # :233
    ratios = []
    for i in range(pred.shape[0] // 2):                                                                    # :234
        pred_np = pred[i].squeeze().detach().cpu().numpy()
        depth_np = depth[i].squeeze().detach().cpu().numpy()
        valid_mask = np.logical_and(depth_np > args.min_depth_eval, depth_np < args.max_depth_eval)
        gt_height, gt_width = depth_np.shape
        eval_mask = np.zeros(valid_mask.shape)
        if args.garg_crop:
            eval_mask[int(0.40810811 * gt_height):int(0.99189189 * gt_height), int(0.03594771 * gt_width):int(0.96405229 * gt_width)] = 1
        elif args.eigen_crop:
            eval_mask[int(0.3324324 * gt_height):int(0.91351351 * gt_height), int(0.0359477 * gt_width):int(0.96405229 * gt_width)] = 1
        else:
            eval_mask[:] = 1
        valid_mask = np.logical_and(valid_mask, eval_mask)
        pred_np, depth_np = pred_np[valid_mask], depth_np[valid_mask]
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                md, mp = np.median(depth_np), np.median(pred_np)
        ratio = 1 if (np.isnan(md) or np.isnan(mp)) else md / mp                                            # :260-263
        ratios.append(float(ratio))
        pred[i] *= ratio                                                                                    # :264
    mask = depth > args.min_depth                                                                           # :268
    loss = SILogLoss()(pred, depth, mask=mask.to(torch.bool), interpolate=False)                            # :271
    loss.backward()
    nn.utils.clip_grad_norm_(model.parameters(), args.clip_grad_norm)                                       # :281
    optimizer.step()
    scheduler.step()
    return loss.detach(), ratios


def compute_errors(gt, pred):
    """finetune/utils.py:76-96 (numpy, in the arrays' own precision: float32 in the reference's validation loop)"""
    import numpy as np
    thresh = np.maximum((gt / pred), (pred / gt))
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    err = np.log(pred) - np.log(gt)
    silog = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    log_10 = (np.abs(np.log10(gt) - np.log10(pred))).mean()
    return dict(a1=a1, a2=a2, a3=a3, abs_rel=abs_rel, rmse=rmse, log_10=log_10, rmse_log=rmse_log, silog=silog, sq_rel=sq_rel)


def validate_image(pred, gt_depth, min_depth_eval, max_depth_eval, garg_crop=True, eigen_crop=False, dataset="kitti"):
    """the per-image body of validate() (train_ft_SQLdepth.py:347-375): pred, gt_depth [H,W] float32 numpy arrays (the prediction
    already resized) -> (metrics dict, ratio, number of valid pixels) or None when no pixel is valid"""
    import numpy as np
    valid_mask = np.logical_and(gt_depth > min_depth_eval, gt_depth < max_depth_eval)
    if garg_crop or eigen_crop:
        gt_height, gt_width = gt_depth.shape
        eval_mask = np.zeros(valid_mask.shape)
        if garg_crop:
            eval_mask[int(0.40810811 * gt_height):int(0.99189189 * gt_height), int(0.03594771 * gt_width):int(0.96405229 * gt_width)] = 1
        elif dataset == "kitti":
            eval_mask[int(0.3324324 * gt_height):int(0.91351351 * gt_height), int(0.0359477 * gt_width):int(0.96405229 * gt_width)] = 1
        else:
            eval_mask[45:471, 41:601] = 1
        valid_mask = np.logical_and(valid_mask, eval_mask)
    if valid_mask.sum() == 0:
        return None
    pred = pred[valid_mask].copy()
    gt = gt_depth[valid_mask]
    ratio = np.median(gt) / np.median(pred)
    pred *= ratio
    pred[pred < min_depth_eval] = min_depth_eval
    pred[pred > max_depth_eval] = max_depth_eval
    pred[np.isinf(pred)] = max_depth_eval
    pred[np.isnan(pred)] = min_depth_eval
    return compute_errors(gt, pred), float(ratio), int(valid_mask.sum())


def eval_image(final, gt, min_depth, max_depth, garg_crop=True, eigen_crop=False, dataset="kitti"):
    """the per-image body of evaluate_metric_depth.py's eval() (:84-139): no median scaling, no clamps"""
    import numpy as np
    final = final.copy()
    final[np.isinf(final)] = max_depth
    final[np.isnan(final)] = min_depth
    valid_mask = np.logical_and(gt > min_depth, gt < max_depth)
    if garg_crop or eigen_crop:
        gt_height, gt_width = gt.shape
        eval_mask = np.zeros(valid_mask.shape)
        if garg_crop:
            eval_mask[int(0.40810811 * gt_height):int(0.99189189 * gt_height), int(0.03594771 * gt_width):int(0.96405229 * gt_width)] = 1
        elif dataset == "kitti":
            eval_mask[int(0.3324324 * gt_height):int(0.91351351 * gt_height), int(0.0359477 * gt_width):int(0.96405229 * gt_width)] = 1
        else:
            eval_mask[45:471, 41:601] = 1
        valid_mask = np.logical_and(valid_mask, eval_mask)
    if valid_mask.sum() == 0:
        return None
    return compute_errors(gt[valid_mask], final[valid_mask])
