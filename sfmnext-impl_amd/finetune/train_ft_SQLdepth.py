"""The training step of the reference's metric-depth finetune loop (finetune/train_ft_SQLdepth.py:148-320) on the device:

    pred = model(img)["disp", 0]                               :231-232   SQLdepth (ConvNeXt-L U-Net + Self-Query head)
    pred = interpolate(pred, depth size, bilinear, align_corners=True)    :233   ops.ResizeAlignCorners
    for the first B // 2 samples: pred[i] *= median(depth[valid]) / median(pred[valid])     :234-264 (numpy on the host there)
                                                                                ops.median_ratio (exact medians on the device)
    loss = SILogLoss(pred, depth, mask = depth > min_depth)     :268-271, loss.py:24-42     ops.SILog
    loss.backward(); clip_grad_norm_(params, 0.1); AdamW.step(); OneCycleLR.step()          :279-282,285,199-205   FusedAdamW

Nothing moves to the host inside a step.  Data loading (`DepthDataLoader`: KITTI / NYU files) is outside this build; `synthetic_batch`
makes {"image", "depth"} batches of the right shapes."""
import torch
import torch.optim as optim

from SQLdepth import SQLdepth
from sqd import nnkernels, ops
from sqd.optim import FusedAdamW

from .loss import SILogLoss


class FinetuneArgs:
    """the flags of the reference's argparse that the step reads (train_ft_SQLdepth.py:324-430), with its defaults"""

    def __init__(self, **kw):
        self.bs, self.lr, self.wd, self.epochs = 16, 0.000357, 0.1, 25
        self.div_factor, self.final_div_factor, self.same_lr = 25, 100, False
        self.min_depth, self.max_depth = 1e-3, 80.0
        self.min_depth_eval, self.max_depth_eval = 1e-3, 80.0
        self.garg_crop, self.eigen_crop, self.dataset = True, False, "kitti"
        self.clip_grad_norm = 0.1
        for k, v in kw.items():
            setattr(self, k, v)


class FinetuneTrainer:
    def __init__(self, opt, args, steps_per_epoch, device=None):
        self.opt, self.args = opt, args
        self.device = device or torch.device("cuda")
        from sqd import nnops
        nnops.configure(opt, self.device)
        self.model = SQLdepth(opt).to(self.device).to(memory_format=torch.channels_last)
        self.model.train()
        if args.same_lr:
            params = self.model.parameters()
        else:                                                  # :180-182
            params = [{"params": list(self.model.get_1x_lr_params()), "lr": args.lr / 10},
                      {"params": list(self.model.get_10x_lr_params()), "lr": args.lr}]
        self.optimizer = FusedAdamW(params, lr=args.lr, weight_decay=args.wd, max_grad_norm=args.clip_grad_norm)
        self.scheduler = optim.lr_scheduler.OneCycleLR(self.optimizer, args.lr, epochs=args.epochs, steps_per_epoch=steps_per_epoch,
                                                       cycle_momentum=True, base_momentum=0.85, max_momentum=0.95,
                                                       div_factor=args.div_factor, final_div_factor=args.final_div_factor)
        self.criterion = SILogLoss()

    def train_step(self, batch):
        a = self.args
        nnkernels.begin_step()
        self.optimizer.zero_grad(set_to_none=True)
        img = batch["image"].to(self.device).contiguous(memory_format=torch.channels_last)
        depth = batch["depth"].to(self.device).contiguous()
        pred = self.model(img)
        pred = ops.ResizeAlignCorners.apply(pred.contiguous(), depth.shape[-2], depth.shape[-1])
        crop = "garg" if a.garg_crop else "eigen" if a.eigen_crop else None
        ratio = ops.median_ratio(pred, depth, pred.shape[0] // 2, a.min_depth_eval, a.max_depth_eval, crop)
        loss = self.criterion(pred, depth, a.min_depth, scale=ratio, interpolate=False)
        loss.backward()
        self.optimizer.step()                                  # gradient clipping is folded into the step (FusedAdamW)
        self.scheduler.step()
        return loss.detach(), ratio


@torch.no_grad()
def validate(trainer, batches):
    """The reference's validate() (train_ft_SQLdepth.py:323-378) on the device: per batch the model's prediction resized to the ground
    truth (align_corners), the dense SILog loss of every image, median-scaled and clamped error metrics over the Garg / Eigen crop — one
    [B,11] table per batch, everything transferred to the host once at the end.  -> (dict of the nine metric means, mean SILog loss).
    Images without a valid pixel are skipped, as `has_valid_depth` does in the reference's loader."""
    a, model = trainer.args, trainer.model
    was_training = model.training
    model.eval()
    tables, losses = [], []
    crop = "garg" if a.garg_crop else ("eigen" if a.dataset == "kitti" else "eigen_nyu") if a.eigen_crop else None
    try:
        for batch in batches:
            img = batch["image"].to(trainer.device)
            depth = batch["depth"].to(trainer.device).contiguous()
            frame = img if getattr(model.encoder, "planar_input", False) and img.is_contiguous() else img.contiguous(memory_format=torch.channels_last)
            pred = ops.ResizeAlignCorners.apply(model(frame).contiguous(), depth.shape[-2], depth.shape[-1])
            for i in range(pred.shape[0]):                      # the reference validates image by image (batch size 1): one loss per image
                losses.append(trainer.criterion(pred[i:i + 1], depth[i:i + 1], a.min_depth, interpolate=False))
            tables.append(ops.metric_depth_eval(pred, depth, a.min_depth_eval, a.max_depth_eval, crop))
    finally:
        model.train(was_training)
    table = torch.cat(tables).cpu().numpy()                     # the only device -> host transfers
    loss = torch.stack(losses).cpu().numpy()
    ok = table[:, 10] > 0
    means = {k: float(table[ok, j].mean()) for j, k in enumerate(ops.METRIC_DEPTH_NAMES)} if ok.any() else {}
    return means, float(loss[ok].mean()) if ok.any() else float("nan")


def synthetic_batch(bs, H, W, Hd=None, Wd=None, seed=0, density=0.3):
    """{"image": [bs,3,H,W] in [0,1], "depth": [bs,1,Hd,Wd] sparse metric depth (0 = no measurement)}"""
    g = torch.Generator().manual_seed(seed)
    Hd, Wd = Hd or H, Wd or W
    img = torch.rand(bs, 3, H, W, generator=g)
    depth = 2.0 + 70.0 * torch.rand(bs, 1, Hd, Wd, generator=g)
    depth[torch.rand(bs, 1, Hd, Wd, generator=g) > density] = 0.0
    return {"image": img, "depth": depth}
