"""The reference's metric-depth evaluation (finetune/evaluate_metric_depth.py:47-141) on the device: flip test-time augmentation, resize to
the image, error metrics over the Garg / Eigen crop WITHOUT median scaling.  Writing the 16-bit prediction PNGs (`--save_dir`) and the
KITTI / NYU file loaders are host I/O outside this build; `evaluate` takes batches of {"image", "depth"} tensors."""
import torch

from sqd import ops


@torch.no_grad()
def predict_tta(model, image):
    """:47-62 — mean of the prediction and of the mirrored image's prediction mirrored back, resized (bilinear, align_corners) to the image"""
    planar = getattr(model.encoder, "planar_input", False)

    def run(x):
        return model(x.contiguous() if planar else x.contiguous(memory_format=torch.channels_last))
    pred = run(image)
    pred_lr = torch.flip(run(torch.flip(image, [3])), [3])
    final = 0.5 * (pred + pred_lr)
    return ops.ResizeAlignCorners.apply(final.contiguous(), image.shape[-2], image.shape[-1])


@torch.no_grad()
def evaluate(model, batches, args):
    """:65-141 -> (dict of the nine metric means rounded to 3 digits as the reference prints them, images without valid ground truth)"""
    was_training = model.training
    model.eval()
    device = next(model.parameters()).device
    crop = "garg" if args.garg_crop else ("eigen" if args.dataset == "kitti" else "eigen_nyu") if args.eigen_crop else None
    tables = []
    try:
        for batch in batches:
            image, gt = batch["image"].to(device), batch["depth"].to(device)
            final = predict_tta(model, image)
            if final.shape[-2:] != gt.shape[-2:]:
                raise ValueError("evaluate: prediction %s and ground truth %s differ in size" % (tuple(final.shape[-2:]), tuple(gt.shape[-2:])))
            tables.append(ops.metric_depth_eval(final, gt, args.min_depth, args.max_depth, crop, median_scaling=False))
    finally:
        model.train(was_training)
    table = torch.cat(tables).cpu().numpy()
    ok = table[:, 10] > 0
    metrics = {k: round(float(table[ok, j].mean()), 3) for j, k in enumerate(ops.METRIC_DEPTH_NAMES)} if ok.any() else {}
    return metrics, int((~ok).sum())
