"""Depth evaluation on the device — the flow of the reference's `evaluate_depth_config.py::evaluate` (:62-290) with every
per-pixel step on the MI355X: encoder + depth head on the test frames (and their flipped copies with --post_process), the
Monodepth-v1 flip post-processing (`ops.disp_post_process`), and per image the resize to the ground truth's size, Garg / Eigen
crop, median scaling and the seven error metrics (`ops.depth_eval`), all in double precision; only the final [n, 9] table comes
back to the host.  The reference moves every prediction to the host and runs numpy / OpenCV per image (:149-152, :225-261).

KITTI itself is not part of this build (datasets are host I/O): `--sqd_synthetic` evaluates on synthetic
KITTI-shaped frames with synthetic sparse ground truth (what the tests drive); `evaluate(opt, frames, gt_depths)` takes real
tensors from whoever loads them.  Flags are the reference's (options.py): --eval_split, --post_process, --eval_stereo,
--disable_median_scaling, --pred_depth_scale_factor, --load_weights_folder, --min_depth / --max_depth."""
import os

import numpy as np
import torch

import networks
from options import MonodepthOptions
from sqd import ops

STEREO_SCALE_FACTOR = 5.4           # reference evaluate_depth_config.py:27
MIN_DEPTH, MAX_DEPTH = 1e-3, 80.0   # :65-66


def build_models(opt, device):
    """encoder + depth head as the reference's evaluate() builds them (:88-118), weights from --load_weights_folder when given"""
    from trainer import Trainer
    shim = Trainer.__new__(Trainer)
    shim.opt = opt
    from sqd import nnops
    nnops.configure(opt, device)
    encoder, depth = Trainer._build_encoder(shim).to(device), Trainer._build_depth_head(shim).to(device)
    if opt.load_weights_folder:
        folder = os.path.expanduser(opt.load_weights_folder)
        for net, fname in ((encoder, "encoder.pth"), (depth, "depth.pth")):
            sd = torch.load(os.path.join(folder, fname), map_location=device)
            own = net.state_dict()
            net.load_state_dict({k: v for k, v in sd.items() if k in own})          # (:100-104: keys such as height / width are dropped)
    for net in (encoder, depth):
        net.to(memory_format=torch.channels_last)
        net.eval()
    return encoder, depth


@torch.no_grad()
def predict(opt, encoder, depth, frames):
    """frames [N,3,H,W] on the device -> predicted depth maps [N,h,w] fp64 (post-processed when --post_process)"""
    x = frames
    if opt.post_process:                    # two forward passes per image, batched as the reference does (:133-137)
        x = torch.cat((x, torch.flip(x, [3])), 0)
    out = depth(encoder(x.contiguous(memory_format=torch.channels_last)))[("disp", 0)][:, 0].contiguous()
    if opt.post_process:
        return ops.disp_post_process(out)
    return out.to(torch.float64)


@torch.no_grad()
def evaluate(opt, frames, gt_depths, encoder=None, depth=None, batch_size=None):
    """frames [N,3,H,W] device tensor, gt_depths: list of N [Hg,Wg] fp32 device tensors (sizes may differ) ->
    dict(errors [N,7], ratios [N] (NaN without median scaling), mean_errors [7], valid [N]) as numpy arrays"""
    device = frames.device
    if encoder is None:
        encoder, depth = build_models(opt, device)
    scale = float(opt.pred_depth_scale_factor)
    median = not opt.disable_median_scaling
    if opt.eval_stereo:                     # :213-217
        median, scale = False, STEREO_SCALE_FACTOR
    bs = batch_size or opt.batch_size
    rows = []
    for i0 in range(0, frames.shape[0], bs):
        pred = predict(opt, encoder, depth, frames[i0:i0 + bs])
        for j in range(pred.shape[0]):
            rows.append(ops.depth_eval(pred[j], gt_depths[i0 + j], eval_split=opt.eval_split, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH,
                                       pred_depth_scale_factor=scale, median_scaling=median))
    table = torch.stack(rows).cpu().numpy()                              # the only device -> host transfer
    res = {"errors": table[:, :7], "ratios": table[:, 7], "valid": table[:, 8], "mean_errors": table[:, :7].mean(0)}
    return res


def report(res, median_scaling):
    if median_scaling:
        med = np.median(res["ratios"])
        print(" Scaling ratios | med: {:0.3f} | std: {:0.3f}".format(med, np.std(res["ratios"] / med)))
    print("\n  " + ("{:>8} | " * 7).format(*ops.EVAL_METRIC_NAMES))
    print(("&{: 8.3f}  " * 7).format(*res["mean_errors"].tolist()) + "\\\\")
    print("\n-> Done!")


def main():
    opt = MonodepthOptions().parse()
    if not torch.cuda.is_available():
        raise RuntimeError("evaluate_depth.py needs the MI355X device (no CPU path)")
    if not opt.sqd_synthetic:
        raise NotImplementedError("real KITTI input is outside this build (SURVEY.md §2): pass --sqd_synthetic, or call "
                                  "evaluate(opt, frames, gt_depths) with tensors you loaded")
    from datasets.synthetic import synthetic_eval_set
    device = torch.device("cuda")
    frames, gts = synthetic_eval_set(16, opt.height, opt.width, device)
    res = evaluate(opt, frames, gts)
    report(res, not (opt.disable_median_scaling or opt.eval_stereo))


if __name__ == "__main__":
    main()
