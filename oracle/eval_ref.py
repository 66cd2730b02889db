"""TEST INFRASTRUCTURE — CPU restatement (numpy, float64) of the reference's depth-evaluation arithmetic
(`/root/reference/evaluate_depth_config.py`).  Only tests import this module; the product path (`evaluate_depth.py`,
`sqd/ops.py`) never does.

Pinned: `compute_errors` and `batch_post_process_disparity` against golden group G19, frozen from the imported reference
(`tests/golden/make_goldens.py::g19_eval`).  Parity unpinned: `resize_linear` restates OpenCV's `cv2.resize(..., INTER_LINEAR)`
(half-pixel centres, no anti-aliasing, float coefficients, double accumulation for 64-bit input) from its published definition —
OpenCV is not available in this container (the reference calls it at evaluate_depth_config.py:230)."""
import numpy as np

MIN_DEPTH, MAX_DEPTH = 1e-3, 80.0                       # evaluate_depth_config.py:74-75
STEREO_SCALE_FACTOR = 5.4                               # evaluate_depth_config.py:27


def compute_errors(gt, pred):
    """evaluate_depth_config.py:30-47"""
    thresh = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def batch_post_process_disparity(l_disp, r_disp):
    """evaluate_depth_config.py:50-59 (Monodepth v1 flip post-processing); r_disp is the flipped image's output flipped back"""
    _, h, w = l_disp.shape
    m_disp = 0.5 * (l_disp + r_disp)
    l, _ = np.meshgrid(np.linspace(0, 1, w), np.linspace(0, 1, h))
    l_mask = (1.0 - np.clip(20 * (l - 0.05), 0, 1))[None, ...]
    r_mask = l_mask[:, :, ::-1]
    return r_mask * l_disp + l_mask * r_disp + (1.0 - l_mask - r_mask) * m_disp


def resize_linear(src, dst_w, dst_h):
    """cv2.resize(src, (dst_w, dst_h)) with the default INTER_LINEAR for a 2-D float array: source coordinate of destination pixel
    d is (d + 0.5) * (src / dst) - 0.5, clamped to the image; the two interpolation weights are float32 (OpenCV computes `fx` in
    float), the weighted sums run in the array's own precision (double for a float64 array), horizontal pass first."""
    h, w = src.shape

    def axis(n_src, n_dst):
        scale = float(n_src) / float(n_dst)
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        f = f - i0.astype(np.float32)
        lo, hi = i0 < 0, i0 >= n_src - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        i0 = np.where(lo, 0, np.where(hi, n_src - 1, i0))
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, (np.float32(1) - f).astype(np.float32), f

    x0, x1, ax0, ax1 = axis(w, dst_w)
    y0, y1, ay0, ay1 = axis(h, dst_h)
    s = src.astype(np.float64)
    rows = s[:, x0] * ax0.astype(np.float64) + s[:, x1] * ax1.astype(np.float64)
    return rows[y0] * ay0.astype(np.float64)[:, None] + rows[y1] * ay1.astype(np.float64)[:, None]


def eval_image(pred_disp, gt_depth, eval_split="eigen", pred_depth_scale_factor=1.0, disable_median_scaling=False):
    """the per-image body of evaluate() (evaluate_depth_config.py:225-261): resize the prediction to the ground truth's size — the
    SQLdepth head's output is used as depth directly (:231) —, Garg/Eigen crop (:233-241), scale, median scaling (:252-255), clamp
    (:257-258), compute_errors.  -> (7 metrics, ratio or None, number of valid pixels)"""
    gt_h, gt_w = gt_depth.shape[:2]
    pred_depth = resize_linear(pred_disp, gt_w, gt_h)
    if eval_split == "eigen":
        mask = np.logical_and(gt_depth > MIN_DEPTH, gt_depth < MAX_DEPTH)
        crop = np.array([0.40810811 * gt_h, 0.99189189 * gt_h, 0.03594771 * gt_w, 0.96405229 * gt_w]).astype(np.int32)
        crop_mask = np.zeros(mask.shape)
        crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
        mask = np.logical_and(mask, crop_mask)
    else:
        mask = gt_depth > 0
    pred = pred_depth[mask]
    gt = gt_depth[mask]
    pred = pred * pred_depth_scale_factor
    ratio = None
    if not disable_median_scaling:
        ratio = np.median(gt) / np.median(pred)
        pred = pred * ratio
    pred = np.clip(pred, MIN_DEPTH, MAX_DEPTH)          # (:257-258: two masked assignments)
    return compute_errors(gt, pred), ratio, int(mask.sum())
