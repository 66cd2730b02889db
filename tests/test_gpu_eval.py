"""GPU parity of the depth-evaluation kernels (csrc/eval.hip through the C ABI: flip post-processing, per-image metrics with
median scaling) against the reference's own vectors (golden group G19) and the oracle (oracle/eval_ref.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_post_process_golden_g19(golden):
    from sqd import ops
    g = golden("g19_eval")
    l, r = g["l_disp"], g["r_disp"]
    # the reference hands batch_post_process_disparity the second pass flipped back (evaluate_depth_config.py:152); the kernel takes
    # the raw second pass, i.e. r flipped
    disp = torch.from_numpy(np.concatenate([l, r[:, :, ::-1]], 0).copy()).cuda()
    out = ops.disp_post_process(disp).cpu().numpy()
    assert out.dtype == np.float64 and out.shape == g["post"].shape
    assert np.abs(out - g["post"]).max() <= 1e-14 * np.abs(g["post"]).max()


def test_compute_errors_golden_g19(golden):
    """identity resize (prediction at the ground truth's size), no crop, no median scaling: the kernel reduces to compute_errors"""
    from sqd import ops
    g = golden("g19_eval")
    gt, pred = g["gt"].reshape(50, 100), g["pred"].reshape(50, 100)
    out = ops.depth_eval(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), eval_split="none", median_scaling=False).cpu().numpy()
    assert out[8] == gt.size and np.isnan(out[7])
    # rmse_log goes through numpy's float32 log of the float32 ground truth (np.log keeps the dtype): the device's logf differs from
    # it in the last bit of single values, 1e-7 relative on the mean; everything else is double arithmetic on the same numbers
    np.testing.assert_allclose(out[[0, 1, 2, 4, 5, 6]], g["errors"][[0, 1, 2, 4, 5, 6]], rtol=1e-12, atol=0)
    np.testing.assert_allclose(out[3], g["errors"][3], rtol=5e-7, atol=0)


def _sparse_gt(rs, Hg, Wg, density):
    gt = rs.uniform(0.5, 90.0, (Hg, Wg)).astype(np.float32)          # some beyond MAX_DEPTH = 80
    gt[rs.uniform(size=(Hg, Wg)) > density] = 0.0
    return gt


@pytest.mark.parametrize("h,w,Hg,Wg,split,med,scale,density", [
    (192, 640, 375, 1242, "eigen", True, 1.0, 0.05), (192, 640, 370, 1226, "eigen", True, 1.0, 0.05), (96, 320, 375, 1242, "eigen", False, 5.4, 0.04),
    (192, 640, 375, 1242, "eigen_benchmark", True, 1.0, 0.2), (24, 80, 37, 121, "eigen", True, 1.0, 0.5), (24, 80, 24, 80, "eigen", True, 2.0, 1.0),
    (320, 1024, 376, 1241, "eigen", True, 1.0, 0.05), (7, 9, 50, 70, "eigen_benchmark", True, 1.0, 0.31)])
def test_depth_eval_vs_oracle(h, w, Hg, Wg, split, med, scale, density):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import eval_ref as R
    from sqd import ops
    rs = np.random.RandomState(h * 7 + Wg)
    pred = (rs.uniform(1.0, 60.0, (h, w)) * rs.uniform(0.3, 0.6)).astype(np.float64)      # off-scale: median scaling matters
    gt = _sparse_gt(rs, Hg, Wg, density)
    want, ratio, n = R.eval_image(pred, gt, eval_split=split, pred_depth_scale_factor=scale, disable_median_scaling=not med)
    out = ops.depth_eval(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda(), eval_split=split,
                         pred_depth_scale_factor=scale, median_scaling=med).cpu().numpy()
    assert int(out[8]) == n and n > 0
    if med:
        assert abs(out[7] - ratio) <= 1e-12 * abs(ratio), (out[7], ratio)          # exact medians
    else:
        assert np.isnan(out[7])
    want = np.array(want, dtype=np.float64)
    np.testing.assert_allclose(out[[0, 1, 2, 4, 5, 6]], want[[0, 1, 2, 4, 5, 6]], rtol=1e-10, atol=0)
    np.testing.assert_allclose(out[3], want[3], rtol=5e-7, atol=0)             # (float32 log of the ground truth, see above)


def test_depth_eval_without_valid_pixels():
    from sqd import ops
    out = ops.depth_eval(torch.ones(8, 8, device="cuda"), torch.zeros(20, 30, device="cuda")).cpu().numpy()
    assert out[8] == 0 and np.isnan(out[:8]).all()


def test_evaluate_flow_matches_oracle_per_image():
    """evaluate_depth.evaluate — networks, flip post-processing, per-image metrics, all on the device — against the same flow with
    the oracle's numpy functions on the host (random weights: the numbers are meaningless, the arithmetic is not)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import evaluate_depth as E
    from datasets.synthetic import synthetic_eval_set
    from options import MonodepthOptions
    from oracle import eval_ref as R
    torch.manual_seed(0)
    opt = MonodepthOptions().parse(["--backbone", "resnet18_lite", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
                                    "--height", "64", "--width", "96", "--batch_size", "2", "--sqd_synthetic", "--post_process",
                                    "--max_depth", "80.0", "--sqd_no_conv_tune"])
    dev = torch.device("cuda")
    frames, gts = synthetic_eval_set(3, 64, 96, dev)
    enc, dep = E.build_models(opt, dev)
    res = E.evaluate(opt, frames, gts, enc, dep)
    assert res["errors"].shape == (3, 7) and np.isfinite(res["errors"]).all()
    with torch.no_grad():
        x = torch.cat((frames, torch.flip(frames, [3])), 0).contiguous(memory_format=torch.channels_last)
        raw = dep(enc(x))[("disp", 0)][:, 0].cpu().numpy()
    post = R.batch_post_process_disparity(raw[:3], raw[3:, :, ::-1])
    for i in range(3):
        want, ratio, n = R.eval_image(post[i], gts[i].cpu().numpy())
        assert int(res["valid"][i]) == n
        # (the two network passes batch the frames differently — 2 + 1 images vs 3 — so their fp32 outputs differ in the last bits;
        #  the kernels' own exactness is test_depth_eval_vs_oracle's subject)
        assert abs(res["ratios"][i] - ratio) <= 1e-6 * abs(ratio)
        np.testing.assert_allclose(res["errors"][i], np.array(want), rtol=2e-5)
    np.testing.assert_allclose(res["mean_errors"], res["errors"].mean(0))
