"""GPU parity of the device-side input pipeline (csrc/input_pipeline.hip through the C ABI, sqd/pipeline.py) against the oracle
(oracle/pipeline_ref.py, itself pinned to PIL in tests/test_oracle_pipeline_vs_pil.py): every byte equal."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _frames(rs, n, h0, w0):
    img = rs.randint(0, 256, (n, h0, w0, 3)).astype(np.uint8)
    img[:, : h0 // 3, : w0 // 3] = rs.randint(0, 2, (n, h0 // 3, w0 // 3, 1)) * 255        # hard edges: negative lobes clip
    img[:, h0 // 2:, w0 // 2:] = (img[:, h0 // 2:, w0 // 2:] // 32) * 32                   # flat patches: grey / low-saturation pixels
    return img


@pytest.mark.parametrize("h0,w0,h,w", [(375, 1242, 192, 640), (370, 1226, 192, 640), (376, 1241, 320, 1024), (37, 121, 24, 80),
                                       (24, 80, 24, 80), (20, 30, 40, 30), (50, 64, 50, 100)])
def test_resize_bytes_equal_oracle(h0, w0, h, w):
    from oracle import pipeline_ref as R
    from sqd.pipeline import DevicePreprocess
    rs = np.random.RandomState(h0 + w)
    img = _frames(rs, 3, h0, w0)
    flip = np.array([False, True, False])
    got = DevicePreprocess(h, w).resize(torch.from_numpy(img).cuda(), flip).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], R.resize_lanczos(img[i], w, h, bool(flip[i]))), i


def test_coefficient_tables_equal_oracle():
    from oracle import pipeline_ref as R
    from sqd.pipeline import lanczos_tables
    for a, b in ((1242, 640), (375, 192), (1241, 1024), (376, 320), (121, 80), (20, 40)):
        bo, co = R.lanczos_coeffs(a, b)
        b2, c2, ks = lanczos_tables(a, b)
        assert ks == co.shape[1] and np.array_equal(bo, b2) and np.array_equal(co, c2)


def test_color_jitter_bytes_equal_oracle():
    from oracle import pipeline_ref as R
    from sqd.pipeline import DevicePreprocess
    rs = np.random.RandomState(11)
    img = _frames(rs, 7, 48, 64)
    aug = [([0, 1, 2, 3], 0.8, 1.2, 0.9, 0.1), ([3, 2, 1, 0], 1.2, 0.8, 1.15, -0.1), None, ([2, 0, 3, 1], 1.0, 1.0563, 0.8137, 0.0371),
           ([1, 3, 0, 2], 0.9137, None, 1.2, -0.0463), ([3, 1, 2, 0], None, None, None, 0.05), ([0, 2, 1, 3], 1.7, 0.0, 1.9, None)]
    got = DevicePreprocess(48, 64).color_jitter(torch.from_numpy(img).cuda(), aug).cpu().numpy()
    for i, a in enumerate(aug):
        want = img[i] if a is None else R.color_jitter(img[i], *a)
        assert np.array_equal(got[i], want), (i, np.abs(got[i].astype(int) - want.astype(int)).max())


def test_all_colours_through_the_hue_step():
    """every RGB triple through RGB -> HSV -> +shift -> RGB on the device against the oracle (which equals PIL on all of them)"""
    from oracle import pipeline_ref as R
    from sqd.pipeline import DevicePreprocess
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(1, 4096, 4096, 3)
    got = DevicePreprocess(4096, 4096).color_jitter(torch.from_numpy(rgb).cuda(), [([3, 0, 1, 2], None, None, None, 0.0371)]).cpu().numpy()
    assert np.array_equal(got[0], R.adjust_hue(rgb[0], 0.0371))


def test_preprocess_batch_equals_oracle():
    from oracle import pipeline_ref as R
    from sqd.pipeline import DevicePreprocess, draw_params
    rs = np.random.RandomState(3)
    B, F, h0, w0, h, w = 3, 3, 75, 248, 48, 160
    raw = _frames(rs, B * F, h0, w0).reshape(B, F, h0, w0, 3)
    rng = np.random.default_rng(5)
    params = [draw_params(rng) for _ in range(B)]
    params[0] = (True, params[0][1] or ([2, 1, 3, 0], 1.1, 0.9, 1.05, -0.03))        # make sure both branches occur
    params[1] = (False, None)
    out = DevicePreprocess(h, w)(torch.from_numpy(raw).cuda(), [p[0] for p in params], [p[1] for p in params])
    color, aug = out["color"].cpu().numpy(), out["color_aug"].cpu().numpy()
    assert color.shape == (B, F, 3, h, w) and color.dtype == np.float32
    for b in range(B):
        for f in range(F):
            c, a = R.preprocess_frame(raw[b, f], w, h, params[b][0], params[b][1])
            assert np.array_equal(color[b, f], c) and np.array_equal(aug[b, f], a), (b, f)
