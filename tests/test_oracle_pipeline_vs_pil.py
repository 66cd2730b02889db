"""oracle/pipeline_ref.py against PIL itself (the library the reference's MonoDataset and torchvision's ColorJitter call):
Lanczos resize, ImageEnhance blends, RGB<->HSV over all 2^24 colours.  CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance  # noqa: E402

from oracle import pipeline_ref as R  # noqa: E402


@pytest.mark.parametrize("h0,w0,h,w", [(375, 1242, 192, 640), (370, 1226, 192, 640), (376, 1241, 320, 1024), (37, 121, 24, 80),
                                       (24, 80, 24, 80), (20, 30, 40, 30), (50, 64, 50, 100)])
def test_resize_lanczos_matches_pil(h0, w0, h, w):
    rs = np.random.RandomState(h0 + w)
    img = rs.randint(0, 256, (h0, w0, 3)).astype(np.uint8)
    img[: h0 // 3, : w0 // 3] = rs.randint(0, 2, (h0 // 3, w0 // 3, 1)) * 255       # hard edges: the filter's negative lobes clip
    want = np.asarray(Image.fromarray(img).resize((w, h), Image.LANCZOS))
    assert np.array_equal(R.resize_lanczos(img, w, h), want)
    wantf = np.asarray(Image.fromarray(img).transpose(Image.FLIP_LEFT_RIGHT).resize((w, h), Image.LANCZOS))
    assert np.array_equal(R.resize_lanczos(img, w, h, flip=True), wantf)


def test_blend_and_enhancers_match_pil():
    rs = np.random.RandomState(5)
    img = rs.randint(0, 256, (64, 96, 3)).astype(np.uint8)
    pil = Image.fromarray(img)
    for f in (0.8, 0.9137, 1.0, 1.0563, 1.2, 0.0, 1.7):
        assert np.array_equal(R.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f))), f
        assert np.array_equal(R.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f))), f
        assert np.array_equal(R.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pil).enhance(f))), f
    assert np.array_equal(R.rgb_to_l(img), np.asarray(pil.convert("L")))
    # every (degenerate, value) pair of the blend at a few factors
    a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8))
    A, B = Image.fromarray(a), Image.fromarray(b)
    for f in (0.8, 0.95, 1.05, 1.2, 1.9):
        assert np.array_equal(R.blend(a, b, f), np.asarray(Image.blend(A, B, f))), f


def test_hsv_conversions_match_pil_on_all_colours():
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    hsv = np.asarray(Image.fromarray(rgb).convert("HSV"))
    assert np.array_equal(R.rgb_to_hsv(rgb), hsv)
    back = np.asarray(Image.fromarray(rgb, "HSV").convert("RGB"))               # the same 2^24 triples read as HSV
    assert np.array_equal(R.hsv_to_rgb(rgb), back)


def test_adjust_hue_matches_the_pil_recipe():
    """torchvision.transforms.functional_pil.adjust_hue spelled with PIL + numpy (its published body)"""
    rs = np.random.RandomState(9)
    img = rs.randint(0, 256, (48, 64, 3)).astype(np.uint8)
    for hf in (-0.1, -0.037, 0.0, 0.05, 0.1):
        h, s, vv = Image.fromarray(img).convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            np_h += np.array(hf * 255).astype(np.uint8)
        want = np.asarray(Image.merge("HSV", (Image.fromarray(np_h, "L"), s, vv)).convert("RGB"))
        assert np.array_equal(R.adjust_hue(img, hf), want), hf
