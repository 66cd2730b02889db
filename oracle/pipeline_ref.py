"""TEST INFRASTRUCTURE — CPU restatement (numpy, integer / float32 arithmetic spelled out) of the per-frame preprocessing of the
reference's `MonoDataset` (`/root/reference/datasets/mono_dataset.py:90-201`): horizontal flip (:153-154,163), resize of the native
frame to (width, height) with `Image.ANTIALIAS` (:57,73-78 — PIL's Lanczos filter, called LANCZOS since Pillow 10), torchvision's
`ColorJitter` on PIL images (:64-71,179-181; brightness / contrast / saturation / hue in a random order) and `ToTensor` (:107-108).
Only tests import this module.

Pinned: every function below is checked against PIL itself (`tests/test_oracle_pipeline_vs_pil.py`: the resize against
`Image.resize(..., LANCZOS)`, the blends against `ImageEnhance`, the colour conversions against `Image.convert` over ALL 2^24
colours) — PIL is what the reference and torchvision call.  Parity unpinned: the *composition* in `color_jitter` (which PIL
operations torchvision.transforms.functional applies for each factor, and `ColorJitter`'s order handling) is restated from
torchvision's published source; torchvision is not installed here."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2            # Pillow src/libImaging/Resample.c: 8 bits for the result, 2 for overflow (Lanczos lobes)


def _lanczos(x):
    """Resample.c lanczos_filter: truncated sinc, support 3"""
    def sinc(v):
        if v == 0.0:
            return 1.0
        v = v * np.pi
        return np.sin(v) / v
    if -3.0 <= x < 3.0:
        return sinc(x) * sinc(x / 3.0)
    return 0.0


def lanczos_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the LANCZOS filter, whole-image box ->
    (bounds [out, 2] int32 = (first source index, tap count), coefficients [out, ksize] int32 fixed point)"""
    scale = filterscale = float(in_size) / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            w = _lanczos((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    # normalize_coeffs_8bpc: (int)(+-0.5 + k * 2^PRECISION_BITS), truncation toward zero
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS))).astype(np.int32)
    return bounds, fixed


def _clip8(acc):
    """Resample.c clip8: (acc >> PRECISION_BITS) through the clamping lookup table"""
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_lanczos(img, out_w, out_h, flip=False):
    """img [H0, W0, C] uint8 -> [out_h, out_w, C] uint8 as `Image.resize((out_w, out_h), Image.ANTIALIAS)` computes it: a horizontal
    pass into an 8-bit image, then a vertical pass (ImagingResample; either pass is skipped when the size does not change).
    flip: the frame is mirrored left-right first (mono_dataset.py:163 `color.transpose(FLIP_LEFT_RIGHT)`)."""
    if flip:
        img = img[:, ::-1]
    h0, w0 = img.shape[:2]
    cur = img.astype(np.int64)
    if w0 != out_w:
        b, k = lanczos_coeffs(w0, out_w)
        out = np.empty((h0, out_w, img.shape[2]), np.uint8)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = (cur[:, x0:x0 + n, :] * k[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, xx, :] = _clip8(acc)
        cur = out.astype(np.int64)
    if h0 != out_h:
        b, k = lanczos_coeffs(h0, out_h)
        out = np.empty((out_h, cur.shape[1], img.shape[2]), np.uint8)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = (cur[y0:y0 + n, :, :] * k[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur.astype(np.uint8)


def blend(deg, img, alpha):
    """libImaging/Blend.c ImagingBlend(imIn1 = deg, imIn2 = img, alpha) on uint8 arrays: float32 arithmetic, truncation to 8 bits;
    outside [0, 1] the result is clipped first.  (ImageEnhance._Enhance.enhance = Image.blend(degenerate, image, factor))"""
    a = np.float32(alpha)
    d = deg.astype(np.int32)
    t = d.astype(np.float32) + a * (img.astype(np.int32) - d).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)                  # (UINT8) of a value inside [0, 255]: truncation
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def rgb_to_l(img):
    """libImaging/Convert.c rgb2l / L24: ITU-R 601-2 luma in 16.16 fixed point"""
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def rgb_to_hsv(img):
    """libImaging/Convert.c rgb2hsv_row (float32 arithmetic as the C code's `float` variables)"""
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    f32 = np.float32
    cr = (maxc - minc).astype(f32)
    safe_cr = np.where(cr == 0, f32(1), cr)
    safe_max = np.where(maxc == 0, 1, maxc).astype(f32)
    s = cr / safe_max
    rc, gc, bc = ((maxc - c).astype(f32) / safe_cr for c in (r, g, b))
    # `h = bc - gc` is float arithmetic; `2.0 + rc - bc` / `4.0 + gc - rc` start from a double literal: evaluated in double, then
    # narrowed to the float variable
    d = np.float64
    h = np.where(r == maxc, (bc - gc).astype(d), np.where(g == maxc, 2.0 + rc.astype(d) - bc.astype(d), 4.0 + gc.astype(d) - rc.astype(d))).astype(f32)
    # h = fmod((h / 6.0 + 1.0), 1.0): the C expression promotes to double (6.0, 1.0 are double literals), then narrows to float
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    grey = minc == maxc
    return np.stack([np.where(grey, 0, uh), np.where(grey, 0, us), maxc], -1).astype(np.uint8)


def hsv_to_rgb(hsv):
    """libImaging/Convert.c hsv2rgb (following colorsys.py; float arithmetic with double literals, round())"""
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    vf = v.astype(np.float32).astype(np.float64)
    f64, fs64 = f.astype(np.float64), fs.astype(np.float64)

    def c_round(x):                                        # C round(): half away from zero (values are >= 0 here)
        return np.floor(x + 0.5).astype(np.int32)
    p = np.clip(c_round(vf * (1.0 - fs64)), 0, 255)
    q = np.clip(c_round(vf * (1.0 - fs64 * f64)), 0, 255)
    t = np.clip(c_round(vf * (1.0 - fs64 * (1.0 - f64))), 0, 255)
    sel = i % 6
    r = np.choose(sel, [v, q, p, p, t, v])
    g = np.choose(sel, [t, v, v, q, p, p])
    b = np.choose(sel, [p, p, t, v, v, q])
    grey = s == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], -1).astype(np.uint8)


def adjust_brightness(img, factor):
    """torchvision.transforms.functional_pil.adjust_brightness = ImageEnhance.Brightness(img).enhance(factor): blend with black"""
    return blend(np.zeros_like(img), img, factor)


def adjust_contrast(img, factor):
    """ImageEnhance.Contrast: degenerate = the grey level int(mean(L) + 0.5) everywhere"""
    mean = int(rgb_to_l(img).astype(np.float64).sum() / (img.shape[0] * img.shape[1]) + 0.5)
    return blend(np.full_like(img, mean), img, factor)


def adjust_saturation(img, factor):
    """ImageEnhance.Color: degenerate = the L image replicated to RGB"""
    return blend(np.repeat(rgb_to_l(img)[..., None], 3, -1), img, factor)


def adjust_hue(img, hue_factor):
    """functional_pil.adjust_hue: HSV, `np_h += np.uint8(hue_factor * 255)` with uint8 wrap-around, back to RGB"""
    hsv = rgb_to_hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + int(np.array(hue_factor * 255).astype(np.uint8))) & 255
    return hsv_to_rgb(hsv)


def color_jitter(img, order, brightness, contrast, saturation, hue):
    """torchvision ColorJitter.forward on a PIL image with sampled parameters: `order` is the permutation of (0 brightness,
    1 contrast, 2 saturation, 3 hue) drawn by get_params; a factor of None skips its operation."""
    for fn in order:
        if fn == 0 and brightness is not None:
            img = adjust_brightness(img, brightness)
        elif fn == 1 and contrast is not None:
            img = adjust_contrast(img, contrast)
        elif fn == 2 and saturation is not None:
            img = adjust_saturation(img, saturation)
        elif fn == 3 and hue is not None:
            img = adjust_hue(img, hue)
    return img


def to_tensor(img):
    """torchvision ToTensor on a PIL RGB image: [H, W, 3] uint8 -> [3, H, W] float32 = value / 255 (float32 division)"""
    return (img.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).astype(np.float32)


def preprocess_frame(raw, out_w, out_h, flip, aug):
    """one frame of MonoDataset.__getitem__ + preprocess: raw [H0, W0, 3] uint8 -> (color [3,H,W], color_aug [3,H,W]) float32.
    aug = None (the `lambda x: x` branch, mono_dataset.py:182-183) or (order, brightness, contrast, saturation, hue)."""
    small = resize_lanczos(raw, out_w, out_h, flip)
    jit = small if aug is None else color_jitter(small, *aug)
    return to_tensor(small), to_tensor(jit)
