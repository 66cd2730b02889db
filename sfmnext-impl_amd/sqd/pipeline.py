"""Device-side input pipeline: what `MonoDataset.__getitem__` + `preprocess` do per frame on the host in the reference
(datasets/mono_dataset.py:90-201: PIL resize with ANTIALIAS, left-right flip, torchvision ColorJitter, ToTensor), as kernels of
libsqd.so on byte frames that are already on the device — byte-exact with Pillow's arithmetic (csrc/input_pipeline.hip).

    pre = DevicePreprocess(height=192, width=640)
    out = pre(raw, flip, aug)          # raw [B,F,H0,W0,3] uint8 (decoded frames of a sample: frame ids along F), on the device
    out["color"], out["color_aug"]     # [B,F,3,H,W] float32 — ("color", f, 0) / ("color_aug", f, 0) of the reference's batch

flip [B] bool: the sample's `do_flip` draw (:153); aug: None or a list of B entries, each None (the `lambda x: x` branch, :182-183) or
(order, brightness, contrast, saturation, hue) — what `transforms.ColorJitter.get_params` returned for the sample (:179-181; one
draw per sample, applied to all its frames).  Drawing the parameters stays with the caller (`draw_params` mirrors the reference's
ranges, :64-71): the RNG protocol is the data loader's business, the per-pixel work is here."""
import ctypes

import numpy as np
import torch

from . import lib as _l

PRECISION_BITS = 32 - 8 - 2


def lanczos_tables(in_size, out_size):
    """Pillow src/libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for the LANCZOS (= ANTIALIAS) filter over the whole
    image -> (bounds [out,2] int32, coef [out,ksize] int32, ksize); vectorised over the output index."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)                 # (int) truncation of a positive / clamped value
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    k = np.arange(ksize, dtype=np.float64)[None, :]
    x = (k + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale)
    with np.errstate(invalid="ignore", divide="ignore"):
        def sinc(v):
            vp = v * np.pi
            return np.where(v == 0.0, 1.0, np.sin(vp) / np.where(vp == 0.0, 1.0, vp))
        w = np.where((x >= -3.0) & (x < 3.0), sinc(x) * sinc(x / 3.0), 0.0)
    w = np.where(k < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size)
    for j in range(ksize):                                                          # the C loop's left-to-right sum
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS))).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, fixed, ksize


def draw_params(rng):
    """the per-sample draws of the reference (mono_dataset.py:64-71,153-154,179-183) from a numpy Generator: (flip, aug)"""
    do_aug, flip = rng.random() > 0.5, rng.random() > 0.5
    if not do_aug:
        return bool(flip), None
    order = [int(i) for i in rng.permutation(4)]
    return bool(flip), (order, float(rng.uniform(0.8, 1.2)), float(rng.uniform(0.8, 1.2)), float(rng.uniform(0.8, 1.2)),
                        float(rng.uniform(-0.1, 0.1)))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class DevicePreprocess:
    def __init__(self, height, width):
        self.height, self.width = int(height), int(width)
        self._tables = {}

    def _table(self, n_in, n_out, device):
        key = (n_in, n_out, str(device))
        t = self._tables.get(key)
        if t is None:
            b, c, ks = lanczos_tables(n_in, n_out)
            t = self._tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), ks)
        return t

    def resize(self, frames, flip=None):
        """frames [n,H0,W0,3] uint8 on the device, flip [n] bool/uint8 or None -> [n,H,W,3] uint8 (Image.resize(..., ANTIALIAS))"""
        if not (frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3):
            raise RuntimeError("sqd: DevicePreprocess takes [n,H0,W0,3] uint8 frames on the device (no CPU fallback)")
        frames = frames.contiguous()
        n, H0, W0, _ = frames.shape
        H, W, L, dev = self.height, self.width, _l.lib(), frames.device
        fl = None if flip is None else torch.as_tensor(flip, device=dev).to(torch.uint8).contiguous()
        cur = frames
        if W0 != W or fl is not None:
            if W0 == W:                               # Pillow skips the pass; a flipped frame still has to be mirrored
                cur = torch.where(fl.view(n, 1, 1, 1).bool(), torch.flip(cur, [2]), cur).contiguous() if fl is not None else cur
            else:
                b, c, ks = self._table(W0, W, dev)
                out = torch.empty(n, H0, W, 3, device=dev, dtype=torch.uint8)
                _l.check(L.sqd_resample_h_u8(_p(cur), _p(out), _p(b), _p(c), ks, n, H0, W0, W, _p(fl), _stream()), "resample_h")
                cur = out
        if H0 != H:
            b, c, ks = self._table(H0, H, dev)
            out = torch.empty(n, H, W, 3, device=dev, dtype=torch.uint8)
            _l.check(L.sqd_resample_v_u8(_p(cur), _p(out), _p(b), _p(c), ks, n, H0, H, W, _stream()), "resample_v")
            cur = out
        return cur

    def color_jitter(self, frames, aug):
        """frames [n,H,W,3] uint8, aug: list of n entries (None or (order, b, c, s, h)) -> [n,H,W,3] uint8 (torchvision ColorJitter)"""
        n, H, W, _ = frames.shape
        if all(a is None for a in aug):
            return frames
        L, dev = _l.lib(), frames.device
        cur = frames.contiguous()
        for step in range(4):
            op = np.full(n, -1, np.int32)
            fac = np.ones(n, np.float32)
            hs = np.zeros(n, np.int32)
            for i, a in enumerate(aug):
                if a is None:
                    continue
                order, br, co, sa, hu = a
                o = int(order[step])
                val = (br, co, sa, hu)[o]
                if val is None:
                    continue
                op[i] = o
                if o == 3:
                    hs[i] = int(np.array(val * 255).astype(np.uint8))                # np.uint8(hue_factor * 255), wraps
                else:
                    fac[i] = np.float32(val)
            if (op < 0).all():
                continue
            op_t, fac_t, hs_t = (torch.from_numpy(a_).to(dev) for a_ in (op, fac, hs))
            lsum = torch.zeros(n, device=dev, dtype=torch.int64)
            if (op == 1).any():
                _l.check(L.sqd_luma_sum_u8(_p(cur), _p(lsum), n, H * W, _stream()), "luma_sum")
            out = torch.empty_like(cur)
            _l.check(L.sqd_color_jitter_step_u8(_p(cur), _p(out), _p(op_t), _p(fac_t), _p(hs_t), _p(lsum), n, H * W, _stream()), "color_jitter")
            cur = out
        return cur

    def to_tensor(self, frames):
        n, H, W, _ = frames.shape
        out = torch.empty(n, 3, H, W, device=frames.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_u8_to_chw_f32(_p(frames.contiguous()), _p(out), n, H * W, _stream()), "to_tensor")
        return out

    def __call__(self, raw, flip=None, aug=None):
        """raw [B,F,H0,W0,3] uint8 -> {"color": [B,F,3,H,W], "color_aug": [B,F,3,H,W]} float32"""
        B, F = raw.shape[:2]
        frames = raw.reshape((B * F,) + tuple(raw.shape[2:]))
        fl = None if flip is None else torch.as_tensor(flip, device=raw.device).to(torch.uint8).repeat_interleave(F)
        small = self.resize(frames, fl)
        color = self.to_tensor(small)
        if aug is None or all(a is None for a in aug):
            color_aug = color.clone()
        else:
            per_frame = [a for a in aug for _ in range(F)]
            color_aug = self.to_tensor(self.color_jitter(small, per_frame))
        shp = (B, F, 3, self.height, self.width)
        return {"color": color.view(shp), "color_aug": color_aug.view(shp)}
