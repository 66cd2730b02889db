"""Synthetic KITTI-shaped frames with the dict schema of the reference's MonoDataset.__getitem__
(reference datasets/mono_dataset.py:114-201): ("color", f, 0), ("color_aug", f, 0) for every frame
id, ("K", 0), ("inv_K", 0), "depth_gt", and — with "s" among the frame ids — "stereo_T".

Each sample is a smooth random texture (a few low-frequency sinusoids + 5 % white noise); the -1/+1
frames are the same texture seen through a known small camera motion over a known smooth depth
field in [2, 60] m, so the photometric loss has a meaningful minimum.  Intrinsics are
the normalised KITTI K scaled by the image size (reference datasets/kitti_dataset.py:29-32).

scene="waves" (default): the depth field has random phases and nothing in the image tells them — a network cannot learn it, abs_rel stays
at what a constant prediction gets (~0.9).  scene="road": a driving-scene layout the image does tell — a ground plane below a horizon row
that varies per sample (camera 1.65 m above it, as on KITTI's car) and a far wall above it, the region above the horizon brighter and the
ground carrying a depth-scaled stripe pattern — so that self-supervised training brings abs_rel down (tests/test_gpu_abs_rel.py's
comparison from trained weights).  scene="drive" (round 6): the CONSISTENT scene — texture and depth are analytic functions of the continuous
target coordinates, and a source frame's pixel y shows the texture of the target point x whose 3-D point projects to y (x found by
fixed-point iteration; round 5 sampled the target texture at the FORWARD projection, which is consistent to first order only), rotations are
exact, nothing is added after the warp.  The photometric loss of the reference then has its minimum at the true depth and the true motion up to
the bilinear interpolation error of a smooth function: a trained model stays trained under the default learning rate."""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

_K_NORM = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def intrinsics(height, width):
    K = _K_NORM.copy()
    K[0, :] *= width
    K[1, :] *= height
    return torch.from_numpy(K), torch.from_numpy(np.linalg.pinv(K).astype(np.float32))


def _texture(gen, height, width, xs, ys):
    img = torch.full((3, height, width), 0.5)
    for c in range(3):
        for _ in range(5):
            fx, fy = (torch.rand(2, generator=gen) * 0.23 + 0.02).tolist()
            ph = float(torch.rand(1, generator=gen)) * 2 * math.pi
            amp = float(torch.rand(1, generator=gen)) * 0.10 + 0.05
            img[c] += amp * torch.sin(fx * xs + fy * ys + ph)
    return img


def _rodrigues(w):
    """exact rotation matrix of the axis-angle vector w (float64)"""
    th = float(w.norm())
    Kx = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
    if th < 1e-12:
        return torch.eye(3, dtype=torch.float64) + Kx
    return torch.eye(3, dtype=torch.float64) + math.sin(th) / th * Kx + (1 - math.cos(th)) / (th * th) * (Kx @ Kx)


def drive_motion(index, f):
    """the camera motion target -> source frame f of sample `index` of the "drive" scene: (axis-angle w, translation t), X_src = R(w) X_tgt + t"""
    gen = torch.Generator().manual_seed(977 + 31 * int(index) + (int(f) if f != "s" else 7))
    if f == "s":
        return torch.zeros(3, dtype=torch.float64), torch.tensor([-0.1, 0.0, 0.0], dtype=torch.float64)
    r = torch.rand(6, generator=gen, dtype=torch.float64)
    t = torch.stack([(r[0] - 0.5) * 0.10, (r[1] - 0.5) * 0.04, -(0.25 + 0.25 * r[2])]) * float(f)     # driving forward: frame -1 lies behind, +1 ahead
    w = (r[3:6] - 0.5) * 0.012 * float(f)
    return w, t


def _drive_sample(index, height, width, frame_ids, with_gt):
    gen = torch.Generator().manual_seed(1234 + int(index))
    dd = torch.float64
    r = torch.rand(40, generator=gen, dtype=dd)
    v0 = 0.30 + 0.10 * float(r[0])                                   # horizon (fraction of the height)
    tilt = (float(r[1]) - 0.5) * 0.06
    fy_n, cam_h = float(_K_NORM[1, 1]), 1.65
    bumps = [(float(r[2 + 4 * i]), 0.45 + 0.4 * float(r[3 + 4 * i]), 0.04 + 0.05 * float(r[4 + 4 * i]), 0.02 + 0.05 * float(r[5 + 4 * i])) for i in range(3)]
    fr = [(0.02 + 0.2 * float(r[14 + 4 * i]), 0.02 + 0.2 * float(r[15 + 4 * i]), 6.283 * float(r[16 + 4 * i]), 0.04 + 0.07 * float(r[17 + 4 * i])) for i in range(6)]

    def inv_depth(x, y):                                             # smooth in the continuous pixel coordinates, defined everywhere
        u, v = (x + 0.5) / width, (y + 0.5) / height
        below = v - v0 - tilt * (u - 0.5)
        ground = torch.nn.functional.softplus(below * 40.0) / 40.0 / (cam_h * fy_n)      # 1 / z of the ground plane, -> 0 above the horizon
        inv = 1.0 / 55.0 + ground
        for (bu, bv, su, amp) in bumps:                              # a few nearer "objects": smooth bumps of inverse depth
            inv = inv + amp * torch.exp(-((u - bu) ** 2) / (2 * su * su) - ((v - bv) ** 2) / (2 * (1.6 * su) ** 2))
        return inv.clamp(1.0 / 60.0, 1.0 / 2.0)

    def texture(x, y):                                               # [3, ...] analytic colours: the image tells the layout (sky brighter, stripes ~ 1/z)
        inv = inv_depth(x, y)
        u, v = (x + 0.5) / width, (y + 0.5) / height
        sky = torch.sigmoid(-(v - v0 - tilt * (u - 0.5)) * 60.0)
        out = []
        for c in range(3):
            img = 0.45 + 0.0 * x
            for i, (fx, fyq, ph, amp) in enumerate(fr):
                img = img + amp * torch.sin(fx * x * (1 + 0.13 * c) + fyq * y + ph + 0.7 * c)
            img = img * (0.8 + 0.3 * sky) + 0.07 * torch.sin(inv * 150.0 + 0.04 * x) * (1 - sky)
            out.append(img)
        return torch.stack(out, 0).clamp(0.02, 0.98)

    ys, xs = torch.meshgrid(torch.arange(height, dtype=dd), torch.arange(width, dtype=dd), indexing="ij")
    K, inv_K = intrinsics(height, width)
    K3, iK3 = K[:3, :3].to(dd), inv_K[:3, :3].to(dd)

    def project(x, y, R, t):                                         # target pixel (x, y) -> its pixel in the source view
        z = 1.0 / inv_depth(x, y)
        ray = torch.stack([iK3[0, 0] * x + iK3[0, 1] * y + iK3[0, 2], iK3[1, 0] * x + iK3[1, 1] * y + iK3[1, 2], torch.ones_like(x)], 0)
        X = ray * z
        p = torch.einsum("ij,j...->i...", K3 @ R, X) + (K3 @ t).reshape(3, *([1] * x.dim()))
        return p[0] / p[2], p[1] / p[2]

    sample = {("K", 0): K, ("inv_K", 0): inv_K}
    for f in frame_ids:
        if f == 0:
            img = texture(xs, ys)
        else:
            w, t = drive_motion(index, f)
            R = _rodrigues(w)
            if f == "s":
                stereo_T = torch.eye(4)
                stereo_T[0, 3] = -0.1
                sample["stereo_T"] = stereo_T
            x, y = xs.clone(), ys.clone()
            for _ in range(6):                                       # the target point whose projection is this source pixel (contraction ~0.05 per step)
                px, py = project(x, y, R, t)
                x, y = x - (px - xs), y - (py - ys)
            img = texture(x, y)
        img = img.to(torch.float32)
        sample[("color", f, 0)] = img
        gain = 0.9 + 0.2 * float(torch.rand(1, generator=gen))
        sample[("color_aug", f, 0)] = (img * gain).clamp(0, 1)
    if with_gt:
        depth = (1.0 / inv_depth(xs, ys)).to(torch.float32)
        sample["depth_gt"] = F.interpolate(depth[None, None], [375, 1242], mode="bilinear", align_corners=False)[0]
    return sample


def make_sample(index, height, width, frame_ids=(0, -1, 1), with_gt=True, scene="waves"):
    if scene == "drive":
        return _drive_sample(index, height, width, frame_ids, with_gt)
    gen = torch.Generator().manual_seed(1234 + int(index))
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32), indexing="ij")
    # a wider canvas so that the source views stay inside the texture
    base = _texture(gen, height, width, xs, ys)
    if scene == "road":
        v = (ys + 0.5) / height
        v0 = 0.30 + 0.12 * float(torch.rand(1, generator=gen))                 # horizon row (fraction of the height)
        tilt = (float(torch.rand(1, generator=gen)) - 0.5) * 0.08              # the horizon is not quite level
        below = v - v0 - tilt * ((xs + 0.5) / width - 0.5)
        depth = (1.65 * _K_NORM[1, 1] / below.clamp(min=1e-3)).clamp(2.0, 60.0)      # z = h f_y / (y - y_horizon)
        sky = (below <= 0).float()
        depth = depth * (1 - sky) + 55.0 * sky
        stripes = 0.08 * torch.sin(40.0 / depth * 6.0 + 0.05 * xs)             # ground markings: their spacing shrinks with distance
        base = (base * (0.75 + 0.35 * sky) + stripes * (1 - sky)).clamp(0, 1)
    else:
        depth = 31.0 + 29.0 * torch.sin(0.011 * xs + float(torch.rand(1, generator=gen)) * 6) * \
            torch.cos(0.023 * ys + float(torch.rand(1, generator=gen)) * 6)
        depth = depth.clamp(2.0, 60.0)
    K, inv_K = intrinsics(height, width)
    sample = {("K", 0): K, ("inv_K", 0): inv_K}
    pix = torch.stack([xs, ys, torch.ones_like(xs)], 0).reshape(3, -1)
    cam = (inv_K[:3, :3] @ pix) * depth.reshape(1, -1)
    for f in frame_ids:
        if f == 0:
            img = base
        else:
            if f == "s":                        # the other camera of the stereo pair: pure baseline shift (mono_dataset.py:193-199, side "l")
                t, w = torch.tensor([-0.1, 0.0, 0.0]), torch.zeros(3)
                stereo_T = torch.eye(4)
                stereo_T[0, 3] = -0.1
                sample["stereo_T"] = stereo_T
            else:
                t = (torch.rand(3, generator=gen) - 0.5) * 0.6 * float(f)       # metres
                w = (torch.rand(3, generator=gen) - 0.5) * 0.02                  # radians (small-angle rotation)
            R = torch.eye(3) + torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            p = K[:3, :3] @ (R @ cam + t.reshape(3, 1))
            u = (p[0] / p[2]).reshape(height, width) / (width - 1) * 2 - 1
            v = (p[1] / p[2]).reshape(height, width) / (height - 1) * 2 - 1
            img = F.grid_sample(base[None], torch.stack([u, v], -1)[None], padding_mode="border", align_corners=True)[0]
        img = (img + 0.05 * (torch.rand(img.shape, generator=gen) - 0.5)).clamp(0, 1)
        sample[("color", f, 0)] = img
        gain = 0.9 + 0.2 * float(torch.rand(1, generator=gen))
        sample[("color_aug", f, 0)] = (img * gain).clamp(0, 1)
    if with_gt:
        sample["depth_gt"] = F.interpolate(depth[None, None], [375, 1242], mode="bilinear", align_corners=False)[0]
    return sample


class SyntheticKITTIDataset(Dataset):
    def __init__(self, height, width, frame_ids=(0, -1, 1), length=240, with_gt=True, offset=0, scene="waves"):
        self.height, self.width, self.frame_ids = height, width, list(frame_ids)
        self.length, self.with_gt, self.offset, self.scene = length, with_gt, offset, scene

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        return make_sample(index + self.offset, self.height, self.width, self.frame_ids, self.with_gt, self.scene)


def synthetic_batch(batch_size, height, width, frame_ids=(0, -1, 1), start=0, device=None, with_gt=False, scene="waves"):
    """A collated batch (dict of stacked tensors) — what the DataLoader would hand to process_batch."""
    samples = [make_sample(start + i, height, width, frame_ids, with_gt, scene) for i in range(batch_size)]
    batch = {k: torch.stack([s[k] for s in samples]) for k in samples[0]}
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    return batch


def synthetic_eval_set(n, height, width, device=None, density=0.05):
    """n test frames [n,3,H,W] and their ground-truth depth maps as the KITTI eigen split stores them: [375,1242] fp32, sparse
    (LiDAR-like: `density` of the pixels valid, the rest 0) -> (frames, [gt_0, ..., gt_{n-1}])"""
    frames, gts = [], []
    for i in range(n):
        s = make_sample(10_000 + i, height, width, (0,), with_gt=True)
        frames.append(s[("color", 0, 0)])
        gen = torch.Generator().manual_seed(77_000 + i)
        gt = s["depth_gt"][0].clone()
        gt[torch.rand(gt.shape, generator=gen) > density] = 0.0
        gts.append(gt)
    frames = torch.stack(frames)
    if device is not None:
        frames, gts = frames.to(device), [g.to(device) for g in gts]
    return frames, gts
