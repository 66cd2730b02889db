"""Deterministic, torch-RNG-independent parameter / input generators shared by the golden-vector
generator (make_goldens.py, runs only where /root/reference exists) and by the tests that replay
the fixtures.  numpy's legacy RandomState stream is stable across numpy versions, so fixtures only
need to store a seed instead of megabytes of weights."""
import math

import numpy as np
import torch


def fill_params(module: torch.nn.Module, seed: int) -> torch.nn.Module:
    """Overwrite every parameter/buffer of `module` (state_dict order) with seeded values."""
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if name.endswith("num_batches_tracked"):
                t.zero_()
                continue
            shape = tuple(t.shape)
            if name.endswith("running_var"):
                v = rs.uniform(0.5, 1.5, shape)
            elif name.endswith("running_mean"):
                v = rs.uniform(-0.1, 0.1, shape)
            elif name.endswith("positional_encodings"):
                v = rs.uniform(0.0, 1.0, shape)
            elif t.ndim >= 2:
                fan_in = int(np.prod(shape[1:]))
                v = rs.uniform(-1.0, 1.0, shape) * math.sqrt(3.0 / fan_in)
            elif name.endswith("weight"):          # 1-D weights only occur in BN / LayerNorm
                v = rs.uniform(0.5, 1.5, shape)
            else:                                  # biases
                v = rs.uniform(-0.1, 0.1, shape)
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape))
    return module


def smooth_images(rs: np.random.RandomState, B: int, H: int, W: int, C: int = 3, noise: float = 0.05):
    """Low-frequency sinusoid texture + a little white noise, clipped to [0,1]  -> float32 [B,C,H,W]."""
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    img = np.zeros((B, C, H, W))
    for b in range(B):
        for c in range(C):
            acc = np.full((H, W), 0.5)
            for _ in range(4):
                fx, fy = rs.uniform(0.02, 0.25, 2)
                ph = rs.uniform(0, 2 * math.pi)
                acc += rs.uniform(0.05, 0.15) * np.sin(fx * xs + fy * ys + ph)
            img[b, c] = acc
    img += noise * rs.uniform(-1, 1, img.shape)
    return np.clip(img, 0.0, 1.0).astype(np.float32)


def kitti_K(B: int, H: int, W: int):
    """KITTI normalised intrinsics scaled to the image (reference datasets/kitti_dataset.py:29-32,
    datasets/mono_dataset.py:166-175) and its pseudo-inverse."""
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= W
    K[1, :] *= H
    inv_K = np.linalg.pinv(K)
    return (np.repeat(K[None], B, 0).astype(np.float32), np.repeat(inv_K[None], B, 0).astype(np.float32))


def chain_inputs(seed: int, B: int, H: int, W: int, S: int = 2, pose_scale: float = 0.01):
    """Inputs of the photometric chain at one scale: half-res depth in [1,21], three frames,
    small poses, tie-break noise.  Returns a dict of float32 numpy arrays."""
    rs = np.random.RandomState(seed)
    frames = smooth_images(rs, B * (S + 1), H, W).reshape(S + 1, B, 3, H, W)
    ys, xs = np.meshgrid(np.arange(H // 2), np.arange(W // 2), indexing="ij")
    disp = np.zeros((B, 1, H // 2, W // 2))
    for b in range(B):
        disp[b, 0] = 11.0 + 8.0 * np.sin(0.05 * xs + rs.uniform(0, 6)) * np.cos(0.07 * ys + rs.uniform(0, 6)) \
            + rs.uniform(-2, 2, xs.shape)
    disp = np.clip(disp, 1.0, 21.0).astype(np.float32)
    K, inv_K = kitti_K(B, H, W)
    d = {"disp": disp, "K": K, "inv_K": inv_K, "color0": frames[0],
         "noise": rs.standard_normal((B, S, H, W)).astype(np.float32)}
    for s in range(S):
        d["color_s%d" % s] = frames[s + 1]
        d["axisangle_s%d" % s] = (pose_scale * rs.standard_normal((B, 1, 1, 3))).astype(np.float32)
        d["translation_s%d" % s] = (pose_scale * 50 * rs.standard_normal((B, 1, 1, 3))).astype(np.float32)
    return d


def decoder_feats(seed: int, chans, h0: int, w0: int, B: int = 2):
    """Synthetic 5-level feature pyramid (level i at h0>>i × w0>>i) for the DecoderBN goldens."""
    rs = np.random.RandomState(seed)
    return [(0.5 * rs.standard_normal((B, c, h0 >> i, w0 >> i))).astype(np.float32) for i, c in enumerate(chans)]


def sparse_gt(seed: int, B: int):
    """Velodyne-like sparse ground-truth depth [B,1,375,1242] (≈30 % valid, values in (0,80))."""
    rs = np.random.RandomState(seed)
    gt = rs.uniform(0.5, 80, (B, 1, 375, 1242)).astype(np.float32)
    gt[rs.uniform(size=gt.shape) < 0.7] = 0
    return gt


def pose_input_case(seed: int, B: int, H: int, W: int, frame_ids, stereo: bool):
    """Inputs of fixture group G24 (predict_poses -> generate_images_pred -> compute_losses under the pose-input variants): the frames
    of chain_inputs() keyed as the trainer's inputs dict (numpy arrays), colour-augmented copies, stereo_T, disp, noise."""
    fids = list(frame_ids) + (["s"] if stereo else [])
    d = chain_inputs(seed, B, H, W, S=len(fids) - 1)
    rs = np.random.RandomState(seed + 1)
    inputs = {("K", 0): d["K"], ("inv_K", 0): d["inv_K"]}
    for i, f in enumerate(fids):
        img = d["color0"] if f == 0 else d["color_s%d" % (i - 1)]
        inputs[("color", f, 0)] = img
        inputs[("color_aug", f, 0)] = np.clip(img * rs.uniform(0.9, 1.1) + rs.uniform(-0.03, 0.03), 0, 1).astype(np.float32)
    stereo_T = np.repeat(np.eye(4, dtype=np.float32)[None], B, 0)
    stereo_T[0, 0, 3], stereo_T[1, 0, 3] = -0.1, 0.1
    if stereo:
        inputs["stereo_T"] = stereo_T
    return fids, inputs, d["disp"], d["noise"]
