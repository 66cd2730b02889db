"""ORACLE (test infrastructure — never imported by the product path).

Plain fp32 PyTorch/CPU restatement of the SQLdepth self-supervised training hot path of
hisfog/SfMNeXt-Impl.  Every function cites the reference file:line it restates.  The restatement is
pinned against golden vectors frozen from the *imported* reference (tests/golden/make_goldens.py,
groups G1..G16) by tests/test_oracle_vs_golden.py.

Who may import this module: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product (sfmnext-impl_amd/) must never import it: it fails loudly without the HIP library.

Parity status: pinned for layers.py / trainer.py / networks/{layers,depth_decoder_QTR,
lite_depth_decoder_QTR,pose_cnn,resnet_encoder(DecoderBN, ResnetEncoder.forward)}.
"parity unpinned" for the ResNet trunk arithmetic itself (torchvision 0.9.1 is absent from
/root/reference and from this image; restated from the public ResNet v1.5 definition).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# pose algebra — reference layers.py:75-150
# --------------------------------------------------------------------------------------


def rot_from_axisangle(vec: torch.Tensor) -> torch.Tensor:
    """Rodrigues formula, reference layers.py:111-150.  vec [B,1,3] -> [B,4,4]."""
    angle = torch.norm(vec, 2, 2, True)                      # :116
    axis = vec / (angle + 1e-7)                              # :117
    ca, sa = torch.cos(angle), torch.sin(angle)              # :119-120
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))  # :123-125
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    rot = torch.zeros((vec.shape[0], 4, 4), dtype=vec.dtype, device=vec.device)
    rot[:, 0, 0] = torch.squeeze(x * xC + ca)
    rot[:, 0, 1] = torch.squeeze(xyC - zs)
    rot[:, 0, 2] = torch.squeeze(zxC + ys)
    rot[:, 1, 0] = torch.squeeze(xyC + zs)
    rot[:, 1, 1] = torch.squeeze(y * yC + ca)
    rot[:, 1, 2] = torch.squeeze(yzC - xs)
    rot[:, 2, 0] = torch.squeeze(zxC - ys)
    rot[:, 2, 1] = torch.squeeze(yzC + xs)
    rot[:, 2, 2] = torch.squeeze(z * zC + ca)
    rot[:, 3, 3] = 1
    return rot


def get_translation_matrix(t: torch.Tensor) -> torch.Tensor:
    """reference layers.py:95-108.  t [B,1,3] -> [B,4,4]."""
    T = torch.zeros(t.shape[0], 4, 4, dtype=t.dtype, device=t.device)
    T[:, 0, 0] = T[:, 1, 1] = T[:, 2, 2] = T[:, 3, 3] = 1
    T[:, :3, 3, None] = t.contiguous().view(-1, 3, 1)
    return T


def transformation_from_parameters(axisangle, translation, invert=False):
    """reference layers.py:75-92: M = T·R, or Rᵀ·T(−t) when invert."""
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


# --------------------------------------------------------------------------------------
# geometry — reference layers.py:186-258, trainer.py:386-439
# --------------------------------------------------------------------------------------


def pixel_grid(B: int, H: int, W: int, device=None, dtype=torch.float32) -> torch.Tensor:
    """Homogeneous pixel coordinates (x, y, 1) row-major, reference layers.py:196-208 -> [B,3,HW].
    (dtype: float32 as the reference; the tests' float64 runs of the oracle measure its own rounding noise)"""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype, device=device),
                            torch.arange(W, dtype=dtype, device=device), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=dtype, device=device)], 0)
    return pix.unsqueeze(0).repeat(B, 1, 1)


def backproject_depth(depth: torch.Tensor, inv_K: torch.Tensor) -> torch.Tensor:
    """reference layers.py:210-215.  depth [B,1,H,W], inv_K [B,4,4] -> cam points [B,4,HW]."""
    B, _, H, W = depth.shape
    pix = pixel_grid(B, H, W, depth.device, depth.dtype)
    cam = torch.matmul(inv_K[:, :3, :3], pix)                 # :211
    cam = depth.view(B, 1, -1) * cam                          # :212
    ones = torch.ones(B, 1, H * W, dtype=depth.dtype, device=depth.device)
    return torch.cat([cam, ones], 1)                          # :213


def project_3d(points: torch.Tensor, K: torch.Tensor, T: torch.Tensor, H: int, W: int,
               eps: float = 1e-7) -> torch.Tensor:
    """reference layers.py:247-258 -> normalised sampling grid [B,H,W,2] in ≈[−1,1]."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]                          # :248
    cam = torch.matmul(P, points)                             # :250
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)   # :252
    pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1).clone()    # :253-254
    pix[..., 0] /= W - 1                                      # :255
    pix[..., 1] /= H - 1                                      # :256
    return (pix - 0.5) * 2                                    # :257


def grid_sample_indices(grid: torch.Tensor, H: int, W: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """The integer taps ATen's grid_sampler_2d (align_corners=True, padding border) derives from a
    grid: ix = ((gx+1)/2)·(W−1), clamp to [0,W−1], x0 = floor(ix).  These are the "integer pixel
    indices from Project3D" of BASELINE.json.  Returns int32 (x0, y0) [B,H,W]."""
    ix = ((grid[..., 0] + 1) / 2) * (W - 1)
    iy = ((grid[..., 1] + 1) / 2) * (H - 1)
    ix = ix.clamp(0, W - 1)
    iy = iy.clamp(0, H - 1)
    return ix.floor().to(torch.int32), iy.floor().to(torch.int32)


def upsample_disp(disp: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """reference trainer.py:395-399 (bilinear, align_corners=False). The decoder output *is* depth."""
    return F.interpolate(disp, [H, W], mode="bilinear", align_corners=False)


def generate_images_pred(disp: torch.Tensor, poses: Dict[int, Tuple[torch.Tensor, torch.Tensor]],
                         K: torch.Tensor, inv_K: torch.Tensor, colors: Dict[int, torch.Tensor],
                         frame_ids: Sequence[int], H: int, W: int, stereo_T: torch.Tensor = None,
                         use_stereo: bool = False, cam_T_cam: Dict = None) -> Dict:
    """reference trainer.py:386-439, posecnn branch (scale 0 only).

    poses[f] = (axisangle [B,1,1,3], translation [B,1,1,3]) for the temporal frames; frame id "s" (the other stereo camera)
    takes T = stereo_T (:406-407).  use_stereo switches the mean-inverse-depth scaling of the translation off (:412).
    colors[f] = source image [B,3,H,W].
    Returns dict with ("depth",0,0), ("sample",f,0), ("color",f,0), ("color_identity",f,0), ("T",f)."""
    out = {}
    depth = upsample_disp(disp, H, W)
    out[("depth", 0, 0)] = depth
    for f in frame_ids[1:]:
        if f == "s":
            T = stereo_T                                                       # :406-407
        else:
            axisangle, translation = poses[f]
            if cam_T_cam is not None:                                          # :409 (predict_poses' matrix: --pose_model_input all, :358-359)
                T = cam_T_cam[f]
            else:
                T = transformation_from_parameters(axisangle[:, 0], translation[:, 0], f < 0)   # cam_T_cam, :336-337 / :409
            if not use_stereo:                                                 # :412
                inv_depth = 1 / depth                                          # :417
                mean_inv_depth = inv_depth.mean(3, True).mean(2, True)         # :418
                T = transformation_from_parameters(axisangle[:, 0], translation[:, 0] * mean_inv_depth[:, 0],
                                                   f < 0)                      # :420-421
        cam = backproject_depth(depth, inv_K)                             # :423
        grid = project_3d(cam, K, T, H, W)                                # :425
        out[("T", f)] = T
        out[("sample", f, 0)] = grid
        out[("color", f, 0)] = F.grid_sample(colors[f], grid, padding_mode="border",
                                             align_corners=True)          # :431-435
        out[("color_identity", f, 0)] = colors[f]                          # :437-439
    return out


# --------------------------------------------------------------------------------------
# photometric losses — reference layers.py:13-46,267-280, trainer.py:441-549
# --------------------------------------------------------------------------------------

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


def ssim(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """reference layers.py:13-46: 7×7 box window over a 3-px reflection pad -> clamp((1−SSIM)/2,0,1)."""
    x = F.pad(x, (3, 3, 3, 3), mode="reflect")
    y = F.pad(y, (3, 3, 3, 3), mode="reflect")
    mu_x = F.avg_pool2d(x, 7, 1)
    mu_y = F.avg_pool2d(y, 7, 1)
    sigma_x = F.avg_pool2d(x ** 2, 7, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 7, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 7, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + SSIM_C1) * (2 * sigma_xy + SSIM_C2)
    d = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (sigma_x + sigma_y + SSIM_C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred: torch.Tensor, target: torch.Tensor, no_ssim: bool = False) -> torch.Tensor:
    """reference trainer.py:441-453 -> [B,1,H,W]."""
    l1 = torch.abs(target - pred).mean(1, True)
    if no_ssim:
        return l1
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def smooth_loss(disp: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """reference layers.py:267-280."""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    gdx = gdx * torch.exp(-gix)
    gdy = gdy * torch.exp(-giy)
    return gdx.mean() + gdy.mean()


def compute_losses(disp: torch.Tensor, target: torch.Tensor, warped: Dict[int, torch.Tensor],
                   sources: Dict[int, torch.Tensor], frame_ids: Sequence[int], noise: torch.Tensor,
                   H: int, W: int, disparity_smoothness: float = 1e-3, no_ssim: bool = False,
                   avg_reprojection: bool = False, disable_automasking: bool = False) -> Dict:
    """reference trainer.py:455-549 at scale 0, with its three loss options (defaults: automasking on, per-pixel minimum, SSIM on).

    `noise` stands for `torch.randn(identity_reprojection_loss.shape)` of trainer.py:516 — [B,S,H,W], or [B,1,H,W] under
    avg_reprojection (already un-scaled; the 1e-5 factor is applied here); unused without automasking.  Returns loss, loss/0,
    identity_selection/0 (automasking only, :528-530) and the intermediate maps."""
    srcs = list(frame_ids[1:])
    reproj = torch.cat([reprojection_loss(warped[f], target, no_ssim) for f in srcs], 1)       # :474-478 (:447-451)
    ident = None
    if not disable_automasking:
        ident = torch.cat([reprojection_loss(sources[f], target, no_ssim) for f in srcs], 1)   # :480-487
        if avg_reprojection:
            ident = ident.mean(1, keepdim=True)                                               # :489-490
    reproj_c = reproj.mean(1, keepdim=True) if avg_reprojection else reproj                   # :508-511
    if not disable_automasking:
        ident = ident + noise * 0.00001                                                       # :514-517
        combined = torch.cat((ident, reproj_c), dim=1)                                        # :519
    else:
        combined = reproj_c                                                                   # :521
    if combined.shape[1] == 1:
        to_optimise, idxs = combined, None                                                    # :523-524
    else:
        to_optimise, idxs = torch.min(combined, dim=1)                                        # :526
    out = {}
    if not disable_automasking:
        out["identity_selection/0"] = (idxs > ident.shape[1] - 1).float()                     # :528-530
    loss = to_optimise.mean()                                                                 # :532
    d = disp
    if d.shape[-2:] != target.shape[-2:]:
        d = F.interpolate(d, [H, W], mode="bilinear", align_corners=False)                    # :533-534
    mean_disp = d.mean(2, True).mean(3, True)                                                 # :535
    norm_disp = d / (mean_disp + 1e-7)                                                        # :536
    sm = smooth_loss(norm_disp, target)                                                       # :540
    loss = loss + disparity_smoothness * sm / (2 ** 0)                                        # :542
    out.update({"loss": loss, "loss/0": loss, "reproj": reproj, "identity": ident, "idxs": idxs, "to_optimise": to_optimise, "smooth": sm})
    return out


def photometric_chain(disp, poses, K, inv_K, colors, frame_ids, noise, H, W, disparity_smoothness=1e-3, stereo_T=None,
                      use_stereo=False, cam_T_cam=None, **loss_options):
    """generate_images_pred + compute_losses in one call (what process_batch does after the networks,
    reference trainer.py:296-297)."""
    out = generate_images_pred(disp, poses, K, inv_K, colors, frame_ids, H, W, stereo_T, use_stereo, cam_T_cam)
    warped = {f: out[("color", f, 0)] for f in frame_ids[1:]}
    sources = {f: colors[f] for f in frame_ids[1:]}
    losses = compute_losses(disp, colors[0], warped, sources, frame_ids, noise, H, W, disparity_smoothness, **loss_options)
    out.update(losses)
    return out


# --------------------------------------------------------------------------------------
# depth metrics — reference layers.py:282-300, trainer.py:551-579
# --------------------------------------------------------------------------------------


def compute_depth_errors(gt: torch.Tensor, pred: torch.Tensor):
    """reference layers.py:282-300 -> (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)."""
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def compute_depth_losses(depth_pred: torch.Tensor, depth_gt: torch.Tensor):
    """reference trainer.py:551-579: ↑ to 375×1242, clamp, eigen crop, batch-global median scaling."""
    pred = torch.clamp(F.interpolate(depth_pred, [375, 1242], mode="bilinear", align_corners=False),
                       1e-3, 80).detach()
    mask = depth_gt > 0
    crop = torch.zeros_like(mask)
    crop[:, :, 153:371, 44:1197] = 1
    mask = mask * crop
    gt = depth_gt[mask]
    pred = pred[mask]
    pred = pred * (torch.median(gt) / torch.median(pred))
    pred = torch.clamp(pred, min=1e-3, max=80)
    return compute_depth_errors(gt, pred)


# --------------------------------------------------------------------------------------
# Self Query Layer — reference networks/layers.py:7-21
# --------------------------------------------------------------------------------------


def full_query_layer(x: torch.Tensor, K: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [B,E,h,w], queries K [B,Q,E] -> energy maps y [B,Q,h,w], summaries [B,Q,E].
    softmax runs over the N = h·w pixels (networks/layers.py:18)."""
    n, c, h, w = x.shape
    _, q, ck = K.shape
    assert c == ck
    xt = x.view(n, c, h * w).permute(0, 2, 1)                 # [B,N,E]
    y = torch.matmul(xt, K.permute(0, 2, 1))                  # [B,N,Q]   :17
    y_norm = torch.softmax(y, dim=1)                          # :18
    summary = torch.matmul(y_norm.permute(0, 2, 1), xt)       # [B,Q,E]   :19
    return y.permute(0, 2, 1).reshape(n, q, h, w), summary    # :20


# --------------------------------------------------------------------------------------
# networks — reference networks/*.py.  State-dict key names follow SURVEY App. C.
# --------------------------------------------------------------------------------------


class QueryTrDecoder(nn.Module):
    """reference networks/depth_decoder_QTR.py:7-74 (ff=1024) / lite_depth_decoder_QTR.py:7-72 (ff=512)."""

    def __init__(self, in_channels, embedding_dim=128, patch_size=16, num_heads=4, query_nums=100,
                 dim_out=256, norm="linear", min_val=0.001, max_val=10, dim_feedforward=1024,
                 dropout=0.1):
        super().__init__()
        self.norm = norm
        self.embedding_convPxP = nn.Conv2d(in_channels, embedding_dim, patch_size, patch_size, 0)
        self.positional_encodings = nn.Parameter(torch.rand(500, embedding_dim), requires_grad=True)
        layer = nn.TransformerEncoderLayer(embedding_dim, num_heads, dim_feedforward=dim_feedforward,
                                           dropout=dropout)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=4, enable_nested_tensor=False)
        self.conv3x3 = nn.Conv2d(in_channels, embedding_dim, 3, 1, 1)
        self.bins_regressor = nn.Sequential(nn.Linear(embedding_dim * query_nums, 16 * query_nums),
                                            nn.LeakyReLU(),
                                            nn.Linear(16 * query_nums, 16 * 16),
                                            nn.LeakyReLU(),
                                            nn.Linear(16 * 16, dim_out))
        self.convert_to_prob = nn.Sequential(nn.Conv2d(query_nums, dim_out, 1, 1, 0), nn.Softmax(dim=1))
        self.query_nums, self.min_val, self.max_val = query_nums, min_val, max_val

    def forward(self, x0):
        emb = self.embedding_convPxP(x0.clone()).flatten(2)                       # :37-38
        emb = emb + self.positional_encodings[:emb.shape[2], :].T.unsqueeze(0)   # :39
        tokens = self.transformer_encoder(emb.permute(2, 0, 1))                  # :40-41
        x0 = self.conv3x3(x0)                                                    # :43
        queries = tokens[:self.query_nums, ...].permute(1, 0, 2)                 # :44-45
        energy, summ = full_query_layer(x0, queries)                             # :47
        bs, Q, E = summ.shape
        y = self.bins_regressor(summ.reshape(bs, Q * E))                         # :49
        if self.norm == "linear":
            y = torch.relu(y) + 0.1                                              # :51-54
        elif self.norm == "softmax":
            return torch.softmax(y, dim=1), energy
        else:
            y = torch.sigmoid(y)
        y = y / y.sum(dim=1, keepdim=True)                                       # :59
        out = self.convert_to_prob(energy)                                       # :61
        widths = (self.max_val - self.min_val) * y                               # :62
        widths = F.pad(widths, (1, 0), mode="constant", value=self.min_val)      # :63
        edges = torch.cumsum(widths, dim=1)                                      # :64
        centers = 0.5 * (edges[:, :-1] + edges[:, 1:])                           # :66
        pred = torch.sum(out * centers.view(bs, -1, 1, 1), dim=1, keepdim=True)  # :70
        return {("disp", 0): pred}


class PoseCNN(nn.Module):
    """reference networks/pose_cnn.py:9-45."""

    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        spec = [(3 * num_input_frames, 16, 7), (16, 32, 5), (32, 64, 3), (64, 128, 3), (128, 256, 3),
                (256, 256, 3), (256, 256, 3)]
        convs = [nn.Conv2d(i, o, k, 2, k // 2) for i, o, k in spec]
        self.pose_conv = nn.Conv2d(256, 6 * (num_input_frames - 1), 1)   # registered before `net` (:25 vs :31)
        self.net = nn.ModuleList(convs)

    def forward(self, out):
        for conv in self.net:
            out = F.relu(conv(out))
        out = self.pose_conv(out).mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_input_frames - 1, 1, 6)
        return out[..., :3], out[..., 3:]


class UpSampleBN(nn.Module):
    """reference networks/resnet_encoder.py:103-117."""

    def __init__(self, skip_input, output_features):
        super().__init__()
        self._net = nn.Sequential(nn.Conv2d(skip_input, output_features, 3, 1, 1),
                                  nn.BatchNorm2d(output_features), nn.LeakyReLU(),
                                  nn.Conv2d(output_features, output_features, 3, 1, 1),
                                  nn.BatchNorm2d(output_features), nn.LeakyReLU())

    def forward(self, x, concat_with):
        up = F.interpolate(x, size=list(concat_with.shape[2:]), mode="bilinear", align_corners=True)
        return self._net(torch.cat([up, concat_with], dim=1))


class DecoderBN(nn.Module):
    """reference networks/resnet_encoder.py:120-147 (skips 1024/512/256/64) and
    lite_res_encoder.py:120-146 (skips 256/128/64/64).  conv2 is a 1×1 conv with padding=1 (:125)."""

    def __init__(self, num_features, num_classes, bottleneck_features, skips):
        super().__init__()
        f = int(num_features)
        self.conv2 = nn.Conv2d(bottleneck_features, f, 1, 1, 1)
        self.up1 = UpSampleBN(f // 1 + skips[0], f // 2)
        self.up2 = UpSampleBN(f // 2 + skips[1], f // 4)
        self.up3 = UpSampleBN(f // 4 + skips[2], f // 8)
        self.up4 = UpSampleBN(f // 8 + skips[3], f // 16)
        self.conv3 = nn.Conv2d(f // 16, num_classes, 3, 1, 1)

    def forward(self, feats):
        b0, b1, b2, b3, b4 = feats
        x = self.conv2(b4)
        x = self.up1(x, b3)
        x = self.up2(x, b2)
        x = self.up3(x, b1)
        x = self.up4(x, b0)
        return self.conv3(x)


# --------------------------------------------------------------------------------------
# EfficientNet-b5 encoder — reference networks/base_encoder.py:58-107.  The trunk itself is torch.hub
# 'rwightman/gen-efficientnet-pytorch' tf_efficientnet_b5_ap (third-party, absent here: PARITY UNPINNED for its
# arithmetic); restated from the public architecture: EfficientNet-B0 stage table x width 1.6 / depth 2.2, TF "SAME"
# padding, BatchNorm eps 1e-3, swish, squeeze-and-excite sized by a quarter of the block's input channels.
# What is pinned: the DecoderBN around it (golden G18 from the reference's own class) and Encoder.forward's tap order.
# --------------------------------------------------------------------------------------
_B5_STAGES = (("ds", 3, 1, 1, 24, 3), ("ir", 3, 2, 6, 40, 5), ("ir", 5, 2, 6, 64, 5), ("ir", 3, 2, 6, 128, 7),
              ("ir", 5, 1, 6, 176, 7), ("ir", 5, 2, 6, 304, 9), ("ir", 3, 1, 6, 512, 3))


def _same_pad(x, k, stride):
    """TensorFlow SAME padding: total = max((ceil(n/s) - 1) s + k - n, 0), leading = total // 2"""
    H, W = x.shape[2:]
    th = max((-(-H // stride) - 1) * stride + k - H, 0)
    tw = max((-(-W // stride) - 1) * stride + k - W, 0)
    return F.pad(x, (tw // 2, tw - tw // 2, th // 2, th - th // 2))


class _SE(nn.Module):
    def __init__(self, chs, reduced):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, reduced, 1)
        self.conv_expand = nn.Conv2d(reduced, chs, 1)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * torch.sigmoid(self.conv_expand(F.silu(self.conv_reduce(s))))


class _DSConv(nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.k, self.stride, self.res = k, stride, stride == 1 and cin == cout
        self.conv_dw = nn.Conv2d(cin, cin, k, stride, 0, groups=cin, bias=False)
        self.bn1 = nn.BatchNorm2d(cin, eps=1e-3)
        self.se = _SE(cin, max(1, int(cin * 0.25 + 0.5)))
        self.conv_pw = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout, eps=1e-3)

    def forward(self, x):
        y = F.silu(self.bn1(self.conv_dw(_same_pad(x, self.k, self.stride))))
        y = self.bn2(self.conv_pw(self.se(y)))
        return y + x if self.res else y


class _MBConv(nn.Module):
    def __init__(self, cin, cout, k, stride, expand):
        super().__init__()
        mid = cin * expand
        self.k, self.stride, self.res = k, stride, stride == 1 and cin == cout
        self.conv_pw = nn.Conv2d(cin, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid, eps=1e-3)
        self.conv_dw = nn.Conv2d(mid, mid, k, stride, 0, groups=mid, bias=False)
        self.bn2 = nn.BatchNorm2d(mid, eps=1e-3)
        self.se = _SE(mid, max(1, int(cin * 0.25 + 0.5)))
        self.conv_pwl = nn.Conv2d(mid, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout, eps=1e-3)

    def forward(self, x):
        y = F.silu(self.bn1(self.conv_pw(x)))
        y = F.silu(self.bn2(self.conv_dw(_same_pad(y, self.k, self.stride))))
        y = self.bn3(self.conv_pwl(self.se(y)))
        return y + x if self.res else y


class EfficientNetB5(nn.Module):
    """module names of gen-efficientnet's GenEfficientNet (global_pool / classifier = Identity, base_encoder.py:99-100)"""

    def __init__(self, stages=_B5_STAGES, stem=48, head=2048):
        super().__init__()
        self.conv_stem = nn.Conv2d(3, stem, 3, 2, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(stem, eps=1e-3)
        blocks, cin = [], stem
        for kind, k, stride, expand, cout, repeats in stages:
            stage = []
            for i in range(repeats):
                st = stride if i == 0 else 1
                stage.append(_DSConv(cin, cout, k, st) if kind == "ds" else _MBConv(cin, cout, k, st, expand))
                cin = cout
            blocks.append(nn.Sequential(*stage))
        self.blocks = nn.Sequential(*blocks)
        if head is not None:
            self.conv_head = nn.Conv2d(cin, head, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(head, eps=1e-3)
            self.global_pool = nn.Identity()
            self.classifier = nn.Identity()

    def stride_features(self, x):
        """timm features_only (Unet.py:114-118 for --backbone tf_efficientnet_b5_ap): the last map of every stride = stages 0, 1, 2, 4, 6"""
        f = F.silu(self.bn1(self.conv_stem(_same_pad(x, 3, 2))))
        out = []
        for i, stage in enumerate(self.blocks):
            f = stage(f)
            if i in (0, 1, 2, 4, 6):
                out.append(f)
        return out

    def features(self, x):
        """reference Encoder.forward (base_encoder.py:63-73): one entry per top-level module, the stages of `blocks` one by one"""
        f = [x, self.conv_stem(_same_pad(x, 3, 2))]
        f.append(self.bn1(f[-1]))
        f.append(F.silu(f[-1]))
        for stage in self.blocks:
            f.append(stage(f[-1]))
        f.append(self.conv_head(f[-1]))
        f.append(self.bn2(f[-1]))
        f.append(F.silu(f[-1]))
        f += [f[-1], f[-1]]                                   # global_pool, classifier (Identity)
        return f


class _B5Encoder(nn.Module):
    def __init__(self, backend):
        super().__init__()
        self.original_model = backend

    def forward(self, x):
        return self.original_model.features(x)


class BaseEncoder(nn.Module):
    """reference networks/base_encoder.py:76-86"""

    def __init__(self, model_dim=32, num_features=2048, stages=_B5_STAGES):
        super().__init__()
        self.encoder = _B5Encoder(EfficientNetB5(stages))
        self.decoder = DecoderBN(num_features, model_dim, 2048, (176, 64, 40, 24))      # base_encoder.py:31-34

    def forward(self, x):
        f = self.encoder(x)
        return self.decoder((f[4], f[5], f[6], f[8], f[11]))                              # base_encoder.py:41


# ---------------------------------------------------------------------------------------------------
# ConvNeXt-L trunk (timm 'convnext_large', features_only — third-party, not in /root/reference: restated from the public
# definition, parity unpinned) and the reference's U-Net decoder (networks/Unet.py:211-312, pinned by golden G20)
class _CNBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(dim, 4 * dim), nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(1e-6 * torch.ones(dim))

    def forward(self, x):
        z = self.conv_dw(x).permute(0, 2, 3, 1)
        z = self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm(z)))).permute(0, 3, 1, 2)
        return x + z * self.gamma.reshape(1, -1, 1, 1)


def _ln2d(x, norm):
    return norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class _CNDown(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.LayerNorm(cin, eps=1e-6), nn.Conv2d(cin, cout, 2, 2))

    def forward(self, x):
        return self[1](_ln2d(x, self[0]))


class _CNStage(nn.Module):
    def __init__(self, cin, cout, depth, first):
        super().__init__()
        self.downsample = nn.Identity() if first else _CNDown(cin, cout)
        self.blocks = nn.Sequential(*[_CNBlock(cout) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class ConvNeXtFeatures(nn.Module):
    def __init__(self, in_chans=3, depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536)):
        super().__init__()
        self.stem_0, self.stem_1 = nn.Conv2d(in_chans, dims[0], 4, 4), nn.LayerNorm(dims[0], eps=1e-6)
        for i in range(4):
            setattr(self, "stages_%d" % i, _CNStage(dims[i - 1] if i else dims[0], dims[i], depths[i], i == 0))
        self.num_chs = list(dims)

    def forward(self, x):
        x = _ln2d(self.stem_0(x), self.stem_1)
        feats = []
        for i in range(4):
            x = getattr(self, "stages_%d" % i)(x)
            feats.append(x)
        return feats


class _UConvBnAct(nn.Module):
    """Unet.py:211-226"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv, self.bn = nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class _UDecoderBlock(nn.Module):
    """Unet.py:229-256"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1, self.conv2 = _UConvBnAct(cin, cout), _UConvBnAct(cout, cout)

    def forward(self, x, skip=None):
        if skip is not None:
            x = F.interpolate(x, size=[skip.size(2), skip.size(3)], mode="bilinear", align_corners=True)
            x = torch.cat([x, skip], dim=1)
        else:
            x = F.interpolate(x, scale_factor=2.0, mode="bilinear")
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    """Unet.py:258-312, center=False"""

    def __init__(self, encoder_channels, decoder_channels, final_channels):
        super().__init__()
        ins = [i + s for i, s in zip([encoder_channels[0]] + list(decoder_channels[:-1]), list(encoder_channels[1:]) + [0])]
        if len(ins) != len(decoder_channels):
            ins.append(ins[-1] // 2)
        self.blocks = nn.ModuleList([_UDecoderBlock(i, o) for i, o in zip(ins, decoder_channels)])
        self.final_conv = nn.Conv2d(decoder_channels[-1], final_channels, 1)

    def forward(self, feats):
        x, skips = feats[0], feats[1:]
        for i, b in enumerate(self.blocks):
            x = b(x, skips[i] if i < len(skips) else None)
        return self.final_conv(x)


class Unet(nn.Module):
    """Unet.py:82-148 with the convnext_large backbone"""

    def __init__(self, in_channels=3, num_classes=5, decoder_channels=(1024, 512, 256, 128), depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536)):
        super().__init__()
        self.encoder = ConvNeXtFeatures(in_channels, depths, dims)
        self.decoder = UnetDecoder(list(dims)[::-1], tuple(decoder_channels), num_classes)

    def forward(self, x):
        feats = self.encoder(x)
        feats.reverse()
        return self.decoder(feats)


class _B5Features(EfficientNetB5):
    def __init__(self, stages=_B5_STAGES):
        super().__init__(stages, head=None)

    def forward(self, x):
        return self.stride_features(x)


class UnetB5(nn.Module):
    """Unet.py:82-148 with the tf_efficientnet_b5_ap backbone (trainer.py:64 under args_files/hisfog/kitti/effb5_320x1024.txt)"""

    def __init__(self, num_classes=32, decoder_channels=(512, 256, 128, 64, 32), stages=_B5_STAGES):
        super().__init__()
        self.encoder = _B5Features(stages)
        chs = [stages[i][4] for i in (0, 1, 2, 4, 6)]
        self.decoder = UnetDecoder(chs[::-1], tuple(decoder_channels), num_classes)

    def forward(self, x):
        feats = self.encoder(x)
        feats.reverse()
        return self.decoder(feats)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # v1.5: stride on the 3×3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + idt)


class ResNetTrunk(nn.Module):
    """Public ResNet v1.5 (18/34/50) with torchvision's state-dict key names (conv1, bn1, layer{1..4}.
    {i}.{conv,bn,downsample.{0,1}}, fc).  PARITY UNPINNED: torchvision (0.9.1, reference
    requirements.txt:82) is not in /root/reference nor in this image."""

    def __init__(self, num_layers=50):
        super().__init__()
        block, counts = {18: (_BasicBlock, [2, 2, 2, 2]), 34: (_BasicBlock, [3, 4, 6, 3]),
                         50: (_Bottleneck, [3, 4, 6, 3])}[num_layers]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, counts[0], 1)
        self.layer2 = self._make(block, 128, counts[1], 2)
        self.layer3 = self._make(block, 256, counts[2], 2)
        self.layer4 = self._make(block, 512, counts[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, 1000)      # never used on the path (App. B-11)

    def _make(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                               nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    """reference networks/resnet_encoder.py:64-100: input normalisation + 5 taps."""

    def __init__(self, num_layers):
        super().__init__()
        self.encoder = ResNetTrunk(num_layers)

    def forward(self, img):
        e = self.encoder
        x = (img - 0.45) / 0.225                                         # :91
        f0 = e.relu(e.bn1(e.conv1(x)))                                   # :92-94
        f1 = e.layer1(e.maxpool(f0))                                     # :95
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]


class ResnetEncoderDecoder(nn.Module):
    """reference networks/resnet_encoder.py:150-157."""

    def __init__(self, num_layers=50, num_features=512, model_dim=32):
        super().__init__()
        self.encoder = ResnetEncoder(num_layers)
        self.decoder = DecoderBN(num_features, model_dim, 2048, (1024, 512, 256, 64))

    def forward(self, x):
        return self.decoder(self.encoder(x))


class LiteResnetEncoderDecoder(nn.Module):
    """reference networks/lite_res_encoder.py:148-157."""

    def __init__(self, model_dim=128):
        super().__init__()
        self.encoder = ResnetEncoder(18)
        self.decoder = DecoderBN(256, model_dim, 512, (256, 128, 64, 64))

    def forward(self, x):
        return self.decoder(self.encoder(x))


# --------------------------------------------------------------------------------------
# one optimisation step — reference trainer.py:236-246,266-299
# --------------------------------------------------------------------------------------


class RefTrainStep:
    """Holds the three networks + Adam and runs process_batch / backward / step exactly as
    reference trainer.py:240-244 does, with the tie-break noise passed in (trainer.py:516 draws it
    on the CPU RNG)."""

    def __init__(self, encoder, depth, pose, frame_ids=(0, -1, 1), H=192, W=640, lr=1e-4,
                 disparity_smoothness=1e-3, use_stereo=False, diff_lr=False, pose_model_input="pairs"):
        self.models = {"encoder": encoder, "depth": depth, "pose": pose}
        self.pose_model_input = pose_model_input
        self.frame_ids, self.H, self.W = list(frame_ids), H, W
        self.use_stereo = use_stereo                     # frame_ids then end with "s" (reference trainer.py:52-53)
        self.smooth_w = disparity_smoothness
        params = [p for m in self.models.values() for p in m.parameters()]
        if diff_lr:                                      # reference trainer.py:128-131: the pose network at lr / 10
            self.optim = torch.optim.Adam([{"params": list(pose.parameters()), "lr": lr / 10},
                                           {"params": list(encoder.parameters()) + list(depth.parameters()), "lr": lr}], lr)
        else:
            self.optim = torch.optim.Adam(params, lr)

    def predict_poses(self, inputs):
        """reference trainer.py:301-337 (pairs / posecnn)."""
        poses = {}
        if self.pose_model_input == "all":                                # :339-361: every temporal frame through ONE pass
            x = torch.cat([inputs[("color_aug", f, 0)] for f in self.frame_ids if f != "s"], 1)
            axisangle, translation = self.models["pose"](x)
            return {f: (axisangle, translation) for f in self.frame_ids[1:] if f != "s"}
        for f in self.frame_ids[1:]:
            if f == "s":                                                  # :317
                continue
            pair = [inputs[("color_aug", f, 0)], inputs[("color_aug", 0, 0)]] if f < 0 else \
                   [inputs[("color_aug", 0, 0)], inputs[("color_aug", f, 0)]]
            poses[f] = self.models["pose"](torch.cat(pair, 1))
        return poses

    def process_batch(self, inputs, noise):
        feats = self.models["encoder"](inputs[("color_aug", 0, 0)])      # :286
        outputs = self.models["depth"](feats)                            # :288
        poses = self.predict_poses(inputs)                               # :294
        colors = {f: inputs[("color", f, 0)] for f in self.frame_ids}
        cam_T_cam = None
        if self.pose_model_input == "all":                               # :355-359: pose i of the shared tensors, not inverted
            cam_T_cam = {f: transformation_from_parameters(poses[f][0][:, i], poses[f][1][:, i])
                         for i, f in enumerate(self.frame_ids[1:]) if f != "s"}
        chain = photometric_chain(outputs[("disp", 0)], poses, inputs[("K", 0)], inputs[("inv_K", 0)],
                                  colors, self.frame_ids, noise, self.H, self.W, self.smooth_w,
                                  inputs.get("stereo_T"), self.use_stereo, cam_T_cam)
        outputs.update(chain)
        for f in self.frame_ids[1:]:
            if f == "s":
                continue
            outputs[("axisangle", 0, f)], outputs[("translation", 0, f)] = poses[f]
            outputs[("cam_T_cam", 0, f)] = cam_T_cam[f] if cam_T_cam is not None else transformation_from_parameters(
                poses[f][0][:, 0], poses[f][1][:, 0], invert=(f < 0))   # :336-337
        return outputs, {"loss": chain["loss"], "loss/0": chain["loss/0"]}

    def step(self, inputs, noise):
        outputs, losses = self.process_batch(inputs, noise)
        self.optim.zero_grad()
        losses["loss"].backward()
        self.optim.step()
        return outputs, losses
