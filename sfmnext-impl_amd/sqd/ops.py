"""torch-facing wrappers over libsqd.so.  PyTorch supplies device memory and the current HIP stream;
all arithmetic of the photometric chain happens in the hand-written gfx950 kernels.

Every wrapper requires CUDA(HIP) fp32 contiguous tensors and raises otherwise — no CPU fallback."""
import ctypes

import torch

from . import lib as _l


def _ptr(t):
    """device address for a c_void_p argument (ctypes converts ints and None itself: no wrapper object per call — an eager
    training step makes ~5000 of these, and the host, not the device, bounds eager throughput)"""
    return t.data_ptr() if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current stream's hipStream_t (the raw-handle query is ~4x cheaper than building a torch.cuda.Stream object)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _req(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("sqd: tensors must live on the MI355X (HIP) device; the hot path has no CPU fallback")
        if not t.is_contiguous():
            raise RuntimeError("sqd: tensor must be contiguous")
        if t.dtype not in (torch.float32, torch.uint8, torch.int32):
            raise RuntimeError("sqd: unsupported dtype %s" % t.dtype)


def _sources_layout(sources):
    """0 for [B,3,H,W] frames in planar memory, lib.SOURCES_HWC for frames whose MEMORY is [B,H,W,3] (torch.channels_last tensors of the same
    shape: pack_pixels, or the static source frames of the captured step) — all frames of a call in the same layout."""
    hwc = [not f.is_contiguous() and f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last) for f in sources]
    for f, h in zip(sources, hwc):
        if not h:
            _req(f)
        elif not f.is_cuda or f.dtype != torch.float32:
            raise RuntimeError("sqd: source frames must be float32 tensors on the device")
    if any(hwc) and not all(hwc):
        raise RuntimeError("sqd: the source frames of a call must share a memory layout (planar or channels_last)")
    return _l.SOURCES_HWC if hwc and all(hwc) else 0


def _i32arr(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


# ---------------------------------------------------------------------------------------------------
def depth_up_fwd(disp_lr, H, W):
    """F.interpolate(disp,[H,W],bilinear,align_corners=False) + per-block (sum 1/d, sum d).
    Returns depth [B,1,H,W], part [B,nblk,2]."""
    _req(disp_lr)
    B, _, h, w = disp_lr.shape
    L = _l.lib()
    nblk = L.sqd_depth_up_nblk(H, W)
    depth = torch.empty(B, 1, H, W, device=disp_lr.device, dtype=torch.float32)
    part = torch.empty(B, nblk, 2, device=disp_lr.device, dtype=torch.float32)
    _l.check(L.sqd_depth_up_fwd(_ptr(disp_lr), _ptr(depth), _ptr(part), B, h, w, H, W, _stream()), "depth_up_fwd")
    return depth, part


def depth_up_bwd(g_depth, depth, g_mid, h, w):
    _req(g_depth, depth, g_mid)
    B, ng, H, W = g_depth.shape
    out = torch.empty(B, 1, h, w, device=depth.device, dtype=torch.float32)
    _l.check(_l.lib().sqd_depth_up_bwd(_ptr(g_depth), ng, _ptr(depth), _ptr(g_mid), _ptr(out), B, h, w, H, W, _stream()),
             "depth_up_bwd")
    return out


def pose_mats_fwd(axisangle, translation, invert, K, part=None, HW=0):
    """axisangle/translation [B,S,3] -> mid [B] (or None), T [B,S,4,4], P [B,S,3,4]."""
    _req(axisangle, translation, K, part)
    B, S, _ = axisangle.shape
    dev = axisangle.device
    T = torch.empty(B, S, 4, 4, device=dev, dtype=torch.float32)
    P = torch.empty(B, S, 3, 4, device=dev, dtype=torch.float32)
    mid = torch.empty(B, device=dev, dtype=torch.float32) if part is not None else None
    nblk = part.shape[1] if part is not None else 0
    _l.check(_l.lib().sqd_pose_mats_fwd(_ptr(axisangle), _ptr(translation), _i32arr(invert), _ptr(K), _ptr(part), nblk,
                                        HW, _ptr(mid), _ptr(T), _ptr(P), B, S, _stream()), "pose_mats_fwd")
    return mid, T, P


def pose_mats_bwd(axisangle, translation, invert, K, mid, g_P):
    _req(axisangle, translation, K, mid, g_P)
    B, S, _ = axisangle.shape
    g_aa = torch.empty_like(axisangle)
    g_tr = torch.empty_like(translation)
    g_mid = torch.empty(B, device=axisangle.device, dtype=torch.float32) if mid is not None else None
    _l.check(_l.lib().sqd_pose_mats_bwd(_ptr(axisangle), _ptr(translation), _i32arr(invert), _ptr(K), _ptr(mid),
                                        _ptr(g_P), _ptr(g_aa), _ptr(g_tr), _ptr(g_mid), B, S, _stream()), "pose_mats_bwd")
    return g_aa, g_tr, g_mid


def identity_fwd(target, sources, noise=None, rows_per_task=0, loss_flags=0):
    """Identity reprojection losses + 1e-5*noise -> [B,S,H,W]  (trainer.py:480-487,514-517).  Under LOSS_AVG_REPROJECTION the result is
    [B,1,H,W]: the mean over the sources + 1e-5 * noise [B,1,H,W] (trainer.py:489-490)."""
    _req(target, noise)
    loss_flags |= _sources_layout(sources)
    B, _, H, W = target.shape
    S = len(sources)
    NI = 1 if loss_flags & _l.LOSS_AVG_REPROJECTION else S
    if noise is not None and tuple(noise.shape) != (B, NI, H, W):
        raise ValueError("identity_fwd: noise must be [B,%d,H,W], got %s" % (NI, tuple(noise.shape)))
    out = torch.empty(B, NI, H, W, device=target.device, dtype=torch.float32)
    arr = (ctypes.c_void_p * S)(*[s.data_ptr() for s in sources])
    _l.check(_l.lib().sqd_identity_fwd_ex(_ptr(target), arr, _ptr(noise), _ptr(out), B, S, H, W, rows_per_task, loss_flags, _stream()),
             "identity_fwd")
    return out


# name of the kernel sqd_photo_fwd launches (the "warp+SSIM kernel" of the roofline record in bench.py)
PHOTO_FWD_KERNEL_NAME = "photo_tile_kernel<1> (fused warp + SSIM + L1 + min/auto-mask forward; training and inference launches are identical)"


PHOTO_FWD_EVENTS = None      # a list: photo_fwd brackets its launch with HIP events on the launch stream and appends the pair


def sources_hwc_ok(B, S, H, W, rows_per_task=0, loss_flags=0):
    """do the photometric kernels of a training step read channels_last source frames at this shape / these loss options?"""
    return bool(_l.lib().sqd_photo_sources_hwc_ok(B, S, H, W, rows_per_task, loss_flags))


def pack_pixels(frames, out=None):
    """Planar [B,3,H,W] frames -> the same frames as channels_last tensors (memory [B,H,W,3]; one launch for up to MAX_SOURCES frames):
    the layout identity_fwd / photo_fwd / photo_bwd read with fewer gathers (SQD_SOURCES_HWC; they recognise it by the strides)."""
    _req(*frames)
    B, _, H, W = frames[0].shape
    if out is None:
        out = [torch.empty(B, 3, H, W, device=f.device, dtype=torch.float32, memory_format=torch.channels_last) for f in frames]
    for o in out:
        if tuple(o.shape) != (B, 3, H, W) or not o.is_contiguous(memory_format=torch.channels_last) or o.dtype != torch.float32 or not o.is_cuda:
            raise ValueError("pack_pixels: outputs must be float32 channels_last device tensors of the frames' shape")
    n = len(frames)
    src = (ctypes.c_void_p * n)(*[f.data_ptr() for f in frames])
    dst = (ctypes.c_void_p * n)(*[o.data_ptr() for o in out])
    _l.check(_l.lib().sqd_pack_pixels(src, dst, n, B, H, W, _stream()), "pack_pixels")
    return out


def photo_fwd(depth, inv_K, P, target, sources, identity, training=True, want_taps=False, want_reproj=False,
              rows_per_task=0, prepared_only=False, loss_flags=0):
    """Fused warp + SSIM/L1 + min/auto-mask.  Returns a dict of device tensors.  `identity` may be None under LOSS_NO_AUTOMASK.
    Source frames in channels_last memory (pack_pixels) are read as such: fewer gathers, same bits."""
    _req(depth, inv_K, P, target)
    loss_flags |= _sources_layout(sources)
    if identity is not None:
        _req(identity)
    B, _, H, W = target.shape
    S = len(sources)
    dev = target.device
    L = _l.lib()
    nt = L.sqd_photo_ntasks(B, H, W, rows_per_task)
    f32 = dict(device=dev, dtype=torch.float32)
    out = {"sample": [torch.empty(B, H, W, 2, **f32) for _ in range(S)],
           "warped": [torch.empty(B, 3, H, W, **f32) for _ in range(S)],
           "sel": torch.empty(B, H, W, **f32),
           "idx": torch.empty(B, H, W, device=dev, dtype=torch.uint8),
           "loss_part": torch.empty(nt, **f32),
           "x0y0": [torch.empty(B, H, W, 2, device=dev, dtype=torch.int32) for _ in range(S)] if want_taps else None,
           "reproj": torch.empty(B, S, H, W, **f32) if want_reproj else None}
    a = _l.PhotoArgs()
    a.depth, a.inv_K, a.P, a.target = (t.data_ptr() for t in (depth, inv_K, P, target))
    a.identity = identity.data_ptr() if identity is not None else None
    a.loss_flags = loss_flags
    for s in range(S):
        a.sources[s] = sources[s].data_ptr()
        a.sample[s] = out["sample"][s].data_ptr()
        a.warped[s] = out["warped"][s].data_ptr()
        if want_taps:
            a.x0y0[s] = out["x0y0"][s].data_ptr()
    a.sel, a.idx, a.loss_part = out["sel"].data_ptr(), out["idx"].data_ptr(), out["loss_part"].data_ptr()
    a.coef = None
    a.reproj = out["reproj"].data_ptr() if want_reproj else None
    a.B, a.S, a.H, a.W, a.rows_per_task = B, S, H, W, rows_per_task
    a.stream = torch.cuda.current_stream().cuda_stream
    if prepared_only:
        return a, out
    if PHOTO_FWD_EVENTS is not None:          # bench.py: duration of this launch inside a (non-captured) training step
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        _l.check(L.sqd_photo_fwd(ctypes.byref(a)), "photo_fwd")
        ev[1].record()
        PHOTO_FWD_EVENTS.append(ev)
        return out
    _l.check(L.sqd_photo_fwd(ctypes.byref(a)), "photo_fwd")
    return out


def photo_fwd_relaunch(a):
    """Re-enqueue a prepared sqd_photo_fwd call (same buffers) — used by bench.py to time the kernel
    without allocator / Python overhead between launches."""
    _l.check(_l.lib().sqd_photo_fwd(ctypes.byref(a)), "photo_fwd")


def photo_coef(target, warped, idx, rows_per_task=0, loss_flags=0):
    """d(to_optimise)/d(window sums) of the winning source from the stored warped images -> coef [B,9,H,W]
    ([B,9 S,H,W] under LOSS_AVG_REPROJECTION: nine planes per source)."""
    _req(target, idx, *warped)
    B, _, H, W = target.shape
    S = len(warped)
    coef = torch.empty(B, 9 * S if loss_flags & _l.LOSS_AVG_REPROJECTION else 9, H, W, device=target.device, dtype=torch.float32)
    arr = (ctypes.c_void_p * S)(*[w.data_ptr() for w in warped])
    _l.check(_l.lib().sqd_photo_coef_ex(_ptr(target), arr, _ptr(idx), _ptr(coef), B, S, H, W, rows_per_task, loss_flags, _stream()),
             "photo_coef")
    return coef


def photo_bwd(depth, inv_K, P, target, sources, samples, warped, idx, gscale, rows_per_task=0, extra_planes=0, loss_flags=0):
    """-> g_depth [B,ceil(S/2)+extra_planes,H,W] (plane k = the pair of sources 2k, 2k+1; extra planes left unwritten), g_P [B,S,3,4]."""
    _req(depth, inv_K, P, target, idx, *samples, *warped)
    B, _, H, W = target.shape
    S = len(sources)
    dev = target.device
    L = _l.lib()
    coef = photo_coef(target, warped, idx, rows_per_task, loss_flags)
    loss_flags |= _sources_layout(sources)
    nt = L.sqd_photo_bwd_ntasks(B, S, H, W, rows_per_task)
    g_depth = torch.empty(B, (S + 1) // 2 + extra_planes, H, W, device=dev, dtype=torch.float32)
    part = torch.empty(nt, 12, device=dev, dtype=torch.float32)
    a = _l.PhotoBwdArgs()
    a.depth, a.inv_K, a.P, a.target, a.coef = (t.data_ptr() for t in (depth, inv_K, P, target, coef))
    for s in range(S):
        a.sources[s] = sources[s].data_ptr()
        a.sample[s] = samples[s].data_ptr()
    a.idx, a.g_depth, a.g_P_part = idx.data_ptr(), g_depth.data_ptr(), part.data_ptr()
    a.gscale = float(gscale)
    a.loss_flags = loss_flags
    a.g_depth_img_stride = ((S + 1) // 2 + extra_planes) * H * W
    a.B, a.S, a.H, a.W, a.rows_per_task = B, S, H, W, rows_per_task
    a.stream = torch.cuda.current_stream().cuda_stream
    _l.check(L.sqd_photo_bwd(ctypes.byref(a)), "photo_bwd")
    g_P = torch.empty(B, S, 3, 4, device=dev, dtype=torch.float32)
    _l.check(L.sqd_photo_bwd_reduce(_ptr(part), _ptr(g_P), nt, nt // (B * S), B, S, _stream()), "photo_bwd_reduce")
    return g_depth, g_P


def smooth_fwd(depth, color, part):
    _req(depth, color, part)
    B, _, H, W = depth.shape
    L = _l.lib()
    nb = L.sqd_smooth_nblk(H, W)
    sm_part = torch.empty(B, nb, 2, device=depth.device, dtype=torch.float32)
    _l.check(L.sqd_smooth_fwd(_ptr(depth), _ptr(color), _ptr(part), part.shape[1], _ptr(sm_part), B, H, W, _stream()),
             "smooth_fwd")
    return sm_part


def smooth_bwd(depth, color, part, sm_part, gout, planes=None, plane=0):
    """writes gout * d(smooth)/d(depth) into planes[:, plane] of a [B,n,H,W] buffer (allocated if None)."""
    _req(depth, color, part, sm_part, planes)
    B, _, H, W = depth.shape
    if planes is None:
        planes, plane = torch.empty(B, 1, H, W, device=depth.device, dtype=torch.float32), 0
    n = planes.shape[1]
    base = planes.data_ptr() + plane * H * W * 4
    _l.check(_l.lib().sqd_smooth_bwd(_ptr(depth), _ptr(color), _ptr(part), part.shape[1], _ptr(sm_part), float(gout),
                                     ctypes.c_void_p(base), n * H * W, B, H, W, _stream()), "smooth_bwd")
    return planes


# ---- stand-alone entries behind the reference's layer classes ----------------------------------------------------------
# The reference's SSIM / BackprojectDepth / Project3D / get_smooth_loss / transformation_from_parameters are ordinary autograd modules
# (layers.py:13-46,75-92,186-258,267-280).  Here the training path differentiates the fused chain (PhotometricChain); the stand-alone
# forms are autograd nodes over adjoint kernels of their own (sqd_ssim_bwd, sqd_backproject_bwd, sqd_project3d_bwd, sqd_smooth_bwd,
# sqd_pose_mats_bwd) w.r.t. the arguments the reference's training graph differentiates — images, depth, points, T, disparity, pose
# vectors.  An argument that is DATA there (intrinsics, the smoothness term's image) or an entry without an adjoint (grid_sample_border)
# refuses a tensor that requires a gradient — nothing detaches silently.
def _forward_only(who, *tensors):
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise RuntimeError("%s: no gradient is computed w.r.t. this argument (a forward-only stand-alone entry of libsqd, or an argument "
                           "that is data in the reference's training graph), and it requires one.  Differentiate through the fused "
                           "chain (Trainer.generate_images_pred / compute_losses) or pass a detached tensor." % who)


class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        depth, inv_K = depth.detach().contiguous().float(), inv_K.detach().contiguous().float()
        _req(depth, inv_K)
        B, _, H, W = depth.shape
        pts = torch.empty(B, 4, H * W, device=depth.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_backproject_fwd(_ptr(depth), _ptr(inv_K), _ptr(pts), B, H, W, _stream()), "backproject_fwd")
        ctx.save_for_backward(inv_K)
        ctx.shape = (B, H, W)
        return pts

    @staticmethod
    def backward(ctx, g_pts):
        (inv_K,) = ctx.saved_tensors
        B, H, W = ctx.shape
        g_pts = g_pts.contiguous().float()
        g_depth = torch.empty(B, 1, H, W, device=g_pts.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_backproject_bwd(_ptr(g_pts), _ptr(inv_K), _ptr(g_depth), B, H, W, _stream()), "backproject_bwd")
        return g_depth, None


def backproject(depth, inv_K):
    _forward_only("BackprojectDepth (inv_K)", inv_K)
    return _Backproject.apply(depth, inv_K)


class _Project3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, K, T, H, W, eps):
        points, K, T = (t.detach().contiguous().float() for t in (points, K, T))
        _req(points, K, T)
        B = points.shape[0]
        grid = torch.empty(B, H, W, 2, device=points.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_project3d_fwd(_ptr(points), _ptr(K), _ptr(T), _ptr(grid), B, H, W, float(eps), _stream()), "project3d_fwd")
        ctx.save_for_backward(points, K, T)
        ctx.geom = (B, H, W, float(eps))
        return grid

    @staticmethod
    def backward(ctx, g_grid):
        points, K, T = ctx.saved_tensors
        B, H, W, eps = ctx.geom
        L = _l.lib()
        g_grid = g_grid.contiguous().float()
        g_pts = torch.empty_like(points)
        part = torch.empty(B * L.sqd_project3d_bwd_nblk(H, W) * 12, device=points.device, dtype=torch.float32)
        g_T = torch.empty(B, 4, 4, device=points.device, dtype=torch.float32)
        _l.check(L.sqd_project3d_bwd(_ptr(points), _ptr(K), _ptr(T), _ptr(g_grid), _ptr(g_pts), _ptr(part), _ptr(g_T), B, H, W, eps, _stream()),
                 "project3d_bwd")
        return g_pts, None, g_T, None, None, None


def project3d(points, K, T, H, W, eps=1e-7):
    _forward_only("Project3D (K)", K)
    return _Project3D.apply(points, K, T, H, W, eps)


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = x.detach().contiguous().float(), y.detach().contiguous().float()
        _req(x, y)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        _l.check(_l.lib().sqd_ssim_fwd(_ptr(x), _ptr(y), _ptr(out), B * C, H, W, _stream()), "ssim_fwd")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        g = g.contiguous().float()
        ws = torch.empty(B * C * 4, H, W, device=x.device, dtype=torch.float32)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        _l.check(_l.lib().sqd_ssim_bwd(_ptr(x), _ptr(y), _ptr(g), _ptr(ws), _ptr(gx), _ptr(gy), B * C, H, W, _stream()), "ssim_bwd")
        return gx, gy


def ssim_map(x, y):
    return _Ssim.apply(x, y)


def grid_sample_border(img, grid, want_taps=False):
    """F.grid_sample(img, grid, padding_mode="border", align_corners=True) (reference trainer.py:431-435), forward only."""
    _forward_only("grid_sample_border", img, grid)
    img, grid = img.detach().contiguous().float(), grid.detach().contiguous().float()
    _req(img, grid)
    B, C, H, W = img.shape
    _, Ho, Wo, _ = grid.shape
    out = torch.empty(B, C, Ho, Wo, device=img.device, dtype=torch.float32)
    taps = torch.empty(B, Ho, Wo, 2, device=img.device, dtype=torch.int32) if want_taps else None
    _l.check(_l.lib().sqd_grid_sample_border_fwd(_ptr(img), _ptr(grid), _ptr(out), _ptr(taps), B, C, H, W, Ho, Wo, _stream()),
             "grid_sample_border_fwd")
    return (out, taps) if want_taps else out


class _SmoothLossPlain(torch.autograd.Function):
    """get_smooth_loss (reference layers.py:267-280) on an already normalised disparity: sqd_smooth_fwd / sqd_smooth_bwd with part == NULL."""

    @staticmethod
    def forward(ctx, disp, img):
        disp, img = disp.detach().contiguous().float(), img.detach().contiguous().float()
        _req(disp, img)
        B, _, H, W = disp.shape
        L = _l.lib()
        nb = L.sqd_smooth_nblk(H, W)
        sm = torch.empty(B, nb, 2, device=disp.device, dtype=torch.float32)
        _l.check(L.sqd_smooth_fwd(_ptr(disp), _ptr(img), ctypes.c_void_p(0), 0, _ptr(sm), B, H, W, _stream()), "smooth_fwd")
        ctx.save_for_backward(disp, img)
        return sm[..., 0].sum() / float(B * H * (W - 1)) + sm[..., 1].sum() / float(B * (H - 1) * W)

    @staticmethod
    def backward(ctx, gout):
        disp, img = ctx.saved_tensors
        B, _, H, W = disp.shape
        g = torch.empty(B, 1, H, W, device=disp.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_smooth_bwd(_ptr(disp), _ptr(img), ctypes.c_void_p(0), 0, ctypes.c_void_p(0), 1.0, _ptr(g), H * W, B, H, W,
                                         _stream()), "smooth_bwd")
        return g * gout, None            # (the adjoint is linear in the upstream gradient: no host synchronisation)


def smooth_loss_plain(disp, img):
    if torch.is_grad_enabled() and img.requires_grad:
        raise RuntimeError("get_smooth_loss: the image argument is data here — no gradient is computed w.r.t. it (detach it)")
    return _SmoothLossPlain.apply(disp, img)


class _PoseMatrix(torch.autograd.Function):
    """transformation_from_parameters (reference layers.py:75-92): sqd_pose_mats_fwd / sqd_pose_mats_bwd with K = I, no translation scale."""

    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        B = axisangle.shape[0]
        aa = axisangle.detach().reshape(B, 1, 3).contiguous().float()
        tr = translation.detach().reshape(B, 1, 3).contiguous().float()
        K = torch.eye(4, device=aa.device).repeat(B, 1, 1).contiguous()
        _, T, _ = pose_mats_fwd(aa, tr, [1 if invert else 0], K)
        ctx.save_for_backward(aa, tr, K)
        ctx.invert, ctx.shapes = bool(invert), (axisangle.shape, translation.shape)
        return T[:, 0]

    @staticmethod
    def backward(ctx, g_T):
        aa, tr, K = ctx.saved_tensors
        g_P = g_T[:, :3, :].reshape(-1, 1, 3, 4).contiguous().float()          # P = (K T)[:3] with K = I: the last row of T is constant
        g_aa, g_tr, _ = pose_mats_bwd(aa, tr, [1 if ctx.invert else 0], K, None, g_P)
        return g_aa.reshape(ctx.shapes[0]), g_tr.reshape(ctx.shapes[1]), None


def pose_matrix(axisangle, translation, invert=False):
    return _PoseMatrix.apply(axisangle, translation, invert)


# ---------------------------------------------------------------------------------------------------
# set by Trainer._backward for the duration of loss.backward() when the loss IS the chain's total (one scale): the upstream gradient is the
# unit seed, and the three element-wise products by it are skipped
UNIT_UPSTREAM = False


class PhotometricChain(torch.autograd.Function):
    """generate_images_pred + compute_losses of the reference (trainer.py:386-549) as one autograd node.

    forward(disp_lr [B,1,h,w], axisangle [B,Sp,3], translation [B,Sp,3], K, inv_K, target, identity [B,S,H,W], meta, *sources)
      -> total loss (differentiable), photo mean, smooth, depth, sel, T, sample_0.., warped_0.. (non-differentiable)
    `meta` = dict(H, W, invert=[...] for the Sp pose-net sources, smooth_weight, rows_per_task, use_stereo, stereo_T, loss_flags);
    loss_flags = lib.loss_flags(no_ssim, avg_reprojection, disable_automasking) (`identity` is then built with the same flags; it
    is ignored under disable_automasking);
    with stereo_T the last of the S sources is the other stereo camera (Sp = S - 1)."""

    @staticmethod
    def forward(ctx, disp_lr, axisangle, translation, K, inv_K, target, identity, meta, *sources):
        ctx.set_materialize_grads(False)         # 9 of the 10+ outputs never carry a gradient: no zero tensors for them
        H, W = meta["H"], meta["W"]
        B, S = target.shape[0], len(sources)
        rows = meta.get("rows_per_task", 0)
        stereo_T = meta.get("stereo_T")           # [B,4,4]: the last source is the other camera of the stereo pair (frame id "s")
        n_pose = S - (1 if stereo_T is not None else 0)
        assert axisangle.shape[1] == n_pose and len(meta["invert"]) == n_pose
        training = any(ctx.needs_input_grad[:3])
        depth, part = depth_up_fwd(disp_lr, H, W)
        # translation * mean inverse depth (reference trainer.py:412-421) only for the posecnn + mono configuration
        scaled = not meta.get("use_stereo", False)
        mid = T = P = None
        if n_pose:
            mid, T, P = pose_mats_fwd(axisangle, translation, meta["invert"], K, part if scaled else None, H * W)
        if stereo_T is not None:                  # T = inputs["stereo_T"] (trainer.py:406-407), P = (K @ T)[:, :3] (layers.py:248)
            Ts = stereo_T.detach().float().reshape(B, 1, 4, 4)
            Ps = torch.matmul(K, Ts[:, 0])[:, None, :3, :]
            T = Ts if T is None else torch.cat([T, Ts], 1)
            P = (Ps if P is None else torch.cat([P, Ps], 1)).contiguous()
        out = photo_fwd(depth, inv_K, P, target, list(sources), identity, training=training, rows_per_task=rows,
                        loss_flags=meta.get("loss_flags", 0))
        sm_part = smooth_fwd(depth, target, part)
        scal = torch.empty(3, device=depth.device, dtype=torch.float32)           # total, photometric mean, smoothness: one launch
        _l.check(_l.lib().sqd_chain_loss(_ptr(out["loss_part"]), out["loss_part"].numel(), _ptr(sm_part), sm_part.numel() // 2,
                                         1.0 / float(B * H * W), 1.0 / float(B * H * (W - 1)), 1.0 / float(B * (H - 1) * W),
                                         float(meta["smooth_weight"]), _ptr(scal), _stream()), "chain_loss")
        total, photo, smooth = scal[0], scal[1], scal[2]
        ctx.meta, ctx.S, ctx.n_pose, ctx.has_mid = meta, S, n_pose, mid is not None
        ctx.save_for_backward(disp_lr, axisangle, translation, K, inv_K, target, depth, part, mid if mid is not None else depth, P,
                              out["idx"], sm_part, *sources, *out["sample"], *out["warped"])
        outs = (total, photo, smooth, depth, out["sel"], T, *out["sample"], *out["warped"])
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_total, *_unused):
        if g_total is None:
            return (None,) * (8 + ctx.S)
        meta, S, n_pose = ctx.meta, ctx.S, ctx.n_pose
        saved = ctx.saved_tensors
        disp_lr, axisangle, translation, K, inv_K, target, depth, part, mid, P, idx, sm_part = saved[:12]
        if not ctx.has_mid:
            mid = None
        sources, samples, warped = list(saved[12:12 + S]), list(saved[12 + S:12 + 2 * S]), list(saved[12 + 2 * S:12 + 3 * S])
        B, _, H, W = target.shape
        h, w = disp_lr.shape[2:]
        rows = meta.get("rows_per_task", 0)
        # all adjoints are linear in the upstream gradient: run them with 1.0 and scale the three small results
        planes, g_P = photo_bwd(depth, inv_K, P, target, sources, samples, warped, idx, 1.0 / float(B * H * W), rows,
                                extra_planes=1, loss_flags=meta.get("loss_flags", 0))
        smooth_bwd(depth, target, part, sm_part, meta["smooth_weight"], planes, (S + 1) // 2)
        g_aa = g_tr = g_mid = None
        if n_pose:                                # (the stereo source's extrinsics are data: its rows of g_P are dropped)
            g_aa, g_tr, g_mid = pose_mats_bwd(axisangle, translation, meta["invert"], K, mid,
                                              g_P if n_pose == S else g_P[:, :n_pose].contiguous())
        g_disp = depth_up_bwd(planes, depth, g_mid, h, w)
        if UNIT_UPSTREAM:                         # (the Trainer differentiates the chain's total itself: the seed is exactly 1 — no scaling launches)
            return (g_disp, g_aa if n_pose else None, g_tr if n_pose else None, None, None, None, None, None) + (None,) * S
        g = g_total
        return (g_disp * g, g_aa * g if n_pose else None, g_tr * g if n_pose else None, None, None, None, None, None) + (None,) * S


# ---------------------------------------------------------------------------------------------------
def _sql_workspace(B, Q, E, N):
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    _l.check(_l.lib().sqd_sql_workspace(B, Q, E, N, ctypes.byref(a), ctypes.byref(b)), "sql_workspace")
    return a.value, b.value


class SelfQueryLayer(torch.autograd.Function):
    """FullQueryLayer.forward of the reference (networks/layers.py:7-21) as one fp32-MFMA kernel pair.
    forward(x [B,E,h,w], queries [B,Q,E]) -> (energy maps [B,Q,h,w], summaries [B,Q,E])."""

    @staticmethod
    def forward(ctx, x, queries):
        # a channels-last feature map (what the producing convolution writes) is read in place; anything else as planar NCHW
        nhwc = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
        if not nhwc:
            x = x.contiguous()
        queries = queries.contiguous()
        _req(x.permute(0, 2, 3, 1) if nhwc else x, queries)       # (the channels-last memory viewed as the dense [B,h,w,E] it is)
        B, E, h, w = x.shape
        Q, N = queries.shape[1], h * w
        dev = x.device
        nf, _ = _sql_workspace(B, Q, E, N)
        y = torch.empty(B, Q, h, w, device=dev, dtype=torch.float32)
        summary = torch.empty(B, Q, E, device=dev, dtype=torch.float32)
        lse = torch.empty(B, Q, 2, device=dev, dtype=torch.float32)
        part = torch.empty(nf, device=dev, dtype=torch.float32)
        _l.check(_l.lib().sqd_sql_fwd(_ptr(x), _ptr(queries), _ptr(y), _ptr(summary), _ptr(lse), _ptr(part), B, Q, E, N, int(nhwc),
                                      _stream()), "sql_fwd")
        ctx.save_for_backward(x, queries, y, summary, lse)
        ctx.nhwc = nhwc
        return y, summary

    @staticmethod
    def backward(ctx, g_y, g_summary):
        x, queries, y, summary, lse = ctx.saved_tensors
        B, E, h, w = x.shape
        Q, N = queries.shape[1], h * w
        dev = x.device
        _, nk = _sql_workspace(B, Q, E, N)
        g_y = g_y.contiguous() if g_y is not None else None
        g_summary = g_summary.contiguous() if g_summary is not None else torch.zeros_like(summary)
        g_x = torch.empty_like(x)                          # x's layout (preserve_format)
        g_K = torch.empty_like(queries)
        part = torch.empty(nk, device=dev, dtype=torch.float32)
        # (g_x is the output gradient of the convolution that produced the features: the kernel records its max |.| for that node's two-term
        #  fp16 operands — a 47 MB pass of its own at configs[1] otherwise)
        from . import nnkernels
        agx = nnkernels._amax_out(dev)
        _l.check(_l.lib().sqd_sql_bwd_amax(_ptr(x), _ptr(queries), _ptr(y), _ptr(g_y), _ptr(g_summary), _ptr(summary), _ptr(lse),
                                           _ptr(g_x), _ptr(g_K), _ptr(part), B, Q, E, N, int(ctx.nhwc), _ptr(agx), _stream()), "sql_bwd")
        nnkernels._amax_tag(g_x, agx)
        return g_x, g_K


class BinsHead(torch.autograd.Function):
    """Adaptive-bins head of Depth_Decoder_QueryTr (reference networks/depth_decoder_QTR.py:61-70) as one kernel pair:
    forward(energy [B,Q,h,w], weight [D,Q,1,1], bias [D], centers [B,D]) -> pred [B,1,h,w]
        = sum_d softmax_d(conv1x1(energy))[d] * centers[:, d]."""

    @staticmethod
    def forward(ctx, energy, weight, bias, centers):
        B, Q, h, w = energy.shape
        D = weight.shape[0]
        energy, centers, bias = energy.contiguous(), centers.contiguous(), bias.contiguous()
        wmat = weight.reshape(D, Q).contiguous()
        _req(energy, wmat, bias, centers)
        pred = torch.empty(B, 1, h, w, device=energy.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_bins_fwd(_ptr(energy), _ptr(wmat), _ptr(bias), _ptr(centers), _ptr(pred), B, Q, D, h * w, _stream()),
                 "bins_fwd")
        ctx.save_for_backward(energy, wmat, bias, centers)
        ctx.wshape = weight.shape
        return pred

    @staticmethod
    def backward(ctx, g_pred):
        energy, wmat, bias, centers = ctx.saved_tensors
        B, Q, h, w = energy.shape
        D = wmat.shape[0]
        g_pred = g_pred.contiguous()
        n = ctypes.c_int64(0)
        _l.check(_l.lib().sqd_bins_workspace(B, Q, D, h * w, ctypes.byref(n)), "bins_workspace")
        part = torch.empty(n.value, device=energy.device, dtype=torch.float32)
        g_e, g_w = torch.empty_like(energy), torch.empty_like(wmat)
        g_b, g_c = torch.empty_like(bias), torch.empty_like(centers)
        _l.check(_l.lib().sqd_bins_bwd(_ptr(energy), _ptr(wmat), _ptr(bias), _ptr(centers), _ptr(g_pred), _ptr(g_e), _ptr(g_w),
                                       _ptr(g_b), _ptr(g_c), _ptr(part), B, Q, D, h * w, _stream()), "bins_bwd")
        return g_e, g_w.view(ctx.wshape), g_b, g_c


class BinCenters(torch.autograd.Function):
    """Adaptive bin centres from the regressor's raw outputs (norm "linear": relu + 0.1, normalise, widths, cumsum, mid-points —
    reference networks/depth_decoder_QTR.py:56-66) in one launch each way.  forward(y [B,D], vmin, vmax) -> centers [B,D]."""

    @staticmethod
    def forward(ctx, y, vmin, vmax):
        y = y.contiguous()
        _req(y)
        B, D = y.shape
        centers = torch.empty_like(y)
        sums = torch.empty(B, device=y.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_bin_centers_fwd(_ptr(y), _ptr(centers), _ptr(sums), B, D, float(vmin), float(vmax), _stream()), "bin_centers_fwd")
        ctx.save_for_backward(y, sums)
        ctx.range = (float(vmin), float(vmax))
        return centers

    @staticmethod
    def backward(ctx, g):
        y, sums = ctx.saved_tensors
        B, D = y.shape
        g_y = torch.empty_like(y)
        _l.check(_l.lib().sqd_bin_centers_bwd(_ptr(y), _ptr(sums), _ptr(g.contiguous()), _ptr(g_y), B, D, ctx.range[0], ctx.range[1],
                                              _stream()), "bin_centers_bwd")
        return g_y, None, None


def bins_supported(Q, D):
    return 1 <= Q <= 128 and 1 <= D <= 128


def sql_supported(E, Q):
    return E in (16, 32, 48, 64) and 1 <= Q <= 128



# ---------------------------------------------------------------------------------------------------
# depth evaluation (reference evaluate_depth_config.py) — device tensors in, device tensors out
def disp_post_process(disp):
    """batch_post_process_disparity (evaluate_depth_config.py:50-59): disp [2N,h,w] fp32 — the outputs for N images followed by
    the outputs for their horizontally flipped copies, as the reference batches them (:133-137) — -> [N,h,w] fp64."""
    _req(disp)
    assert disp.dim() == 3 and disp.shape[0] % 2 == 0
    disp = disp.contiguous()
    N, h, w = disp.shape[0] // 2, disp.shape[1], disp.shape[2]
    out = torch.empty(N, h, w, device=disp.device, dtype=torch.float64)
    _l.check(_l.lib().sqd_disp_post_process(_ptr(disp), _ptr(out), N, h, w, _stream()), "disp_post_process")
    return out


EVAL_METRIC_NAMES = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")


def depth_eval(pred, gt, eval_split="eigen", min_depth=1e-3, max_depth=80.0, pred_depth_scale_factor=1.0, median_scaling=True):
    """the per-image body of the reference's evaluate() (evaluate_depth_config.py:225-261) on the device.
    pred [h,w] (fp32 or fp64 depth as the head predicts it), gt [Hg,Wg] fp32 -> fp64 tensor [9]: the seven metrics of
    compute_errors, the median-scaling ratio (NaN when disabled), the number of valid pixels."""
    if not (pred.is_cuda and gt.is_cuda):
        raise RuntimeError("sqd: depth_eval takes device tensors (no CPU fallback)")
    pred = pred.to(torch.float64).contiguous()
    gt = gt.to(torch.float32).contiguous()
    out = torch.empty(9, device=pred.device, dtype=torch.float64)
    _l.check(_l.lib().sqd_depth_eval(_ptr(pred), pred.shape[0], pred.shape[1], _ptr(gt), gt.shape[0], gt.shape[1],
                                     1 if eval_split == "eigen" else 0, float(min_depth), float(max_depth),
                                     float(pred_depth_scale_factor), 1 if median_scaling else 0, _ptr(out), _stream()), "depth_eval")
    return out


# ---------------------------------------------------------------------------------------------------
# supervised metric-depth finetune step (reference finetune/train_ft_SQLdepth.py:219-285)
class ResizeAlignCorners(torch.autograd.Function):
    """nn.functional.interpolate(pred, size, mode='bilinear', align_corners=True) for a [B,1,h,w] prediction"""

    @staticmethod
    def forward(ctx, x, H, W):
        _req(x)
        B, _, h, w = x.shape
        y = torch.empty(B, 1, H, W, device=x.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_resize_ac_fwd(_ptr(x), _ptr(y), B, h, w, H, W, _stream()), "resize_ac_fwd")
        ctx.dims = (B, h, w, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, h, w, H, W = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(B, 1, h, w, device=dy.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_resize_ac_bwd(_ptr(dy), None, _ptr(dx), B, h, w, H, W, _stream()), "resize_ac_bwd")
        return dx, None, None


METRIC_DEPTH_NAMES = ("a1", "a2", "a3", "abs_rel", "rmse", "log_10", "rmse_log", "silog", "sq_rel")     # reference finetune/utils.py:95-96
_CROPS = {None: 0, "garg": 1, "eigen": 2, "eigen_nyu": 3}


def metric_depth_eval(pred, depth, min_eval, max_eval, crop=None, median_scaling=True):
    """pred, depth [B,1,H,W] or [B,H,W] float32 (prediction at the ground truth's size) -> [B,11] float64 on the device: the nine metrics
    of METRIC_DEPTH_NAMES, the median-scaling ratio, the number of valid pixels — the per-image body of the reference's validate()
    (finetune/train_ft_SQLdepth.py:347-375) without the trip through numpy; median_scaling=False: that of evaluate_metric_depth.py's
    eval() (:65-141: the prediction unscaled and unclamped)."""
    if not pred.is_cuda:
        raise RuntimeError("sqd: metric_depth_eval needs tensors on the MI355X device — there is no CPU fallback")
    H, W = pred.shape[-2:]
    B = pred.numel() // (H * W)
    pred, depth = pred.detach().contiguous().float(), depth.contiguous().float()
    if depth.numel() != pred.numel():
        raise ValueError("metric_depth_eval: prediction and ground truth differ in size")
    out = torch.empty(B, 11, device=pred.device, dtype=torch.float64)
    _l.check(_l.lib().sqd_metric_depth_eval(_ptr(pred), _ptr(depth), _ptr(out), B, H, W, float(min_eval), float(max_eval), _CROPS[crop],
                                            1 if median_scaling else 0, _stream()), "metric_depth_eval")
    return out


def median_ratio(pred, depth, nscale, min_eval, max_eval, crop):
    """ratio [nscale] = median(depth[valid]) / median(pred[valid]) per sample (train_ft_SQLdepth.py:234-263); crop: None, 'garg' or 'eigen'."""
    B, _, H, W = depth.shape
    ratio = torch.ones(B, device=pred.device, dtype=torch.float32)
    if nscale > 0:
        code = {None: 0, "garg": 1, "eigen": 2}[crop]
        _l.check(_l.lib().sqd_median_ratio(_ptr(pred.detach().contiguous()), _ptr(depth.contiguous()), _ptr(ratio), nscale, H, W,
                                           float(min_eval), float(max_eval), code, _stream()), "median_ratio")
    return ratio


class SILog(torch.autograd.Function):
    """SILogLoss (reference finetune/loss.py:24-42, interpolate=False) of the ratio-scaled prediction over depth > min_depth.
    forward(pred [B,1,H,W], depth [B,1,H,W], scale [B] (constants), min_depth) -> scalar loss"""

    @staticmethod
    def forward(ctx, pred, depth, scale, min_depth):
        _req(pred, depth, scale)
        pred, depth = pred.contiguous(), depth.contiguous()
        B, _, H, W = pred.shape
        L = _l.lib()
        part = torch.empty(3 * L.sqd_silog_nblk(B * H * W), device=pred.device, dtype=torch.float64)
        stats = torch.empty(4, device=pred.device, dtype=torch.float32)
        _l.check(L.sqd_silog_fwd(_ptr(pred), _ptr(depth), _ptr(scale), _ptr(part), _ptr(stats), B, H * W, float(min_depth), _stream()), "silog_fwd")
        ctx.save_for_backward(pred, depth, scale, stats)
        ctx.min_depth = float(min_depth)
        return stats[3].clone()

    @staticmethod
    def backward(ctx, g):
        pred, depth, scale, stats = ctx.saved_tensors
        B, _, H, W = pred.shape
        g = g.contiguous().reshape(1)
        dpred = torch.empty_like(pred)
        _l.check(_l.lib().sqd_silog_bwd(_ptr(pred), _ptr(depth), _ptr(scale), _ptr(stats), _ptr(g), _ptr(dpred), B, H * W, ctx.min_depth,
                                        _stream()), "silog_bwd")
        return dpred, None, None, None
