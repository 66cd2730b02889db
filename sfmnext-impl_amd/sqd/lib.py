"""ctypes binding of libsqd.so (C ABI declared in include/sqd.h) + the hipcc build recipe.

The product path is HIP-only: `lib()` raises if the library is missing and every wrapper in
sqd.ops raises on CPU tensors.  There is no CPU fallback (the CPU restatement lives in oracle/ and is
test infrastructure)."""
import ctypes
import glob
import os
import shutil
import subprocess

# Load order matters: PyTorch-ROCm bundles its own libamdhip64.so; libsqd.so must bind to THAT runtime
# instance (the streams and device pointers it receives belong to it).  Importing torch first puts
# torch's runtime in the process so the loader resolves libsqd's libamdhip64.so.7 dependency to it;
# the other order yields two HIP runtimes and "no ROCm-capable device is detected" at the first launch.
import torch  # noqa: F401  (must precede ctypes.CDLL(libsqd.so))

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
SO_PATH = os.path.join(_HERE, "libsqd.so")
MAX_SOURCES = 4
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

c_float_p = ctypes.POINTER(ctypes.c_float)
c_void_p = ctypes.c_void_p
ABI_VERSION = 3   # SQD_ABI_VERSION of include/sqd.h this binding was written against
# SQD_LOSS_* (include/sqd.h): the reference's --no_ssim / --avg_reprojection / --disable_automasking
LOSS_NO_SSIM, LOSS_AVG_REPROJECTION, LOSS_NO_AUTOMASK = 1, 2, 4
SOURCES_HWC = 256       # SQD_SOURCES_HWC: the layout bit that travels with the loss options (source frames in [B,H,W,3] memory)


def loss_flags(no_ssim=False, avg_reprojection=False, disable_automasking=False):
    return (LOSS_NO_SSIM if no_ssim else 0) | (LOSS_AVG_REPROJECTION if avg_reprojection else 0) | (LOSS_NO_AUTOMASK if disable_automasking else 0)


class PhotoArgs(ctypes.Structure):
    """struct sqd_photo_args (include/sqd.h)."""
    _fields_ = [("depth", c_void_p), ("inv_K", c_void_p), ("P", c_void_p), ("target", c_void_p),
                ("sources", c_void_p * MAX_SOURCES), ("identity", c_void_p),
                ("sample", c_void_p * MAX_SOURCES), ("warped", c_void_p * MAX_SOURCES),
                ("sel", c_void_p), ("idx", c_void_p), ("x0y0", c_void_p * MAX_SOURCES),
                ("coef", c_void_p), ("reproj", c_void_p), ("loss_part", c_void_p),
                ("B", ctypes.c_int32), ("S", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("loss_flags", ctypes.c_int32), ("rows_per_task", ctypes.c_int32), ("stream", c_void_p)]


class PhotoBwdArgs(ctypes.Structure):
    """struct sqd_photo_bwd_args (include/sqd.h)."""
    _fields_ = [("depth", c_void_p), ("inv_K", c_void_p), ("P", c_void_p), ("target", c_void_p),
                ("coef", c_void_p), ("sources", c_void_p * MAX_SOURCES), ("sample", c_void_p * MAX_SOURCES),
                ("idx", c_void_p), ("g_depth", c_void_p), ("g_P_part", c_void_p), ("g_depth_img_stride", ctypes.c_int64),
                ("gscale", ctypes.c_float),
                ("B", ctypes.c_int32), ("S", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("loss_flags", ctypes.c_int32), ("rows_per_task", ctypes.c_int32), ("stream", c_void_p)]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


OBJ_DIR = os.path.join(CSRC, "_build")


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "sqd.h")]


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _stale(src, newest_header):
    o = _obj(src)
    return not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), newest_header)


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(f) > t for f in sources() + _headers())


def build(force=False, verbose=False):
    """hipcc cross-compiles every csrc/*.hip for gfx950 (works without a GPU): one object per source file, only the stale
    ones, in parallel; then one link into sqd/libsqd.so."""
    if not force and not needs_build():
        return SO_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest = max(os.path.getmtime(h) for h in _headers())
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + os.environ.get("SQD_HIPCC_EXTRA", "").split()      # (dev knob)
    todo = [s for s in sources() if force or _stale(s, newest)]

    def compile_one(src):
        cmd = [hipcc] + flags + ["-c", src, "-o", _obj(src) + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(_obj(src) + ".tmp", _obj(src))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    keep = {_obj(s) for s in sources()}
    for o in glob.glob(os.path.join(OBJ_DIR, "*.o")):       # objects of deleted sources must not be linked
        if o not in keep:
            os.remove(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + sorted(keep) + ["-o", SO_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(SO_PATH + ".tmp", SO_PATH)
    global _LIB
    _LIB = None
    return SO_PATH


_LIB = None
_I, _F, _P = ctypes.c_int, ctypes.c_float, c_void_p
_SIGNATURES = {
    "sqd_abi_version": (ctypes.c_int, []),
    "sqd_last_error": (ctypes.c_char_p, []),
    "sqd_comm_load": (_I, [ctypes.c_char_p]),
    "sqd_comm_unique_id": (_I, [ctypes.c_char_p]),
    "sqd_comm_init": (_I, [ctypes.c_char_p, _I, _I, ctypes.POINTER(c_void_p)]),
    "sqd_comm_rank": (_I, [_P]),
    "sqd_comm_world": (_I, [_P]),
    "sqd_comm_rccl_version": (_I, [ctypes.POINTER(ctypes.c_int)]),
    "sqd_comm_joined": (_I, [_P, ctypes.POINTER(ctypes.c_int)]),
    "sqd_comm_allreduce": (_I, [_P, _P, ctypes.c_int64, _I, _I, _P]),
    "sqd_comm_broadcast": (_I, [_P, _P, ctypes.c_int64, _I, _I, _P]),
    "sqd_comm_destroy": (_I, [_P]),
    "sqd_depth_up_nblk": (_I, [_I, _I]),
    "sqd_depth_up_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_depth_up_bwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_pose_mats_fwd": (_I, [_P, _P, ctypes.POINTER(ctypes.c_int32), _P, _P, _I, _I, _P, _P, _P, _I, _I, _P]),
    "sqd_pose_mats_bwd": (_I, [_P, _P, ctypes.POINTER(ctypes.c_int32), _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sqd_photo_ntasks": (_I, [_I, _I, _I, _I]),
    "sqd_photo_fwd": (_I, [ctypes.POINTER(PhotoArgs)]),
    "sqd_photo_set_fwd_variant": (_I, [_I]),
    "sqd_photo_sources_hwc_ok": (_I, [_I, _I, _I, _I, _I, _I]),
    "sqd_pack_pixels": (_I, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), _I, _I, _I, _I, _P]),
    "sqd_identity_fwd": (_I, [_P, ctypes.POINTER(c_void_p), _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_photo_coef": (_I, [_P, ctypes.POINTER(c_void_p), _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_identity_fwd_ex": (_I, [_P, ctypes.POINTER(c_void_p), _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sqd_photo_coef_ex": (_I, [_P, ctypes.POINTER(c_void_p), _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sqd_photo_bwd_ntasks": (_I, [_I, _I, _I, _I, _I]),
    "sqd_photo_bwd": (_I, [ctypes.POINTER(PhotoBwdArgs)]),
    "sqd_photo_bwd_reduce": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sqd_sql_workspace": (_I, [_I, _I, _I, _I, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "sqd_sql_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_sql_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_sql_bwd_amax": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "sqd_bn_nblk": (_I, [_I, _I]),
    "sqd_bn_train_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P]),
    "sqd_bn_eval_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "sqd_bn_train_fwd_pool": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P, _I, _P]),
    "sqd_bn_train_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sqd_dw_weight_layout": (_I, [_P, _P, _I, _I, _I, _P]),
    "sqd_dw_conv_fwd": (_I, [_P, _P, _P] + [_I] * 10 + [_P]),
    "sqd_dw_conv_dgrad": (_I, [_P, _P, _P] + [_I] * 10 + [_P]),
    "sqd_dw_conv_dgrad_add": (_I, [_P, _P, _P, _P] + [_I] * 10 + [_P]),
    "sqd_dw_conv_wgrad_chunks": (_I, [_I, _I, _I]),
    "sqd_dw_conv_wgrad": (_I, [_P, _P, _P] + [_I] * 10 + [_P]),
    "sqd_se_chunks": (_I, [_I]),
    "sqd_se_pool": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sqd_se_gate_fwd": (_I, [_P] * 8 + [_I, _I, _I, _I, _P]),
    "sqd_se_gate_bwd": (_I, [_P] * 11 + [_I, _I, _I, _I, _P]),
    "sqd_se_scale": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "sqd_upcat_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sqd_upcat_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sqd_upcat_bwd_bn_rows": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "sqd_upcat_bwd_bn": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "sqd_upcat_bwd_bn_amax": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "sqd_backproject_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sqd_project3d_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "sqd_ssim_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sqd_ssim_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "sqd_backproject_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sqd_project3d_bwd_nblk": (_I, [_I, _I]),
    "sqd_project3d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "sqd_grid_sample_border_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sqd_adam_chunk_elems": (_I, []),
    "sqd_adam_step": (_I, [_P, _P, _P, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P]),
    "sqd_adam_hyper": (_I, [ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P]),
    "sqd_adam_step_dev": (_I, [_P, _P, _P, _I, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P]),
    "sqd_adam_step_amax": (_I, [_P, _P, _P, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P, _P]),
    "sqd_adam_step_dev_amax": (_I, [_P, _P, _P, _I, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P, _P]),
    "sqd_conv_supported": (_I, [_I, _I]),
    "sqd_conv_set_plan": (_I, [_I] * 16),
    "sqd_conv_plan": (_I, [_I] * 12 + [ctypes.POINTER(ctypes.c_int64)]),
    "sqd_transpose2d": (_I, [_P, _P, _I, _I, _P, _P]),
    "sqd_conv_fwd_stats_rows": (_I, [_I] * 11),
    "sqd_conv_fwd": (_I, [_P, _P, _P, _P, _P, _P] + [_I] * 12 + [_P]),
    "sqd_conv_dgrad_stats_rows": (_I, [_I] * 11),
    "sqd_conv_dgrad_bn": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P] + [_I] * 11 + [_P]),
    "sqd_bn_train_bwd_pre": (_I, [_P] * 13 + [_I, _I, _I, _I, _P]),
    "sqd_bn_train_bwd_pre_red": (_I, [_P] * 13 + [_I, _I, _I, _I, _P, _P, ctypes.c_int64, _I, _P]),
    "sqd_conv_dgrad": (_I, [_P, _P, _P, _P, _P] + [_I] * 11 + [_P]),
    "sqd_resize_ac_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_resize_ac_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_median_ratio": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _I, _P]),
    "sqd_metric_depth_eval": (_I, [_P, _P, _P, _I, _I, _I, ctypes.c_float, ctypes.c_float, _I, _I, _P]),
    "sqd_silog_nblk": (_I, [ctypes.c_int64]),
    "sqd_silog_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "sqd_silog_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "sqd_grad_sumsq": (_I, [_P, _P, _P, _I, _P, _P]),
    "sqd_clip_coef": (_I, [_P, _I, ctypes.c_double, _P, _P]),
    "sqd_adamw_step": (_I, [_P, _P, _P, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P, _P]),
    "sqd_ln_rows_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "sqd_ln_rows_fwd_amax": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P]),
    "sqd_gelu_fwd_amax": (_I, [_P, _P, ctypes.c_int64, _P, _P]),
    "sqd_gelu_bwd_amax": (_I, [_P, _P, _P, ctypes.c_int64, _P, _P]),
    "sqd_scale_residual_bwd_amax": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "sqd_scale_residual_bwd_sums": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "sqd_gelu_bwd_rows": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "sqd_ln_rows_nblk": (_I, [_I]),
    "sqd_ln_rows_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sqd_gelu_fwd": (_I, [_P, _P, ctypes.c_int64, _P]),
    "sqd_gelu_bwd": (_I, [_P, _P, _P, ctypes.c_int64, _P]),
    "sqd_scale_residual_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "sqd_scale_residual_nblk": (_I, [_I]),
    "sqd_scale_residual_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "sqd_upsample2x_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sqd_upsample2x_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sqd_resample_h_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "sqd_resample_v_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_luma_sum_u8": (_I, [_P, _P, _I, _I, _P]),
    "sqd_color_jitter_step_u8": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sqd_u8_to_chw_f32": (_I, [_P, _P, _I, _I, _P]),
    "sqd_disp_post_process": (_I, [_P, _P, _I, _I, _I, _P]),
    "sqd_depth_eval": (_I, [_P, _I, _I, _P, _I, _I, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P, _P]),
    "sqd_pose_head_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "sqd_pose_head_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "sqd_act_bwd": (_I, [_P, _P, _P, ctypes.c_int64, _I, _P]),
    "sqd_conv_wgrad_plan": (_I, [_I] * 7 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)]),
    "sqd_conv_wgrad_set_plan": (_I, [_I] * 9),
    "sqd_conv_wgrad": (_I, [_P, _P, _P, _P, _P] + [_I] * 11 + [_P]),
    "sqd_conv_wgrad_partials": (_I, [_P, _P, _P, _P, _P] + [_I] * 11 + [ctypes.POINTER(ctypes.c_int), _P]),
    "sqd_bn_train_fwd_amax": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P, _I, _P, _P]),
    "sqd_bn_train_bwd_amax": (_I, [_P] * 13 + [_I, _I, _I, _I, _P, _P, ctypes.c_int64, _I, _P, _P, _P]),
    "sqd_bn_train_bwd_res": (_I, [_P] * 13 + [_I, _I, _I, _I, _P, _P, ctypes.c_int64, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sqd_bn_bwd_res_rows": (_I, [_I, _I, _I, _I]),
    "sqd_upcat_fwd_amax": (_I, [_P, _P, _P] + [_I] * 7 + [_P, _P]),
    "sqd_space_to_depth2_planar_amax": (_I, [_P, _P, _P] + [_I] * 6 + [ctypes.c_int64, ctypes.c_float, ctypes.c_float, _P, _P]),
    "sqd_conv_wgrad_effective_impl": (_I, [_I] * 11),
    "sqd_amax": (_I, [_P, ctypes.c_int64, _P, _P]),
    "sqd_amax_multi": (_I, [_P, _P, _I, _I, _P, _P]),
    "sqd_conv_fwd_scaled": (_I, [_P] * 9 + [_I] * 12 + [_P]),
    "sqd_conv_dgrad_scaled": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P] + [_I] * 11 + [_P]),
    "sqd_act_bwd_amax": (_I, [_P, _P, _P, ctypes.c_int64, _I, _P, _P]),
    "sqd_conv_wgrad_scaled": (_I, [_P] * 7 + [_I] * 11 + [ctypes.POINTER(ctypes.c_int), _P]),
    "sqd_split_reduce": (_I, [_P, _P, ctypes.c_int64, _I, _P]),
    "sqd_bins_supported": (_I, [_I, _I]),
    "sqd_bins_set_arith": (_I, [_I]),
    "sqd_bins_workspace": (_I, [_I, _I, _I, _I, ctypes.POINTER(ctypes.c_int64)]),
    "sqd_bins_fwd": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "sqd_bins_bwd": (_I, [_P] * 10 + [_I] * 4 + [_P]),
    "sqd_conv_set_precision": (_I, [_I]),
    "sqd_conv_precision": (_I, []),
    "sqd_vit_supported": (_I, [_I, _I]),
    "sqd_addln_fwd": (_I, [_P, _P, _I] + [_P] * 7 + [_I, _I, _F, _F, _P]),
    "sqd_addln_nblk": (_I, [_I]),
    "sqd_addln_bwd": (_I, [_P, _P, _I] + [_P] * 7 + [_I, _I, _F, _P]),
    "sqd_ffn_groups": (_I, [_I]),
    "sqd_ffn_tiles": (_I, [_I]),
    "sqd_ffn_fwd": (_I, [_P] * 6 + [_I, _I, _I, _F, _P]),
    "sqd_ffn_bwd": (_I, [_P] * 11 + [_I, _I, _I, _F, _P]),
    "sqd_colsum_multi": (_I, [_P] * 5 + [_I, _P]),
    "sqd_mha_supported": (_I, [_I, _I, _I]),
    "sqd_mha_fwd": (_I, [_P] * 8 + [_I, _I, _I, _I, _F, _P]),
    "sqd_mha_bwd": (_I, [_P] * 13 + [_I, _I, _I, _I, _F, _P]),
    "sqd_bin_centers_fwd": (_I, [_P, _P, _P, _I, _I, _F, _F, _P]),
    "sqd_bin_centers_bwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _F, _P]),
    "sqd_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "sqd_maxpool3x3s2_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sqd_maxpool3x3s2_bwd_bn_rows": (_I, [_I, _I, _I, _I]),
    "sqd_maxpool3x3s2_bwd_bn": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "sqd_space_to_depth2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_space_to_depth2_planar": (_I, [_P, _P, _P] + [_I] * 6 + [ctypes.c_int64, ctypes.c_float, ctypes.c_float, _P]),
    "sqd_tokens_pos_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sqd_tokens_pos_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "sqd_first_queries": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_sum_parts": (_I, [_P, _P, _P, _I, ctypes.c_int64, _P]),
    "sqd_stem_regroup": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sqd_stem_regroup_ex": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sqd_smooth_nblk": (_I, [_I, _I]),
    "sqd_smooth_fwd": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "sqd_smooth_bwd": (_I, [_P, _P, _P, _I, _P, _F, _P, ctypes.c_int64, _I, _I, _I, _P]),
    "sqd_chain_loss": (_I, [_P, _I, _P, _I, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P]),
}


def exported_symbols():
    """Every entry point include/sqd.h declares (used by the symbol-export test)."""
    return sorted(_SIGNATURES)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError("libsqd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). The SQLdepth hot path has no CPU fallback.")
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.sqd_abi_version() != ABI_VERSION:
            raise RuntimeError("libsqd.so ABI version mismatch")
        _LIB = L
    return _LIB


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libsqd %s failed (%d): %s" % (what, rc, lib().sqd_last_error().decode()))
