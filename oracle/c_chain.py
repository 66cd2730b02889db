"""ORACLE (test infrastructure): ctypes loader for oracle/warp_chain.c — never imported by the product."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsqd_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "warp_chain.c")):
        build()
    return ctypes.CDLL(_SO)


def _p(a, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def depth_up(disp, H, W):
    disp = np.ascontiguousarray(disp, np.float32)
    B, _, h, w = disp.shape
    out = np.empty((B, 1, H, W), np.float32)
    lib().sqo_depth_up(_p(disp), _p(out), B, h, w, H, W)
    return out


def warp(depth, inv_K, P, src):
    """depth [B,1,H,W], inv_K [B,4,4], P [B,3,4], src [B,C,H,W] -> grid, x0, y0, warped."""
    depth = np.ascontiguousarray(depth, np.float32); inv_K = np.ascontiguousarray(inv_K, np.float32)
    P = np.ascontiguousarray(P, np.float32); src = np.ascontiguousarray(src, np.float32)
    B, C, H, W = src.shape
    grid = np.empty((B, H, W, 2), np.float32)
    x0 = np.empty((B, H, W), np.int32); y0 = np.empty((B, H, W), np.int32)
    warped = np.empty((B, C, H, W), np.float32)
    lib().sqo_warp(_p(depth), _p(inv_K), _p(P), _p(src), _p(grid), _p(x0, ctypes.c_int32), _p(y0, ctypes.c_int32),
                   _p(warped), B, C, H, W)
    return grid, x0, y0, warped
