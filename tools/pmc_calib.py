"""Launch the FETCH_SIZE / WRITE_SIZE calibration kernels (tools/pmc_calib.hip) on a 1 GiB buffer (4x the Infinity Cache),
5 launches each.  Run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`; tools/pmc_traffic.py turns
the two passes plus the same passes over tools/bench_fused.py into the corrected HBM traffic of the fused kernel.
usage: python tools/pmc_calib.py        (builds tools/_pmc_calib.so with hipcc if it is missing)"""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_pmc_calib.so")


def build():
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "pmc_calib.hip"), "-o", SO])


if __name__ == "__main__":
    if not os.path.exists(SO):
        build()
    import torch
    L = ctypes.CDLL(SO)
    L.calib_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    BYTES = 1 << 30
    buf = torch.zeros(BYTES // 4, device="cuda")
    sink = torch.zeros(256, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for which in range(4):
        for _ in range(5):
            rc = L.calib_run(which, buf.data_ptr(), sink.data_ptr(), BYTES, st)
            assert rc == 0, rc
        torch.cuda.synchronize()
    print("calibration launches done: 4 kernels x 5 launches over %d bytes" % BYTES)
