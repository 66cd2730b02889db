// Calibration kernels for rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, "HBM": the counters are
// calibrated for wide coalesced reads only — "calibrate on a known byte count in your own access pattern").
// Known byte counts with the access widths the photometric kernels use: coalesced dword loads / stores, plus the
// 16 B/lane read the guide documents (expected to report 1/2).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void calib_read_dword(const float *__restrict__ x, float *__restrict__ sink, size_t n) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += x[i];
    if (a == 123.456f) sink[threadIdx.x] = a;            // never true for the calibration data; keeps the loads alive
}
__global__ __launch_bounds__(256) void calib_read_f4(const float4 *__restrict__ x, float *__restrict__ sink, size_t n4) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        a += (v.x + v.y) + (v.z + v.w);
    }
    if (a == 123.456f) sink[threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void calib_write_dword(float *__restrict__ y, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = v;
}
__global__ __launch_bounds__(256) void calib_write_f4(float4 *__restrict__ y, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) y[i] = make_float4(v, v, v, v);
}

extern "C" int calib_run(int which, void *buf, void *sink, uint64_t bytes, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(256 * 16), block(256);
    if (which == 0) hipLaunchKernelGGL(calib_read_dword, grid, block, 0, s, (const float *)buf, (float *)sink, (size_t)(bytes / 4));
    if (which == 1) hipLaunchKernelGGL(calib_read_f4, grid, block, 0, s, (const float4 *)buf, (float *)sink, (size_t)(bytes / 16));
    if (which == 2) hipLaunchKernelGGL(calib_write_dword, grid, block, 0, s, (float *)buf, (size_t)(bytes / 4), 1.0f);
    if (which == 3) hipLaunchKernelGGL(calib_write_f4, grid, block, 0, s, (float4 *)buf, (size_t)(bytes / 16), 1.0f);
    return (int)hipGetLastError();
}
