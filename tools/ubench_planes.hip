// ubench_planes.hip — what the load path delivers for the head kernels' access pattern (dev tool, round 5)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_planes.hip -o tools/bin/ubench_planes && tools/bin/ubench_planes
// The bins head and the Self Query Layer read / write energy maps [B][Q][N] (planar) one pixel TILE at a time: a wave's lane is a pixel, its
// k-slots are planes, so one wave-instruction touches 2 planes x 128 bytes and a tile 128 planes x 128 bytes.  bins_fwd_h_kernel (matrix pipe
// 20 % busy, two waves per SIMD) runs at 3.1 TB/s on a 335 MB tensor and no faster on an 84 MB one.  Is that the pattern's ceiling?  Variants:
//   W = 1 / 2 / 4 : 4 / 8 / 16 bytes per lane = tiles of 32 / 64 / 128 pixels (128- / 256- / 512-byte plane rows per half wave)
//   OCC           : waves per SIMD the launch allows (register budget of the real kernels: 2; a pure streaming kernel: 8)
//   stream        : the same bytes as one contiguous float4 stream (the chip's read rate)
// Every variant sums what it reads (64 loads in flight per lane and tile, as the real kernel) and writes one float per pixel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W, int OCC>
__global__ __launch_bounds__(256, OCC) void k_tiles(const float *__restrict__ E, float *__restrict__ out, int Q, int N, int tiles_per_image, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5, b = blockIdx.y;
    const float *Eb = E + (size_t)b * Q * N;
    for (int it = 0; it < iters; ++it) {
        const int tile = (it * gridDim.x + blockIdx.x) * 4 + wave;
        if (tile >= tiles_per_image) break;
        const int p = tile * 32 * W + i * W;
        float acc[W];
#pragma unroll
        for (int w = 0; w < W; ++w) acc[w] = 0.f;
        // plane 16 ks + 8 h + j, as bins_fwd_h_kernel; 64 / W loads of W dwords in flight per lane for Q = 128
#pragma unroll 1
        for (int ks0 = 0; ks0 < Q / 16; ks0 += 8 / W < 1 ? 1 : 8 / W * W / W) {
            float v[8][8][W];
            const int nks = (Q / 16 - ks0) < (8 / W) ? (Q / 16 - ks0) : (8 / W);
#pragma unroll
            for (int ks = 0; ks < 8 / W; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float *src = Eb + (size_t)(16 * (ks0 + ks) + 8 * h + j) * N + p;
                    if (ks < nks) {
                        if constexpr (W == 1) v[ks][j][0] = *src;
                        else if constexpr (W == 2) { const float2 t = *reinterpret_cast<const float2 *>(src); v[ks][j][0] = t.x; v[ks][j][1] = t.y; }
                        else { const float4 t = *reinterpret_cast<const float4 *>(src); v[ks][j][0] = t.x; v[ks][j][1] = t.y; v[ks][j][2] = t.z; v[ks][j][3] = t.w; }
                    } else {
#pragma unroll
                        for (int w = 0; w < W; ++w) v[ks][j][w] = 0.f;
                    }
                }
#pragma unroll
            for (int ks = 0; ks < 8 / W; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int w = 0; w < W; ++w) acc[w] += v[ks][j][w];
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const float s = acc[w] + __shfl_xor(acc[w], 32, 64);
            if (h == 0) out[(size_t)b * N + p + w] = s;
        }
    }
}

__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ E, float *__restrict__ out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 t = E[i];
        s += (t.x + t.y) + (t.z + t.w);
    }
    if (s == 12345.678f) out[0] = s;
}

template <typename F>
float time_us(F f, int n = 20) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < n; ++i) f();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / n;
}

template <int W, int OCC>
void run(const float *E, float *out, int B, int Q, int N, int wg_per_cu) {
    const int tiles = N / (32 * W), wg = (256 * wg_per_cu) / B, iters = (tiles + wg * 4 - 1) / (wg * 4);
    const float us = time_us([&] { hipLaunchKernelGGL((k_tiles<W, OCC>), dim3(wg, B), dim3(256), 0, 0, E, out, Q, N, tiles, iters); });
    printf("  %2d bytes/lane, launch bound %d waves/SIMD, %d workgroups/CU: %7.1f us = %.2f TB/s\n", 4 * W, OCC, wg_per_cu, us, (double)B * Q * N * 4 / us / 1e6);
}

int main() {
    const int Q = 128, N = 160 * 512;
    for (int B : {2, 8}) {
        float *E, *out;
        const size_t n = (size_t)B * Q * N;
        CHECK(hipMalloc(&E, n * 4)); CHECK(hipMalloc(&out, (size_t)B * N * 4));
        CHECK(hipMemset(E, 0, n * 4));
        printf("B = %d: %.0f MB%s\n", B, n * 4 / 1e6, n * 4 < 200e6 ? " (Infinity-Cache resident)" : "");
        const float us = time_us([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (const float4 *)E, out, n / 4); });
        printf("  contiguous float4 stream: %7.1f us = %.2f TB/s\n", us, n * 4 / us / 1e6);
        run<1, 2>(E, out, B, Q, N, 2); run<1, 4>(E, out, B, Q, N, 4); run<1, 8>(E, out, B, Q, N, 8);
        run<2, 2>(E, out, B, Q, N, 2); run<2, 4>(E, out, B, Q, N, 4);
        run<4, 2>(E, out, B, Q, N, 2); run<4, 4>(E, out, B, Q, N, 4);
        CHECK(hipFree(E)); CHECK(hipFree(out));
    }
    return 0;
}
