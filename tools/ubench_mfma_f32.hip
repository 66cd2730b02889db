// ubench_mfma_f32.hip — sustained fp32 matrix-pipe rate of the whole chip (dev tool): every SIMD of every CU issues independent
// v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 back to back (W waves per SIMD), nothing else.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f32.hip -o tools/bin/ubench_mfma_f32 && tools/bin/ubench_mfma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 4096;

template <int FORM>
__global__ __launch_bounds__(256) void k(float *out) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float s = 0.f;
    if (FORM == 0) {
        f4 c[8] = {};
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][3];
    } else {
        f16 c[4] = {};
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][15];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out;
    CHECK(hipMalloc(&out, 4096 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int form = 0; form < 2; ++form)
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
            for (int rep = 0; rep < 3; ++rep) {
                const int grid = 256 * wgs_per_cu;
                CHECK(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double flop = (form == 0 ? 8.0 * 2048 : 4.0 * 4096) * ITER * 4.0 * grid;      // per wave x 4 waves x workgroups
                printf("%s  %d wave(s)/SIMD  rep %d: %.3f ms  %.1f TFLOP/s\n", form == 0 ? "16x16x4" : "32x32x2", wgs_per_cu, rep, ms, flop / ms * 1e-9);
            }
    return 0;
}
