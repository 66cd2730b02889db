// ubench_mfma.hip — does the bf16 matrix pipe of a gfx950 SIMD run next to VALU / LDS work? (dev tool)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/bin/ubench_mfma && tools/bin/ubench_mfma
// Every form runs ITER x 16 "units" per wave; a unit is one v_mfma_f32_32x32x16_bf16 and/or NV v_fma_f32 and/or one ds_read_b128.
// Reported: ns per unit per SIMD at 1 and 2 waves per SIMD.  MIX forms: 8-wave blocks, waves 0-3 issue only MFMA, waves 4-7 only
// VALU (or LDS reads): co-execution shows as max(), serialisation as the sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr int ITER = 256;

#define MF(acc) "v_mfma_f32_32x32x16_bf16 %" #acc ", %4, %5, %" #acc "\n"
#define V8 "v_fma_f32 %6, %6, %14, %7\n v_fma_f32 %7, %7, %14, %8\n v_fma_f32 %8, %8, %14, %9\n v_fma_f32 %9, %9, %14, %10\n" \
           "v_fma_f32 %10, %10, %14, %11\n v_fma_f32 %11, %11, %14, %12\n v_fma_f32 %12, %12, %14, %13\n v_fma_f32 %13, %13, %14, %6\n"
#define DS "ds_read_b128 %15, %16\n"
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7), "v"(m), "v"(d), "v"(addr)
// (f0..f7 are read-write in the asm text but declared as inputs: their values are never used outside, only the issue matters)

template <int FORM>
__global__ __launch_bounds__(512) void k(float *out) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    u4v a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x}, d = {};
    float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    const float m = 0.999f;
    const unsigned addr = (threadIdx.x & 63) * 80 % 16000;
    const int wave = threadIdx.x >> 6;
    const int role = FORM < 10 ? FORM : (wave < 4 ? 0 : (FORM == 10 ? 2 : 5));
    for (int i = 0; i < ITER; ++i) {
        if (role == 0) {          // MFMA only, 4 independent accumulators
            asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) OPS);
        } else if (role == 1) {   // MFMA + 8 VALU each, same wave
            asm volatile(MF(0) V8 MF(1) V8 MF(2) V8 MF(3) V8 MF(0) V8 MF(1) V8 MF(2) V8 MF(3) V8 MF(0) V8 MF(1) V8 MF(2) V8 MF(3) V8 MF(0) V8 MF(1) V8 MF(2) V8 MF(3) V8 OPS);
        } else if (role == 2) {   // 8 VALU per unit only
            asm volatile(V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 V8 OPS);
        } else if (role == 3) {   // MFMA, one dependent chain
            asm volatile(MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) MF(0) OPS);
        } else if (role == 4) {   // MFMA + one ds_read_b128 each
            asm volatile(MF(0) DS MF(1) DS MF(2) DS MF(3) DS MF(0) DS MF(1) DS MF(2) DS MF(3) DS MF(0) DS MF(1) DS MF(2) DS MF(3) DS MF(0) DS MF(1) DS MF(2) DS MF(3) DS "s_waitcnt lgkmcnt(0)\n" OPS);
        } else if (role == 5) {   // ds_read_b128 only
            asm volatile(DS DS DS DS DS DS DS DS DS DS DS DS DS DS DS DS "s_waitcnt lgkmcnt(0)\n" OPS);
        } else if (role == 6) {   // MFMA + 8 VALU + ds_read
            asm volatile(MF(0) V8 DS MF(1) V8 DS MF(2) V8 DS MF(3) V8 DS MF(0) V8 DS MF(1) V8 DS MF(2) V8 DS MF(3) V8 DS MF(0) V8 DS MF(1) V8 DS MF(2) V8 DS MF(3) V8 DS MF(0) V8 DS MF(1) V8 DS MF(2) V8 DS MF(3) V8 DS "s_waitcnt lgkmcnt(0)\n" OPS);
        }
    }
    float r = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + __uint_as_float(d.x);
    if (r == 123.456f) out[0] = r;
}

template <int FORM>
void run(const char *name, float *out) {
    const int threads = FORM >= 10 ? 512 : 256;
    printf("%-52s", name);
    for (int w = 1; w <= 2; ++w) {
        const int blocks = 256 * w;
        hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(threads), 0, 0, out);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(threads), 0, 0, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // units per SIMD: w blocks per CU, each block puts one (FORM<10) wave of the role on every SIMD
        printf("  %d blk/CU: %7.2f ns/unit", w, ms / 5 * 1e6 / (ITER * 16.0 * w));
    }
    printf("\n");
}

int main() {
    float *out;
    CHECK(hipMalloc(&out, 4096));
    run<0>("MFMA 32x32x16 bf16, 4 accumulators", out);
    run<3>("MFMA, 1 dependent chain", out);
    run<2>("8 x v_fma", out);
    run<1>("MFMA + 8 v_fma, same wave", out);
    run<5>("ds_read_b128", out);
    run<4>("MFMA + ds_read_b128, same wave", out);
    run<6>("MFMA + 8 v_fma + ds_read_b128, same wave", out);
    run<10>("MIX: waves 0-3 MFMA | waves 4-7 8 x v_fma", out);
    run<11>("MIX: waves 0-3 MFMA | waves 4-7 ds_read_b128", out);
    return 0;
}
