// ubench_valu.hip — issue cost of the VALU forms the photometric kernels are made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
// Each kernel runs ITER x 16 independent instances of ONE instruction form per wave (inline asm, so the compiler neither
// fuses nor removes them); grid = 256 CUs x 4 SIMDs x W waves per SIMD.  Reported: SIMD cycles per wave-instruction
// (at the clock measured by a plain v_fma_f32 stream assumed to cost 2 cycles... no assumption needed: s_memtime ticks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 512;

#define REP16(S) S S S S S S S S S S S S S S S S

template <int FORM>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f;
    const v2f pm = {0.999f, 1.001f};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
        if (FORM == 0) {          // v_fma_f32
            asm volatile(REP16("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                               "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (FORM == 1) {   // v_pk_fma_f32
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %8, %1\n v_pk_fma_f32 %1, %1, %8, %2\n v_pk_fma_f32 %2, %2, %8, %3\n v_pk_fma_f32 %3, %3, %8, %4\n"
                               "v_pk_fma_f32 %4, %4, %8, %5\n v_pk_fma_f32 %5, %5, %8, %6\n v_pk_fma_f32 %6, %6, %8, %7\n v_pk_fma_f32 %7, %7, %8, %0\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
        } else if (FORM == 2) {   // v_add_f32 dpp wave_shr:1
            asm volatile(REP16("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %2, %3, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %4, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %5, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %6, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %6, %7, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %0, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (FORM == 3) {   // v_rcp_f32
            asm volatile(REP16("v_rcp_f32 %0, %1\n v_rcp_f32 %1, %2\n v_rcp_f32 %2, %3\n v_rcp_f32 %3, %4\n v_rcp_f32 %4, %5\n v_rcp_f32 %5, %6\n v_rcp_f32 %6, %7\n v_rcp_f32 %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (FORM == 4) {   // v_mov_b32
            asm volatile(REP16("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (FORM == 5) {   // v_add_f32 dpp row_shr:1 (row-local)
            asm volatile(REP16("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %0, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (FORM == 6) {   // v_pk_mul_f32
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));
        } else if (FORM == 7) {   // v_cndmask_b32 (VOP3, sgpr-pair mask via vcc)
            asm volatile(REP16("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                               "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (FORM == 8) {   // ds_bpermute_b32 (LDS crossbar, no memory)
            int i0 = __builtin_bit_cast(int, a0), ad = ((threadIdx.x + 1) & 63) * 4;
            asm volatile(REP16("ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n"
                               "ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n")
                         : "+v"(i0) : "v"(ad));
            a0 = __builtin_bit_cast(float, i0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if (r == 123.456f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int FORM>
void run(const char *name, float *out, long long *cyc) {
    for (int w = 1; w <= 4; w *= 2) {
        const int blocks = 256 * w;          // 256-thread blocks = one wave per SIMD each
        hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, cyc);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long c;
        CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        const double ninstr = (double)ITER * 16 * 8;
        const double ns_per = ms * 1e6 / 5 / (ninstr * w);                 // ns of SIMD time per wave-instruction
        printf("%-26s waves/SIMD %d: %.3f ns per wave-instr per SIMD (%.2f cyc @2.4GHz), s_memtime ticks/instr (wave 0) %.2f\n", name, w, ns_per,
               ns_per * 2.4, (double)c / ninstr);
    }
}

int main() {
    float *out;
    long long *cyc;
    CHECK(hipMalloc(&out, 1024));
    CHECK(hipMalloc(&cyc, 64));
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<6>("v_pk_mul_f32", out, cyc);
    run<2>("v_add_f32_dpp wave_shr:1", out, cyc);
    run<5>("v_add_f32_dpp row_shr:1", out, cyc);
    run<3>("v_rcp_f32", out, cyc);
    run<4>("v_mov_b32", out, cyc);
    run<7>("v_cndmask_b32", out, cyc);
    run<8>("ds_bpermute_b32", out, cyc);
    return 0;
}
