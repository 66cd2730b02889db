// ubench_valu2.hip — issue cost of the select / compare / clamp / convert forms of the photometric kernels on gfx950 (companion of
// ubench_valu.hip, same method: ITER x 16 x 8 instances of one form per wave, inline asm, 4 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu2.hip -o tools/bin/ubench_valu2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 512;
#define REP16(S) S S S S S S S S S S S S S S S S
#define FORM8(OP, TAIL) OP " %0, %0, %1" TAIL "\n " OP " %1, %1, %2" TAIL "\n " OP " %2, %2, %3" TAIL "\n " OP " %3, %3, %4" TAIL "\n " \
                        OP " %4, %4, %5" TAIL "\n " OP " %5, %5, %6" TAIL "\n " OP " %6, %6, %7" TAIL "\n " OP " %7, %7, %0" TAIL "\n"
#define FORM8U(OP) OP " %0, %1\n " OP " %1, %2\n " OP " %2, %3\n " OP " %3, %4\n " OP " %4, %5\n " OP " %5, %6\n " OP " %6, %7\n " OP " %7, %0\n"
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int FORM>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long mask) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    for (int i = 0; i < ITER; ++i) {
        if (FORM == 0) asm volatile(REP16(FORM8("v_add_f32", "")) REGS);
        else if (FORM == 1) asm volatile(REP16(FORM8("v_mul_f32", "")) REGS);
        else if (FORM == 2) asm volatile(REP16(FORM8("v_max_f32", "")) REGS);
        else if (FORM == 3) asm volatile("s_mov_b64 vcc, %8\n" REP16(FORM8("v_cndmask_b32", ", vcc")) REGS : "s"(mask) : "vcc");
        else if (FORM == 4) asm volatile(REP16(FORM8("v_cndmask_b32", ", %8")) REGS : "s"(mask));
        else if (FORM == 5) asm volatile(REP16(FORM8U("v_floor_f32")) REGS);
        else if (FORM == 6) asm volatile(REP16(FORM8U("v_cvt_i32_f32")) REGS);
        else if (FORM == 7) asm volatile(REP16("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                                               "v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_lt_f32 vcc, %5, %6\n v_cndmask_b32 %7, %7, %0, vcc\n") REGS : : "vcc");
        else if (FORM == 8) asm volatile(REP16(FORM8("v_med3_f32", ", %0")) REGS);
        else if (FORM == 9) asm volatile(REP16(FORM8("v_sub_f32", "")) REGS);
        else if (FORM == 10) asm volatile(REP16("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %4, %4, %6\n v_pk_add_f32 %6, %6, %0\n"
                                                "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %4, %4, %0\n v_pk_add_f32 %6, %6, %2\n")
                                          : "+v"(*(double *)&a0), "+v"(a1), "+v"(*(double *)&a2), "+v"(a3), "+v"(*(double *)&a4), "+v"(a5), "+v"(*(double *)&a6), "+v"(a7));
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 123.456f) out[0] = r;
}

template <int FORM>
void run(const char *name, float *out) {
    const int w = 4, blocks = 256 * w;
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, 0x5555aaaa3333ccccull);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, 0x5555aaaa3333ccccull);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ninstr = (double)ITER * 16 * 8;
    printf("%-34s 4 waves/SIMD: %.3f ns per wave-instr per SIMD\n", name, ms * 1e6 / 5 / (ninstr * w));
}

int main() {
    float *out;
    CHECK(hipMalloc(&out, 1024));
    run<0>("v_add_f32", out);
    run<9>("v_sub_f32", out);
    run<1>("v_mul_f32", out);
    run<2>("v_max_f32", out);
    run<8>("v_med3_f32", out);
    run<3>("v_cndmask_b32 (vcc)", out);
    run<4>("v_cndmask_b32 (sgpr pair, e64)", out);
    run<7>("v_cmp_lt_f32 + v_cndmask (per pair)", out);
    run<5>("v_floor_f32", out);
    run<6>("v_cvt_i32_f32", out);
    return 0;
}
