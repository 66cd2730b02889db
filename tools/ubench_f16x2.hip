// ubench_f16x2.hip — hardware facts the two-term fp16 operand split rests on (dev tool, round 5)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_f16x2.hip -o tools/bin/ubench_f16x2 && tools/bin/ubench_f16x2
// 1. does v_mfma_f32_32x32x16_f16 keep subnormal fp16 inputs (the low term of a small element is subnormal)?
// 2. does v_cvt_pk_f16_f32 round to nearest-even into the subnormal range?
// 3. issue cost per converted element: three bf16 terms (truncating split, the round-2..4 arithmetic) vs two fp16 terms
//    (plain C and with v_fma_mix_f32 for the residual), VALU only, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_denorm(float *out, float av, float bv) {
    // A[i][k] = av for all i,k ; B[k][j] = bv: C = 16 * av * bv
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)av; b[j] = (_Float16)bv; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
__global__ void k_cvt(const float *in, float *out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        h2 h = __builtin_convertvector((f2){in[i], in[i]}, h2);
        out[i] = (float)h[0];
    }
}

struct Split4 { uint2 t[3]; };
__device__ __forceinline__ unsigned hi16pair(float lo_elem, float hi_elem) {
    return __builtin_amdgcn_perm(__float_as_uint(hi_elem), __float_as_uint(lo_elem), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ Split4 split3(float4 v) {
    Split4 r;
    r.t[0] = make_uint2(hi16pair(v.x, v.y), hi16pair(v.z, v.w));
    const float4 a = make_float4(v.x - trunc_bf16(v.x), v.y - trunc_bf16(v.y), v.z - trunc_bf16(v.z), v.w - trunc_bf16(v.w));
    r.t[1] = make_uint2(hi16pair(a.x, a.y), hi16pair(a.z, a.w));
    const float4 b = make_float4(a.x - trunc_bf16(a.x), a.y - trunc_bf16(a.y), a.z - trunc_bf16(a.z), a.w - trunc_bf16(a.w));
    r.t[2] = make_uint2(hi16pair(b.x, b.y), hi16pair(b.z, b.w));
    return r;
}
__device__ __forceinline__ void split2c(float4 v, float s, uint2 &hi, uint2 &lo) {
    const float x0 = v.x * s, x1 = v.y * s, x2 = v.z * s, x3 = v.w * s;
    const h2 a = __builtin_convertvector((f2){x0, x1}, h2), b = __builtin_convertvector((f2){x2, x3}, h2);
    const h2 c = __builtin_convertvector((f2){x0 - (float)a[0], x1 - (float)a[1]}, h2);
    const h2 d = __builtin_convertvector((f2){x2 - (float)b[0], x3 - (float)b[1]}, h2);
    hi = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    lo = make_uint2(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d));
}
// residual x*s - h through v_fma_mix_f32 (src2 = the fp16 half, negated): one instruction instead of cvt + sub, the scale folded in
__device__ __forceinline__ float resid_lo(float x, float s, unsigned hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(hpk));
    return r;
}
__device__ __forceinline__ float resid_hi(float x, float s, unsigned hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(hpk));
    return r;
}
__device__ __forceinline__ void split2m(float4 v, float s, uint2 &hi, uint2 &lo) {
    const h2 a = __builtin_convertvector((f2){v.x * s, v.y * s}, h2), b = __builtin_convertvector((f2){v.z * s, v.w * s}, h2);
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    const h2 c = __builtin_convertvector((f2){resid_lo(v.x, s, ua), resid_hi(v.y, s, ua)}, h2);
    const h2 d = __builtin_convertvector((f2){resid_lo(v.z, s, ub), resid_hi(v.w, s, ub)}, h2);
    hi = make_uint2(ua, ub);
    lo = make_uint2(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d));
}

template <int FORM>
__global__ __launch_bounds__(256) void k_split(const float4 *in, unsigned *out, float s, int iters) {
    float4 v = in[threadIdx.x];
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (FORM == 0) {
                const Split4 sp = split3(v);
                acc ^= sp.t[0].x ^ sp.t[0].y ^ sp.t[1].x ^ sp.t[1].y ^ sp.t[2].x ^ sp.t[2].y;
            } else {
                uint2 hi, lo;
                if (FORM == 1) split2c(v, s, hi, lo); else split2m(v, s, hi, lo);
                acc ^= hi.x ^ hi.y ^ lo.x ^ lo.y;
            }
            v.x += 1.0f; v.y += 3.0f; v.z += 5.0f; v.w += 7.0f;       // 4 VALU of loop overhead per 4 elements, the same for every form
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void k_check(const float4 *in, float *err, float s, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint2 h1, l1, h2_, l2;
    split2c(in[i], s, h1, l1);
    split2m(in[i], s, h2_, l2);
    err[i] = (h1.x == h2_.x && h1.y == h2_.y && l1.x == l2.x && l1.y == l2.y) ? 0.f : 1.f;
}

template <int BF>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters) {
    u32x4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, 6, threadIdx.x};
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        if (BF) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        }
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 123.456f) out[0] = 1.f;
}

int main() {
    float *d_out, *d_in;
    CHECK(hipMalloc(&d_out, 1 << 22));
    CHECK(hipMalloc(&d_in, 1 << 22));
    float h[8];
    // 1. subnormal inputs
    const float cases[4][2] = {{1.0f, 1.0f}, {ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 32768.f}, {ldexpf(3.f, -24), ldexpf(5.f, -24)}};
    for (int i = 0; i < 4; ++i) {
        hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, d_out, cases[i][0], cases[i][1]);
        CHECK(hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost));
        printf("mfma f16: a=%g b=%g -> c=%.9g (exact %.9g)  a as f16 reads back %g\n", cases[i][0], cases[i][1], h[0], 16.0 * cases[i][0] * cases[i][1], h[1]);
    }
    // 2. conversion into the subnormal range
    float vin[6] = {ldexpf(1.f, -15), ldexpf(1.5f, -24), ldexpf(2.5f, -24), ldexpf(1.f, -26), 65504.f, 65520.f};
    CHECK(hipMemcpy(d_in, vin, sizeof(vin), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, d_in, d_out, 6);
    CHECK(hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost));
    for (int i = 0; i < 6; ++i) printf("cvt_pk_f16_f32(%.9g) = %.9g\n", vin[i], h[i]);
    // 2b. the v_fma_mix_f32 split equals the plain one
    {
        const int n = 1 << 16;
        float *hv = (float *)malloc(n * 16);
        srand(1);
        for (int i = 0; i < n * 4; ++i) hv[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 40 - 30);
        CHECK(hipMemcpy(d_in, hv, n * 16, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, (const float4 *)d_in, d_out, 1024.f, n);
        float *he = (float *)malloc(n * 4);
        CHECK(hipMemcpy(he, d_out, n * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += he[i] != 0.f;
        printf("fma_mix split vs plain split: %d of %d float4 differ\n", bad, n);
    }
    // 3. issue cost
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    const char *names[3] = {"three bf16 terms (truncating)", "two fp16 terms, plain", "two fp16 terms, v_fma_mix"};
    for (int form = 0; form < 3; ++form) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (form == 0) hipLaunchKernelGGL(k_split<0>, dim3(1024), dim3(256), 0, 0, (const float4 *)d_in, (unsigned *)d_out, 1024.f, iters);
            if (form == 1) hipLaunchKernelGGL(k_split<1>, dim3(1024), dim3(256), 0, 0, (const float4 *)d_in, (unsigned *)d_out, 1024.f, iters);
            if (form == 2) hipLaunchKernelGGL(k_split<2>, dim3(1024), dim3(256), 0, 0, (const float4 *)d_in, (unsigned *)d_out, 1024.f, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // 1024 blocks x 4 waves on 256 CUs x 4 SIMDs = 4 waves per SIMD; each wave converts iters * 8 * 4 elements
        const double ns_per_elem_simd = ms * 1e6 / (4.0 * iters * 8 * 4);
        printf("%-34s %.3f ns per wave-element per SIMD (incl. 1 VALU of loop overhead per element)\n", names[form], ns_per_elem_simd);
    }
    for (int bf = 0; bf < 2; ++bf) {
        const int it = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (bf) hipLaunchKernelGGL(k_mfma<1>, dim3(1024), dim3(256), 0, 0, d_out, it);
            else hipLaunchKernelGGL(k_mfma<0>, dim3(1024), dim3(256), 0, 0, d_out, it);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 1024.0 * 4 * it * 4 * 2.0 * 32 * 32 * 16;
        printf("v_mfma_f32_32x32x16_%s: %.0f TFLOP/s\n", bf ? "bf16" : "f16", flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
