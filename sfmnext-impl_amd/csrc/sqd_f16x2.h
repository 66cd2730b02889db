// sqd_f16x2.h — two-term fp16 operands for the head kernels (bins.hip, sql.hip), the arithmetic conv.hip's PREC 5 plans introduced.
//
// An fp32 value x is the sum s^-1 (h + l) of two fp16 terms of x s (round to nearest: |x s - h - l| <= 2^-22 |x s|), s a power of two that
// puts the largest magnitude of the scaled group into [2^14, 2^15): h never overflows, elements within 2^18 of the maximum keep a normal low
// term, smaller ones an absolute accuracy of 2^-40 of the maximum (fp16 subnormals, which v_cvt_pk_f16_f32 produces and
// v_mfma_f32_32x32x16_f16 consumes: tools/ubench_f16x2.hip).  A product keeps h_a h_b + h_a l_b + l_a h_b (every partial product exact in
// the fp32 accumulator; l_a l_b <= 2^-22 is dropped): three matrix instructions at 16x the fp32 instruction's rate.  The scale of an operand
// may vary along its FREE dimension (a row of A, a column of B), never along the contraction.
//
// Operand layout of v_mfma_f32_32x32x16_f16 (tools/ubench_tr.hip checks it on the hardware): lane l holds A[l % 32][8 (l / 32) + j] and
// B[8 (l / 32) + j][l % 32], j = 0..7 — "k-slot (h, j)", h = l / 32; accumulator register r of lane l is C[(r & 3) + 8 (r >> 2) + 4 h][l % 32].
// Which contraction index a k-slot stands for is the kernel's choice, as long as both operands agree.
#pragma once
#include "sqd_common.h"

namespace sqd {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_h16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// a b with both operands as (high, low) pairs: three instructions
__device__ __forceinline__ f32x16 mfma_h16x2(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x16 c) {
    c = mfma_h16(ah, bh, c);
    c = mfma_h16(ah, bl, c);
    return mfma_h16(al, bh, c);
}

// biased exponent of the scale for a group whose max |.| has the bit pattern `abits` (abs_bits); s = 2^(be - 127) puts the maximum into
// [2^14, 2^15); an all-zero group takes the largest scale (0 s = 0)
__device__ __forceinline__ unsigned h2_scale_exp(unsigned abits) { return (unsigned)min(max(268 - (int)((abits >> 23) & 0xffu), 1), 253); }
__device__ __forceinline__ float h2_scale(unsigned be) { return __uint_as_float(be << 23); }
__device__ __forceinline__ float h2_inv_scale(unsigned be) { return __uint_as_float((254u - be) << 23); }

// x s - h in one instruction: v_fma_mix_f32 reads src2 as the (negated) fp16 half of the packed high terms; exact
__device__ __forceinline__ float h2_resid_lo(float x, float s, unsigned hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(hpk));
    return r;
}
__device__ __forceinline__ float h2_resid_hi(float x, float s, unsigned hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s), "v"(hpk));
    return r;
}
// two elements -> packed high terms, packed low terms
__device__ __forceinline__ void h2_split2(float x, float y, float s, unsigned &hi, unsigned &lo) {
    const h16x2 a = __builtin_convertvector((f32x2){x * s, y * s}, h16x2);
    hi = __builtin_bit_cast(unsigned, a);
    const h16x2 c = __builtin_convertvector((f32x2){h2_resid_lo(x, s, hi), h2_resid_hi(y, s, hi)}, h16x2);
    lo = __builtin_bit_cast(unsigned, c);
}
// the eight elements of a k-slot group -> one operand register quad of high terms and one of low terms
__device__ __forceinline__ void h2_split8(const float (&v)[8], float s, u32x4 &hi, u32x4 &lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    h2_split2(v[0], v[1], s, h0, l0);
    h2_split2(v[2], v[3], s, h1, l1);
    h2_split2(v[4], v[5], s, h2, l2);
    h2_split2(v[6], v[7], s, h3, l3);
    hi = (u32x4){h0, h1, h2, h3};
    lo = (u32x4){l0, l1, l2, l3};
}

// ds_read_b64_tr_b16 (gfx950 LDS transpose read; tools/ubench_tr.hip): the 16 lanes of a group each address 4 contiguous 16-bit elements —
// lane t row t / 4, columns 4 (t % 4) .. + 3 of a [4][16] block whose row pitch is free — and lane c receives COLUMN c: {block[0][c],
// block[1][c], block[2][c], block[3][c]}.  `p` = this lane's own address (8-byte aligned).
__device__ __forceinline__ uint2 lds_read_tr16(const unsigned short *p) {
    const i16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3))) *)(p));
    return __builtin_bit_cast(uint2, v);
}
}  // namespace sqd
