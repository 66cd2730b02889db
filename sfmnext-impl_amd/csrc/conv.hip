// conv.hip — 2-D convolution as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32),
// channels-last activations ([N,H,W,C] memory) and KRSC weights (= torch channels_last weight memory).
// replaces: every nn.Conv2d of the networks (reference networks/resnet_encoder.py, pose_cnn.py,
//           depth_decoder_QTR.py) — forward, data gradient and weight gradient.
//
//   forward : y[m, k]  = sum_{r,s,c} x[pix(m; r,s), c] * w[k, r, s, c]        M = N*Ho*Wo, Kgemm = R*S*C
//   dgrad   : dx[p, c] = sum_{r,s,k} dy[opix(p; r,s), k] * w[k, r, s, c]      M = N*H*W,   Kgemm = R*S*K
//   wgrad   : dw[k,r,s,c] = sum_m dy[m, k] * x[pix(m; r,s), c]                split over pixel ranges
//
// Tiling: 4 (or 8) wavefronts per workgroup, BM x BN output tile, a reduction slice of 16 / 32 / 64 channels per step held in
// LDS as [row][slice (+4 pad)] for both operands — exactly the memory order of NHWC pixels and KRSC filters, so staging is
// plain 16-byte copies.  A lane feeds the MFMA with 8 consecutive reduction elements of its row (two ds_read_b128,
// conflict-free with the padded row pitch); the two half-waves own the two halves of the slice.  Global loads of slice t+1
// are issued before the MFMAs of slice t (register staging) and written to the other LDS buffer after them — or, in the
// single-buffered variants, to the same buffer behind one more barrier: half the LDS per workgroup means twice the resident
// workgroups, which at batch-12 grid sizes (about two waves per SIMD resident) is worth more than the saved barrier.
// Every global access is a raw buffer load / store: padding taps, rows beyond the tile and channels beyond the tensor get an
// offset beyond the descriptor's extent (reads return 0, writes are dropped) instead of a branch.  Which tile, slice width,
// split-K factor and buffering a layer uses is measured per geometry by the caller (sqd_conv_set_plan).
// fp32 MFMA is an exact k-ordered fmaf chain, so results match a direct fp32 convolution to accumulation-order rounding.
// Roofline: fp32 MFMA, 157.3 TFLOP/s dense.
#include "sqd_common.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

namespace {
using namespace sqd;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16, LDP = 20;     // reduction slice, LDS row pitch (floats)

// BatchNorm-backward statistics in the data-gradient epilogue (sqd_conv_dgrad_bn): the tensor this convolution differentiates is
// the output act(BN(x_bn) [+ res]) of a training-mode BatchNorm; its backward needs per channel  sum dz  and  sum dz * xhat
// (dz = dx * act'(.), xhat = (x_bn - mean) * rstd) — sums over exactly the elements the epilogue holds.  x == NULL: off.
struct BnBwdSrc {
    const float *x;                 // the BatchNorm's input [N,H,W,C] (the producing convolution's output)
    const unsigned char *mask;      // sign bits of its pre-activation, 1 byte per 4 channels (ReLU / LeakyReLU); NULL: no activation
    const float *mean, *rstd;       // [C] batch statistics of the forward
    int act;                        // 0 none, 1 ReLU, 2 LeakyReLU(0.01)
};
__device__ __forceinline__ float bn_bwd_act(unsigned bits, int j, int act) {
    const bool pos = (bits >> j) & 1u;
    return act == 1 ? (pos ? 1.f : 0.f) : act == 2 ? (pos ? 1.f : 0.01f) : 1.f;
}

struct ConvGeom {
    int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- split-precision operands (PREC = 1): an fp32 value is the exact sum of three bf16 terms (8 significant bits each:
// hi = its upper 16 bits, mid = the upper 16 bits of x - hi, lo = the rest), so a product a*b is the sum of 9 bf16 x bf16
// products, each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  The kernel keeps the
// 6 terms down to 2^-24 relative (drops mid*lo, lo*mid, lo*lo) — fp32-level accuracy at 6/16 of the matrix-core time.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split4 {
    uint2 t[3];                       // term -> 4 bf16 (elements 0..3, element 0 in the low half of .x)
};
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
// round-to-nearest-even bf16 of a finite fp32 (PREC 3)
__device__ __forceinline__ unsigned rne_bf16(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <int NTERM>
__device__ __forceinline__ Split4 splitN(float4 v) {
    if (NTERM == 3) return split3(v);
    Split4 r;
    r.t[0] = make_uint2(rne_bf16(v.x) | (rne_bf16(v.y) << 16), rne_bf16(v.z) | (rne_bf16(v.w) << 16));
    r.t[1] = r.t[2] = make_uint2(0u, 0u);
    return r;
}
__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- two-term fp16 operands (PREC 5, "f16x2", round 5).  Three bf16 terms cost 5.5 VALU instructions per staged element and six
// matrix instructions per product; an fp32 value is just as well the sum  s^-1 (h + l)  of two fp16 terms (11 significant bits each,
// round-to-nearest: |x s - h - l| <= 2^-22 |x s|, rms 2^-24.8 — the size of ONE fp32 rounding) once the tensor has been scaled by
// a power of two s that puts its largest magnitude into [2^14, 2^15): h never overflows fp16 and the low term of every element
// within 2^18 of the maximum is a normal number (smaller elements keep an absolute accuracy of 2^-40 of the maximum: fp16
// subnormals, which v_cvt_pk_f16_f32 produces and v_mfma_f32_32x32x16_f16 consumes — tools/ubench_f16x2.hip).  The kernel keeps
// h_a h_b + h_a l_b + l_a h_b (drops l_a l_b <= 2^-22), every product exact in the fp32 accumulator: three matrix instructions and
// 3-4 conversion instructions per element instead of six and 5.5, two LDS planes instead of three.  Against float64 the result is
// as close as the fp32 MFMA chain's (tests/test_gpu_conv.py::test_f16x2_*; oracle/f16x2_model.py).  The scale comes from the
// tensor's max |.|, which the producing kernel leaves in device memory (sqd_amax_* / the `amax` outputs of the producers): a
// record of 64 words in 64 cache lines (SQD_AMAX_WAYS, sqd_common.h) holding bit patterns of non-negative floats.  s = 2^(141 - biased exponent), clamped to a normal float.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
struct OpScale {
    const float *amax_a, *amax_b;     // max |.| of the A operand (x / dy) and of the B operand (w; x for the weight gradient); NULL: unscaled plans
    unsigned *amax_out;               // may be NULL: where the epilogue records max |output| (for the convolution that reads the output next)
};
__device__ __forceinline__ unsigned scale_exp(const float *amax) {        // biased exponent of s
    const int e = (int)((amax_record_bits(amax) >> 23) & 0xffu);
    return (unsigned)min(max(268 - e, 1), 253);
}
__device__ __forceinline__ float scale_from_exp(unsigned be) { return __uint_as_float(be << 23); }
__device__ __forceinline__ float inv_scale_from_exp(unsigned be) { return __uint_as_float((254u - be) << 23); }
struct Split4h {
    uint2 t[2];                       // t[0] = four high terms, t[1] = four low terms (element 0 in the low half of .x)
};
#ifndef SQD_F16X2_PLAIN
// x s - h in one instruction: v_fma_mix_f32 reads src2 as the (negated) fp16 half of the packed high terms; exact (the result is
// representable), identical bits to the convert-and-subtract form (tools/ubench_f16x2.hip checks 2^18 values)
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
#endif
__device__ __forceinline__ Split4h split2h(float4 v, float s) {
    Split4h r;
    const f16x2v a = __builtin_convertvector((f32x2v){v.x * s, v.y * s}, f16x2v), b = __builtin_convertvector((f32x2v){v.z * s, v.w * s}, f16x2v);
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
#ifdef SQD_F16X2_PLAIN
    const f16x2v c = __builtin_convertvector((f32x2v){v.x * s - (float)a[0], v.y * s - (float)a[1]}, f16x2v);
    const f16x2v d = __builtin_convertvector((f32x2v){v.z * s - (float)b[0], v.w * s - (float)b[1]}, f16x2v);
#else
    const f16x2v c = __builtin_convertvector((f32x2v){resid_lo(v.x, s, ua), resid_hi(v.y, s, ua)}, f16x2v);
    const f16x2v d = __builtin_convertvector((f32x2v){resid_lo(v.z, s, ub), resid_hi(v.w, s, ub)}, f16x2v);
#endif
    r.t[0] = make_uint2(ua, ub);
    r.t[1] = make_uint2(__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d));
    return r;
}
__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the staged terms of either arithmetic behind one interface: NT terms of four elements
template <int NT> struct SplitT { uint2 t[NT]; };
template <int NT, bool H2>
__device__ __forceinline__ SplitT<NT> split_terms(float4 v, float s) {
    SplitT<NT> r;
    if constexpr (H2) {
        const Split4h h = split2h(v, s);
        r.t[0] = h.t[0];
        r.t[1 % NT] = h.t[1];
    } else {
        const Split4 b = splitN<NT>(v);
#pragma unroll
        for (int i = 0; i < NT; ++i) r.t[i] = b.t[i];
    }
    return r;
}
// acc += A B over the kept term pairs, smallest first: 6 of 9 (three bf16 terms), 3 of 4 (two fp16 terms), 1 (one bf16 term)
template <int NT, bool H2>
__device__ __forceinline__ f32x16 mma_terms(const u32x4 *a, const u32x4 *b, f32x16 c) {
    if constexpr (H2) {
        c = mfma_h(a[0], b[1], c);
        c = mfma_h(a[1], b[0], c);
        return mfma_h(a[0], b[0], c);
    } else if constexpr (NT == 1) {
        return mfma_bf(a[0], b[0], c);
    } else {
        c = mfma_bf(a[0], b[2], c);
        c = mfma_bf(a[2], b[0], c);
        c = mfma_bf(a[1], b[1], c);
        c = mfma_bf(a[0], b[1], c);
        c = mfma_bf(a[1], b[0], c);
        return mfma_bf(a[0], b[0], c);
    }
}

// Operand fetches go through raw buffer loads: the descriptor carries the tensor's size, an offset beyond it returns zeros.
// Padding taps, rows beyond the tile and channels beyond the tensor set the offset to 0xffffffff (one v_cndmask) instead of
// branching around the load (s_and_saveexec / s_cbranch_execz per float4 in the global_load version).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

// MODE 0: forward (A rows = output pixels, gather x; B rows = out channels k, reduction over (r,s,c))
// MODE 1: dgrad   (A rows = input pixels, gather dy; B rows = in channels c, reduction over (r,s,k))
template <int MODE, int BM, int BN, int WGM, int WGN, int BKT, int PREC = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_gemm_kernel(const float *__restrict__ a_src, const float *__restrict__ wgt,
                                                        const float *__restrict__ bias, float *__restrict__ out, ConvGeom g,
                                                        int act, int zsplits, int order, float *__restrict__ stats, BnBwdSrc bnb,
                                                        OpScale sc) {
    constexpr int WTM = BM / WGM / 32, WTN = BN / WGN / 32;     // 32x32 MFMA tiles per wave
    constexpr int NT = WGM * WGN * 64;                          // 4 or 8 wavefronts per workgroup
    static_assert((WGM * WGN == 4 || WGM * WGN == 8) && WTM >= 1 && WTN >= 1, "4 or 8 waves per workgroup");
    // BKT = reduction slice per step (16 or 32 channels): a wider slice halves the barriers and per-step bookkeeping for
    // twice the LDS footprint
    constexpr int NQ = BKT / 4, LDPT = BKT + 4, RPP = NT / NQ;  // float4 per staged row, LDS row pitch, rows per staging pass
    constexpr int A_F4 = BM * NQ / NT;
    static_assert(A_F4 >= 1 && (BKT == 16 || BKT == 32 || BKT == 64), "BM >= 64, BKT in {16, 32, 64}");
    constexpr bool H2 = PREC == 5;                              // PREC 5: two fp16 terms of the power-of-two scaled operands (f16x2, see above)
    constexpr bool BF = PREC == 1 || PREC == 3 || PREC == 4 || H2;   // 16-bit operands: PREC 1 / 4 = three-term bf16 split (fp32-level accuracy),
    constexpr int NTERM = (PREC == 1 || PREC == 4) ? 3 : H2 ? 2 : 1; // PREC 3 = one round-to-nearest bf16 term (bf16 training arithmetic)
    constexpr int NBUF = (PREC == 2 || PREC == 4 || H2) ? 1 : 2;     // PREC 2 / 4 / 5: single LDS buffer (half the LDS, one more barrier per slice)
    constexpr int LDH = BKT + 8;                                // PREC 1: bf16 row pitch (48 / 80 bytes: conflict-free b128 reads)
    __shared__ __attribute__((aligned(16))) float As[BF ? 1 : NBUF][BF ? 1 : BM][LDPT];
    __shared__ __attribute__((aligned(16))) float Bs[BF ? 1 : NBUF][BF ? 1 : BN][LDPT];
    __shared__ __attribute__((aligned(16))) unsigned short Ah[BF ? NBUF : 1][NTERM][BF ? BM : 1][LDH];   // [buffer][term][row][k]
    __shared__ __attribute__((aligned(16))) unsigned short Bh[BF ? NBUF : 1][NTERM][BF ? BN : 1][LDH];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm0 = (wave / WGN) * (WTM * 32), wn0 = (wave % WGN) * (WTN * 32);
    const unsigned bea = H2 ? scale_exp(sc.amax_a) : 127u, beb = H2 ? scale_exp(sc.amax_b) : 127u;      // (scalar loads, wave-uniform)
    const float sca = scale_from_exp(bea), scb = scale_from_exp(beb);
    // workgroups are dealt round-robin to the 8 XCDs (private L2 each): XCD x gets the contiguous range
    // [x*G/8, (x+1)*G/8) of the logical tile order, in which the N-tiles of one M-tile are neighbours, so the
    // activation tile they share is fetched into that L2 once
    // order bit 0: XCD chunking, bit 1: N-tiles fastest (else M-tiles fastest), bit 2: stride classes interleaved
    const int tiles_n = (MODE == 0 ? g.K + BN - 1 : g.C + BN - 1) / BN;
    const int logical = (order & 1) ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    // dgrad runs one dense problem per stride class (ph, pw) = (hi % stride, wi % stride): the input pixels of a
    // class all see the same taps r = r0 + stride*jr, s = s0 + stride*js, with ho = hi' + base_h - jr, so no
    // slice is spent on taps that do not divide (stride 1: a single class, all taps)
    const int Hc = (g.H + g.stride - 1) / g.stride, Wc = (g.W + g.stride - 1) / g.stride;
    const int Mrows = MODE == 0 ? g.N * g.Ho * g.Wo : g.N * Hc * Wc;      // rows of one GEMM (per class for dgrad)
    const int tiles_m = (Mrows + BM - 1) / BM;
    // stride classes interleave (class index varies faster than the M-tile) so that every XCD gets its share of the
    // classes that carry taps
    const int ncls_all = MODE == 0 ? 1 : g.stride * g.stride;
    const int tiles_m_all = tiles_m * ncls_all;
    if (logical >= tiles_m_all * tiles_n) return;                // grid padding (multiple of 8)
    const int tile_m_all = (order & 2) ? logical / tiles_n : logical % tiles_m_all;
    const int n0 = ((order & 2) ? logical % tiles_n : logical / tiles_m_all) * BN;
    const int cls = (order & 4) ? tile_m_all % ncls_all : tile_m_all / tiles_m;
    const int m0 = ((order & 4) ? tile_m_all / ncls_all : tile_m_all % tiles_m) * BM;
    const int ph = cls / g.stride, pw = cls - ph * g.stride;
    const int r0 = (ph + g.pad) % g.stride, s0 = (pw + g.pad) % g.stride;
    const int Rc = MODE == 0 ? g.R : (r0 < g.R ? (g.R - r0 + g.stride - 1) / g.stride : 0);
    const int Sc = MODE == 0 ? g.S : (s0 < g.S ? (g.S - s0 + g.stride - 1) / g.stride : 0);
    const int Ncols = MODE == 0 ? g.K : g.C;                    // output channels of this GEMM
    const int Cred = MODE == 0 ? g.C : g.K;                     // channels reduced per (r,s)
    const int cchunks = (Cred + BKT - 1) / BKT;              // the last chunk may be partial (Cred % 4 == 0): its tail is masked
    const int Tall = Rc * Sc * cchunks;
    // split-K: workgroup z reduces slices [s_beg, s_end) and writes a partial tile (summed by gemm_reduce_kernel)
    const int s_beg = (int)((long long)Tall * blockIdx.z / zsplits), s_end = (int)((long long)Tall * (blockIdx.z + 1) / zsplits);
    const int T = s_end - s_beg;

    // ---- per-thread A staging rows: float4 column c4 of rows t / NQ + RPP*i
    // bf16 tiles with 32-channel slices (row pitch 80 bytes = 20 banks): the 32 lanes an 8-byte LDS store serves per clock cover
    // four staging rows, and rows r, r+1, r+2, r+3 wrap onto each other's banks (a third of the LDS cycles were conflicts).  Rows
    // r, r+4, r+8, r+12 start 16 banks apart: staging row x of a group of 16 is dealt as 4 * (x % 4) + x / 4 (any row order is a
    // valid one — the rows of a tile are independent).
    auto srow = [](int x) { return (BF && BKT == 32) ? ((x & ~15) | ((x & 3) << 2) | ((x >> 2) & 3)) : x; };
    const int c4 = t % NQ, arow = srow(t / NQ);
    int an[A_F4], ah[A_F4], aw[A_F4];
    bool aval[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int m = m0 + arow + RPP * i;
        aval[i] = m < Mrows;
        const int mm = aval[i] ? m : 0;
        if (MODE == 0) {
            const int wo = mm % g.Wo, t2 = mm / g.Wo;
            an[i] = t2 / g.Ho;
            ah[i] = (t2 % g.Ho) * g.stride - g.pad;
            aw[i] = wo * g.stride - g.pad;
        } else {
            const int wq = mm % Wc, t2 = mm / Wc, hq = t2 % Hc;
            an[i] = t2 / Hc;
            aval[i] = aval[i] && hq * g.stride + ph < g.H && wq * g.stride + pw < g.W;
            ah[i] = hq + (ph + g.pad - r0) / g.stride;
            aw[i] = wq + (pw + g.pad - s0) / g.stride;
        }
    }
    __shared__ int rowpix[MODE == 1 ? BM : 1];               // dgrad: output pixel index of each tile row (-1 = none)
    if (MODE == 1 && t < BM) {
        const int m = m0 + t;
        const int wq = m % Wc, t2 = m / Wc, hq = t2 % Hc, n = t2 / Hc;
        const int hi = hq * g.stride + ph, wi = wq * g.stride + pw;
        rowpix[t] = (m < Mrows && hi < g.H && wi < g.W) ? (n * g.H + hi) * g.W + wi : -1;
    }
    // slice counters of the NEXT load: channel chunk cc, tap (r, s) (dgrad: (jr, js) of the class); advanced once per
    // slice instead of dividing the slice index every step
    int l_cc = 0, l_r = 0, l_s = 0;
    if (T > 0) {
        l_cc = s_beg % cchunks;
        const int rs = s_beg / cchunks;
        l_r = rs / Sc;
        l_s = rs - l_r * Sc;
    }
    // element offset of each staging row's pixel at tap (0,0) (32-bit: tensors are checked to stay below 4 GiB)
    int apix[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
        apix[i] = MODE == 0 ? (an[i] * g.H + ah[i]) * g.W + aw[i] : (an[i] * g.Ho + ah[i]) * g.Wo + aw[i];
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(a_src, (unsigned)(MODE == 0 ? g.N * g.H * g.W * g.C : g.N * g.Ho * g.Wo * g.K) * 4u);
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(wgt, (unsigned)(g.K * g.R * g.S * g.C) * 4u);
    auto load_a = [&](float4 *ra) {
        const int cc = l_cc, r = l_r, s = l_s;
        const int tapoff = MODE == 0 ? r * g.W + s : -(r * g.Wo + s);
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            bool ok;
            unsigned off;
            if (MODE == 0) {
                ok = aval[i] && (unsigned)(ah[i] + r) < (unsigned)g.H && (unsigned)(aw[i] + s) < (unsigned)g.W && cc * BKT + c4 * 4 < g.C;
                off = (unsigned)((apix[i] + tapoff) * g.C + cc * BKT + c4 * 4) * 4u;
            } else {
                ok = aval[i] && (unsigned)(ah[i] - r) < (unsigned)g.Ho && (unsigned)(aw[i] - s) < (unsigned)g.Wo && cc * BKT + c4 * 4 < g.K;
                off = (unsigned)((apix[i] + tapoff) * g.K + cc * BKT + c4 * 4) * 4u;
            }
            ra[i] = buf_load4(a_rsrc, ok ? off : 0xffffffffu);
        }
    };
    // ---- B staging.  forward: rows = k, 16 consecutive c of filter tap (r,s): float4 copies.
    //      dgrad: rows = c, 16 k's strided by R*S*C: read float4 along c, transpose into LDS.
    constexpr int B_F4 = (BN * NQ + NT - 1) / NT;
    constexpr bool BDW = MODE == 1 && BF && (BN * BKT) % (NT * 8) == 0;      // dgrad, bf16 operands: B staged as groups of 8 k (see load_b)
    auto load_b = [&](float4 *rb) {
        const int cc = l_cc;
        const int rs = MODE == 0 ? l_r * g.S + l_s : (r0 + g.stride * l_r) * g.S + s0 + g.stride * l_s;   // filter tap of the slice
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int idx = t + NT * i;
            bool ok;
            unsigned off;
            if (BDW) {
                // bf16 operands: a thread owns 8 consecutive k of one in-channel (rb[2j], rb[2j+1]) — dword loads, lanes along c
                // (256 contiguous bytes per wave and load), so that the packed terms go to LDS as one 16-byte write each
                const int gidx = t + NT * (i >> 1), cl = gidx % BN, kg = gidx / BN;
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = cc * BKT + kg * 8 + (i & 1) * 4 + j;
                    const bool okk = kg * 8 < BKT && n0 + cl < g.C && k < g.K;
                    const unsigned o = (unsigned)((k * g.R * g.S + rs) * g.C + n0 + cl) * 4u;
                    e[j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(w_rsrc, okk ? o : 0xffffffffu, 0, 0));
                }
                rb[i] = make_float4(e[0], e[1], e[2], e[3]);
                continue;
            }
            if (MODE == 0) {
                const int row = srow(idx / NQ), q4 = idx % NQ;           // row = out channel, q4 = float4 along c
                ok = row < BN && n0 + row < g.K && cc * BKT + q4 * 4 < g.C;
                off = (unsigned)(((n0 + row) * g.R * g.S + rs) * g.C + cc * BKT + q4 * 4) * 4u;
            } else {
                const int kk = idx % BKT, cq = idx / BKT;                // kk = k inside the slice, cq = float4 of in-channels
                ok = cq * 4 < BN && n0 + cq * 4 < g.C && cc * BKT + kk < g.K;
                off = (unsigned)(((cc * BKT + kk) * g.R * g.S + rs) * g.C + n0 + cq * 4) * 4u;
            }
            rb[i] = buf_load4(w_rsrc, ok ? off : 0xffffffffu);
        }
    };
    auto store_ab = [&](int buf, const float4 *ra, const float4 *rb) {
        if (BF) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const SplitT<NTERM> sp = split_terms<NTERM, H2>(ra[i], sca);
#pragma unroll
                for (int tm = 0; tm < NTERM; ++tm) *reinterpret_cast<uint2 *>(&Ah[buf][tm][arow + RPP * i][c4 * 4]) = sp.t[tm];
            }
            if (BDW) {
#pragma unroll
                for (int i2 = 0; i2 < B_F4 / 2; ++i2) {
                    const int gidx = t + NT * i2, cl = gidx % BN, kg = gidx / BN;
                    const SplitT<NTERM> s0 = split_terms<NTERM, H2>(rb[2 * i2], scb), s1 = split_terms<NTERM, H2>(rb[2 * i2 + 1], scb);
                    if (kg * 8 < BKT) {
#pragma unroll
                        for (int tm = 0; tm < NTERM; ++tm) {
                            u32x4 v;
                            v.x = s0.t[tm].x; v.y = s0.t[tm].y; v.z = s1.t[tm].x; v.w = s1.t[tm].y;
                            *reinterpret_cast<u32x4 *>(&Bh[buf][tm][cl][kg * 8]) = v;
                        }
                    }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int idx = t + NT * i;
                const SplitT<NTERM> sp = split_terms<NTERM, H2>(rb[i], scb);
                if (MODE == 0) {
                    const int row = srow(idx / NQ), q4 = idx % NQ;
                    if (row < BN) {
#pragma unroll
                        for (int tm = 0; tm < NTERM; ++tm) *reinterpret_cast<uint2 *>(&Bh[buf][tm][row][q4 * 4]) = sp.t[tm];
                    }
                } else {
                    const int kk = idx % BKT, cq = idx / BKT;
                    if (cq * 4 < BN) {
#pragma unroll
                        for (int tm = 0; tm < NTERM; ++tm) {
                            Bh[buf][tm][cq * 4 + 0][kk] = (unsigned short)(sp.t[tm].x & 0xffffu);
                            Bh[buf][tm][cq * 4 + 1][kk] = (unsigned short)(sp.t[tm].x >> 16);
                            Bh[buf][tm][cq * 4 + 2][kk] = (unsigned short)(sp.t[tm].y & 0xffffu);
                            Bh[buf][tm][cq * 4 + 3][kk] = (unsigned short)(sp.t[tm].y >> 16);
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4 *>(&As[buf][arow + RPP * i][c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int idx = t + NT * i;
            if (MODE == 0) {
                const int row = idx / NQ, q4 = idx % NQ;
                if (row < BN) *reinterpret_cast<float4 *>(&Bs[buf][row][q4 * 4]) = rb[i];
            } else {
                const int kk = idx % BKT, cq = idx / BKT;
                if (cq * 4 < BN) {
                    Bs[buf][cq * 4 + 0][kk] = rb[i].x;
                    Bs[buf][cq * 4 + 1][kk] = rb[i].y;
                    Bs[buf][cq * 4 + 2][kk] = rb[i].z;
                    Bs[buf][cq * 4 + 3][kk] = rb[i].w;
                }
            }
        }
    };

    if (MODE == 1 && Tall == 0) {               // stride > filter extent: this class receives no gradient
        __syncthreads();
        float *dst = out + (size_t)blockIdx.z * g.N * g.H * g.W * Ncols;     // split-K: every partial slot gets its zeros
        constexpr int QN = BN / 4;                  // float4 per tile row
        unsigned am0 = 0u;
        for (int idx = t; idx < BM * QN; idx += NT) {
            const int ml = idx / QN, c = n0 + (idx % QN) * 4;
            const int px = rowpix[ml];
            if (px >= 0 && c < Ncols) {
                const float4 v = (bias && zsplits == 1) ? *reinterpret_cast<const float4 *>(bias + (size_t)px * Ncols + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(dst + (size_t)px * Ncols + c) = v;
                am0 = max(max(am0, abs_bits(v.x)), max(abs_bits(v.y), max(abs_bits(v.z), abs_bits(v.w))));
            }
        }
        if (sc.amax_out != nullptr && zsplits == 1) amax_commit(am0, sc.amax_out);
        return;
    }
    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // (a second register stage — loads issued two slices ahead — was measured and is slower: +3 % fwd, +6 % dgrad over the
    // config-B layers; HBM latency is not what limits this loop)
    float4 ra[A_F4], rb[B_F4];
    auto advance = [&]() {
        if (++l_cc == cchunks) {
            l_cc = 0;
            if (++l_s == Sc) { l_s = 0; ++l_r; }
        }
    };
    if (T > 0) {
        load_a(ra);
        load_b(rb);
        advance();
        store_ab(0, ra, rb);
    }
    __syncthreads();
    const int row = lane & 31, h = lane >> 5;
    for (int step = 0; step < T; ++step) {
        const int cur = NBUF == 2 ? (step & 1) : 0;
        if (step + 1 < T) {                       // prefetch the next slice into registers
            load_a(ra);
            load_b(rb);
            advance();
        }
        // half-wave h owns reduction elements [h*BKT/2, (h+1)*BKT/2) of the slice, 8 at a time
#pragma unroll
        for (int part = 0; part < BKT / 16; ++part) {
            const int e0 = (BKT / 2) * h + 8 * part;
            if (BF) {
                // one 32x32x16 bf16 / fp16 MFMA consumes the 8 elements of both half-waves; 6 (3) term pairs, smallest first
                u32x4 ah[WTM][NTERM], bh[WTN][NTERM];
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int tm = 0; tm < NTERM; ++tm) ah[i][tm] = *reinterpret_cast<const u32x4 *>(&Ah[cur][tm][wm0 + i * 32 + row][e0]);
#pragma unroll
                for (int j = 0; j < WTN; ++j)
#pragma unroll
                    for (int tm = 0; tm < NTERM; ++tm) bh[j][tm] = *reinterpret_cast<const u32x4 *>(&Bh[cur][tm][wn0 + j * 32 + row][e0]);
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) {
                        acc[i][j] = mma_terms<NTERM, H2>(ah[i], bh[j], acc[i][j]);
                    }
                continue;
            }
            float af[WTM][8], bf[WTN][8];
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const float4 v0 = *reinterpret_cast<const float4 *>(&As[cur][wm0 + i * 32 + row][e0]);
                const float4 v1 = *reinterpret_cast<const float4 *>(&As[cur][wm0 + i * 32 + row][e0 + 4]);
                af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
                af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const float4 v0 = *reinterpret_cast<const float4 *>(&Bs[cur][wn0 + j * 32 + row][e0]);
                const float4 v1 = *reinterpret_cast<const float4 *>(&Bs[cur][wn0 + j * 32 + row][e0 + 4]);
                bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
                bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) acc[i][j] = mfma32(af[i][e], bf[j][e], acc[i][j]);
        }
        if (NBUF == 1) __syncthreads();           // single buffer: everyone has read the slice before it is overwritten
        if (step + 1 < T) store_ab(NBUF == 2 ? (cur ^ 1) : 0, ra, rb);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    // forward with `stats`: per-channel (sum, sum of squares) of this tile's valid rows, i.e. the partials BatchNorm's first
    // pass would otherwise re-read the whole output for (stats [tiles_m][K][2], same layout as bn_reduce_kernel's)
    // data gradient with `stats` + bnb: per-channel (sum dz, sum dz * xhat) of the tile — the partials of the BatchNorm backward whose
    // output this convolution differentiates (same layout; rows = M-tiles x stride classes)
    __shared__ float sred[WGM * WGN][WTN * 32][2];
    const bool want_stats = stats != nullptr && zsplits == 1 && (MODE == 0 || bnb.x != nullptr);
    const __amdgpu_buffer_rsrc_t bx_r = make_rsrc(MODE == 1 && want_stats ? bnb.x : out, MODE == 1 && want_stats ? (unsigned)(g.N * g.H * g.W) * (unsigned)Ncols * 4u : 0u);
    const bool has_mask = MODE == 1 && want_stats && bnb.mask != nullptr;
    const __amdgpu_buffer_rsrc_t bm_r = make_rsrc(has_mask ? (const float *)bnb.mask : out, has_mask ? (unsigned)(g.N * g.H * g.W) * (unsigned)Ncols / 4u : 0u);
    // output (and the data gradient's addend) through raw buffer accesses: rows beyond the tile's valid range and columns beyond
    // the channel count get an offset beyond the descriptor's extent — the store is dropped, the addend reads 0 — so the 16
    // addend loads of a tile are all in flight at once instead of one load-wait-store sequence per row
    const unsigned out_rows = MODE == 0 ? (unsigned)Mrows : (unsigned)(g.N * g.H * g.W);
    const unsigned out_bytes = out_rows * (unsigned)Ncols * 4u;
    const __amdgpu_buffer_rsrc_t dst_r = make_rsrc(out + (size_t)blockIdx.z * out_rows * Ncols, out_bytes);   // zsplits > 1: partial workspace
    const bool has_add = MODE == 1 && bias != nullptr && zsplits == 1;                   // dgrad: `bias` is the [N,H,W,C] addend
    const __amdgpu_buffer_rsrc_t add_r = make_rsrc(has_add ? bias : out, has_add ? out_bytes : 0u);
    const float isa = inv_scale_from_exp(bea), isb = inv_scale_from_exp(beb);     // f16x2: the accumulators hold s_a s_b times the result
    const bool want_amax = sc.amax_out != nullptr && zsplits == 1;                // (split plans: the sum over the splits records it)
    unsigned am = 0u;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int col = n0 + wn0 + j * 32 + row;
        float s1 = 0.f, s2 = 0.f;
        const bool cv = col < Ncols;
        const float bv = (MODE == 0 && bias && zsplits == 1 && cv) ? bias[col] : 0.f;
        const float bmu = (MODE == 1 && want_stats && cv) ? bnb.mean[col] : 0.f, brs = (MODE == 1 && want_stats && cv) ? bnb.rstd[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            unsigned off[16];
            float addv[16], bxv[16];
            unsigned bmb[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ml = wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int m = MODE == 0 ? m0 + ml : rowpix[ml];
                const bool ok = cv && (MODE == 0 ? m < Mrows : m >= 0);
                off[e] = ok ? ((unsigned)m * (unsigned)Ncols + (unsigned)col) * 4u : 0xffffffffu;
            }
            if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) addv[e] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(add_r, off[e], 0, 0));
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        bxv[e] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(bx_r, off[e], 0, 0));
                        bmb[e] = has_mask ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(bm_r, off[e] >> 4, 0, 0) & 0xffu : 0xffu;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = (H2 ? acc[i][j][e] * isa * isb : acc[i][j][e]) + bv;
                if (MODE == 1) v += addv[e];
                if (MODE == 0 && act == 1 && zsplits == 1) v = v > 0.f ? v : 0.f;
                if (MODE == 0 && act == 2 && zsplits == 1) v = v > 0.f ? v : 0.01f * v;      // LeakyReLU(0.01), nn.LeakyReLU's default slope
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), dst_r, off[e], 0, 0);
                if (want_amax) am = max(am, off[e] != 0xffffffffu ? abs_bits(v) : 0u);
                if (MODE == 0) {
                    const float vs = off[e] != 0xffffffffu ? v : 0.f;
                    s1 += vs;
                    s2 = fmaf(vs, vs, s2);
                } else if (want_stats) {
                    const float dz = off[e] != 0xffffffffu ? v * bn_bwd_act(bmb[e], col & 3, bnb.act) : 0.f;
                    s1 += dz;
                    s2 = fmaf(dz, (bxv[e] - bmu) * brs, s2);
                }
            }
        }
        if (want_stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                sred[wave][j * 32 + row][0] = s1;
                sred[wave][j * 32 + row][1] = s2;
            }
        }
    }
    if (want_amax) amax_commit(am, sc.amax_out);
    if (want_stats) {
        __syncthreads();
        if (t < BN && n0 + t < Ncols) {
            // the WGM waves stacked along M that cover column t, in wave order
            const int wn = t / (WTN * 32), cl = t - wn * (WTN * 32);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wmi = 0; wmi < WGM; ++wmi) {
                a1 += sred[wmi * WGN + wn][cl][0];
                a2 += sred[wmi * WGN + wn][cl][1];
            }
            float *o = stats + ((size_t)(MODE == 0 ? m0 / BM : tile_m_all) * Ncols + n0 + t) * 2;
            o[0] = a1;
            o[1] = a2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions, three-term bf16 operands ("halo" plans, bk + 2048).  The implicit-GEMM kernel above
// fetches and converts every input pixel once per filter tap and N-tile (9x for a 3x3 filter); here a workgroup owns a
// TH x 16 patch of output pixels of one image, stages the (TH+2) x 18 input patch of a 32-channel chunk in LDS ONCE — converted to
// the three bf16 planes — and all nine taps read their A fragments from it at shifted addresses.  The filter operand does not
// pass through LDS: a wave owns 32 output channels for all pixels of the patch (wave tile 32*WTM x 32), so its B fragments
// (8 consecutive reduction channels per lane) are loaded straight from L2 into registers one tap ahead and converted there —
// no barrier inside a chunk.  WK = 2 (tiles 64 channels wide): two waves share a 32-channel column and split the two 16-channel
// MFMA steps of a chunk; their accumulators are added through LDS at the end.
// MODE 0 forward (source x, reduction over c, tap (r,s) reads patch cell (ty+r, tx+s)); MODE 1 data gradient (source dy,
// reduction over k, tap (r,s) reads (ty+2-r, tx+2-s)).
// ---------------------------------------------------------------------------------------------------
template <int MODE, int WTM, int WM, int WN, int WK, int R = 3, bool H2 = false>
__global__ __launch_bounds__(WM * WN * WK * 64, ((WTM / WM >= 4 || MODE == 1 || R == 4) ? 2 : 3)) void conv3x3_halo_kernel(const float *__restrict__ a_src, const float *__restrict__ wgt,
                                                                    const float *__restrict__ bias, float *__restrict__ out, ConvGeom g,
                                                                    int act, int zsplits, float *__restrict__ stats, BnBwdSrc bnb, OpScale sc) {
    constexpr int NTM = H2 ? 2 : 3;                       // operand terms: three bf16 (truncating split) or two fp16 of the scaled operand (H2)
    constexpr int TH = 2 * WTM, TW = 16, PW = TW + R - 1, PH = TH + R - 1, HP = PH * PW;   // output patch, input patch (R x R taps)
    constexpr int RS = R * R, RING = RS % 3 == 0 ? 3 : 4;                          // filter-fragment ring: its size divides the taps of a chunk
    static_assert((R == 3 || R == 4) && (R == 3 || MODE == 0) && RS % RING == 0, "3x3, or 4x4 forward (the space-to-depth stems)");
    constexpr int NT = WM * WN * WK * 64, BN = WN * 32;
    constexpr int WS = WTM / WM;                          // 32-pixel sub-tiles per wave
    constexpr int CK = 32, LDH = CK + 8;                  // channels per chunk, pitch of a patch cell in bf16 (80 bytes)
    // Pitch of a patch ROW: a multiple of 256 bytes.  ds_read_b128 serves a wave in four groups of 16 lanes — {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} and the same two for the upper half-wave —, i.e. eight cells of fragment row m >> 4 = 0 and eight of row 1,
    // and a group is conflict-free when its 16 lanes hit 16 different 16-byte slots of the 256-byte bank row.  Cell pitch 80 bytes puts
    // column x at slot 5 x mod 16; with the rows a multiple of 256 bytes apart the two rows' column sets {0-3, 12-15} and {4-11} fill the
    // 16 slots exactly.  Round 5's dense rows (PW * 80 = 1440 bytes: slot 10 row + 5 x) put two lanes of every group on an occupied
    // slot — the 34-44 % bank-conflict cycles of profiles/r05p_conv_sq_counters.md, every instantiation of this kernel.
    constexpr int RPH = (PW * LDH * 2 + 255) / 256 * 128; // row pitch in bf16
    constexpr int HPP = (PH * RPH + LDH - 1) / LDH;       // cells' worth of LDS per plane (plane = PH rows)
    constexpr int KSW = 2 / WK;                           // 16-channel MFMA steps of a chunk per wave
    constexpr int A_ITEMS = (HP + 15) / 16 * 16 * 4, A_PASS = (A_ITEMS + NT - 1) / NT;        // item = 8 channels of one patch cell
    static_assert((WK == 1 || WK == 2) && WTM % WM == 0, "WK, WM");
    constexpr int XCH_BYTES = (WK == 2 ? WN * WTM * 16 * 64 * 4 : 0) + WM * WN * 64 * 4;      // the accumulator / statistics exchange reuses the patch planes
    constexpr int NPL = XCH_BYTES <= NTM * HPP * LDH * 2 ? NTM : 3;
    static_assert(XCH_BYTES <= NPL * HPP * LDH * 2, "the accumulator / statistics exchange reuses the patch planes");
    __shared__ __attribute__((aligned(16))) unsigned short Ah[NPL][HPP][LDH];
    const unsigned bea = H2 ? scale_exp(sc.amax_a) : 127u, beb = H2 ? scale_exp(sc.amax_b) : 127u;
    const float sca = scale_from_exp(bea), scb = scale_from_exp(beb);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wn = wave % WN, wk = (wave / WN) % WK, wmi = wave / (WN * WK);
    const int Ncols = MODE == 0 ? g.K : g.C, Cred = MODE == 0 ? g.C : g.K;
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    const int tiles_n = (Ncols + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, n0 = (blockIdx.x - tm * tiles_n) * BN;     // N-tiles of a patch are neighbours: they share it in L2
    const int txi = tm % tiles_x, t2 = tm / tiles_x, tyi = t2 % tiles_y, img = t2 / tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int cchunks = (Cred + CK - 1) / CK;
    const int c_beg = (int)((long long)cchunks * blockIdx.z / zsplits), c_end = (int)((long long)cchunks * (blockIdx.z + 1) / zsplits);

    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(a_src, (unsigned)(g.N * g.H * g.W * Cred) * 4u);
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(wgt, (unsigned)(g.K * RS * g.C) * 4u);

    // ---- A staging: thread -> (patch cell, group of 8 channels); cells of a 16-group are dealt 0,4,8,12,1,5,... so that the four
    // cells a 16-byte LDS store instruction serves per clock start 16 banks apart
    unsigned a_off[A_PASS];            // byte offset of the cell's channel 0 (0xffffffff: outside the image / no such cell)
    int a_cell[A_PASS];                // the cell's place in a patch plane (bf16 units), -1: none
#pragma unroll
    for (int i = 0; i < A_PASS; ++i) {
        const int idx = t + NT * i, praw = idx >> 2;
        const int p = (praw & ~15) | ((praw & 3) << 2) | ((praw >> 2) & 3);
        const int hy = p / PW, hx = p - hy * PW;
        const int iy = y0 - g.pad + hy, ix = x0 - g.pad + hx;
        const bool ok = idx < A_ITEMS && p < HP && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        a_cell[i] = (idx < A_ITEMS && p < HP) ? hy * RPH + hx * LDH : -1;
        a_off[i] = ok ? (unsigned)(((img * g.H + iy) * g.W + ix) * Cred) * 4u : 0xffffffffu;
    }
    const int c8 = (t & 3) * 8;        // NT is a multiple of 4: the channel group does not depend on the pass
    float4 ra[A_PASS][2];
    auto load_a = [&](int cc) {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            const int c = cc * CK + c8;
            const bool in = a_off[i] != 0xffffffffu;
            ra[i][0] = buf_load4(a_rsrc, (in && c < Cred) ? a_off[i] + (unsigned)c * 4u : 0xffffffffu);
            ra[i][1] = buf_load4(a_rsrc, (in && c + 4 < Cred) ? a_off[i] + (unsigned)(c + 4) * 4u : 0xffffffffu);
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < A_PASS; ++i) {
            if (a_cell[i] < 0) continue;
            const SplitT<NTM> s0 = split_terms<NTM, H2>(ra[i][0], sca), s1 = split_terms<NTM, H2>(ra[i][1], sca);
#pragma unroll
            for (int tmn = 0; tmn < NTM; ++tmn) {
                u32x4 v;
                v.x = s0.t[tmn].x; v.y = s0.t[tmn].y; v.z = s1.t[tmn].x; v.w = s1.t[tmn].y;
                *reinterpret_cast<u32x4 *>(&Ah[tmn][0][0] + a_cell[i] + c8) = v;
            }
        }
    };

    // ---- B fragments: lane = (column n = lane & 31, reduction half kg = lane >> 5): 8 consecutive reduction channels.  The
    // per-lane part of the address is formed once; chunk, tap and step enter as a scalar offset.
    const int col = n0 + wn * 32 + (lane & 31), kg = lane >> 5;
    const bool colv = col < Ncols;
    const unsigned b_lane = !colv ? 0xffffffffu : MODE == 0 ? (unsigned)(col * RS * g.C + kg * 8) * 4u : (unsigned)(kg * 8 * RS * g.C + col) * 4u;
    // Fragments are fetched two taps ahead into a ring of three register sets (a tap is 12-48 MFMAs, 0.2-0.7 us: one tap ahead
    // leaves most of an L2 round trip exposed — with the loads removed the config-B layers ran 25 % faster; a whole filter row ahead
    // costs 40-70 more registers and was 3 % slower).
    float rb[RING][KSW][8];
    auto load_b = [&](float (&dst)[KSW][8], int cc, int rs) {
#pragma unroll
        for (int q = 0; q < KSW; ++q) {
            const int ks = WK == 2 ? wk : q;
            // (the scalar offset does not take part in the descriptor's range check: channels beyond the last one — only in a
            // partial last chunk — are masked per lane; elsewhere the A planes' zeros meet finite filter values)
            const int c_lane = cc * CK + ks * 16 + kg * 8;
            if (MODE == 0) {
                const int so = __builtin_amdgcn_readfirstlane((rs * g.C + cc * CK + ks * 16) * 4);
                const i32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, c_lane < Cred ? b_lane : 0xffffffffu, so, 0);
                const i32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, c_lane + 4 < Cred ? b_lane : 0xffffffffu, so + 16, 0);
                dst[q][0] = __int_as_float(v0.x); dst[q][1] = __int_as_float(v0.y); dst[q][2] = __int_as_float(v0.z); dst[q][3] = __int_as_float(v0.w);
                dst[q][4] = __int_as_float(v1.x); dst[q][5] = __int_as_float(v1.y); dst[q][6] = __int_as_float(v1.z); dst[q][7] = __int_as_float(v1.w);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int so = __builtin_amdgcn_readfirstlane((((cc * CK + ks * 16 + j) * RS + rs) * g.C) * 4);
                    dst[q][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(w_rsrc, c_lane + j < Cred ? b_lane : 0xffffffffu, so, 0));
                }
            }
        }
    };

    f32x16 acc[WS];
#pragma unroll
    for (int i = 0; i < WS; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // A fragment of row m = lane & 31 of sub-tile i: patch cell (2i + (m >> 4) + dy, (m & 15) + dx), channels ks*16 + kg*8 ..
    const int frag_base = (2 * WS * wmi + ((lane & 31) >> 4)) * RPH + (lane & 15) * LDH + kg * 8;       // in bf16 units

    if (c_beg < c_end) {
        load_a(c_beg);
        load_b(rb[0], c_beg, 0);
        load_b(rb[1], c_beg, 1);
    }
    for (int cc = c_beg; cc < c_end; ++cc) {
        __syncthreads();                                  // the previous chunk's fragments have been read
        store_a();
        __syncthreads();
        if (cc + 1 < c_end) load_a(cc + 1);               // in flight during the nine taps
#pragma unroll
        for (int rs = 0; rs < RS; ++rs) {
            const int r = rs / R, s = rs - R * r;
            u32x4 bh[KSW][NTM];
#pragma unroll
            for (int q = 0; q < KSW; ++q) {
                const SplitT<NTM> s0 = split_terms<NTM, H2>(make_float4(rb[rs % RING][q][0], rb[rs % RING][q][1], rb[rs % RING][q][2], rb[rs % RING][q][3]), scb);
                const SplitT<NTM> s1 = split_terms<NTM, H2>(make_float4(rb[rs % RING][q][4], rb[rs % RING][q][5], rb[rs % RING][q][6], rb[rs % RING][q][7]), scb);
#pragma unroll
                for (int tmn = 0; tmn < NTM; ++tmn) {
                    bh[q][tmn].x = s0.t[tmn].x; bh[q][tmn].y = s0.t[tmn].y; bh[q][tmn].z = s1.t[tmn].x; bh[q][tmn].w = s1.t[tmn].y;
                }
            }
            if (rs + 2 < RS) load_b(rb[(rs + 2) % RING], cc, rs + 2);
            else if (cc + 1 < c_end) load_b(rb[(rs + 2) % RING], cc + 1, rs + 2 - RS);
            const int dy = MODE == 0 ? r : R - 1 - r, dx = MODE == 0 ? s : R - 1 - s;
            const unsigned short *tap = &Ah[0][0][0] + frag_base + dy * RPH + dx * LDH;
#pragma unroll
            for (int q = 0; q < KSW; ++q) {
                const int ks = WK == 2 ? wk : q;
                if (cc * CK + ks * 16 >= Cred) continue;      // (uniform) a step beyond the last channel: the 16-channel stems
#pragma unroll
                for (int i = 0; i < WS; ++i) {
                    const unsigned short *ap = tap + 2 * i * RPH + ks * 16;
                    u32x4 at[NTM];
#pragma unroll
                    for (int tmn = 0; tmn < NTM; ++tmn) at[tmn] = *reinterpret_cast<const u32x4 *>(ap + tmn * HPP * LDH);
                    acc[i] = mma_terms<NTM, H2>(at, bh[q], acc[i]);          // smallest terms first
                }
            }
            __builtin_amdgcn_sched_barrier(0);            // keep the taps apart: hoisting the next tap's reads costs 60+ registers
                                                          // (reading the A fragments one tap ahead by hand: +30-90 registers, spills, no gain)
        }
    }

    // ---- WK = 2: the upper waves hand their accumulators over through the (now idle) patch planes
    if (WK == 2) {
        float *xch = reinterpret_cast<float *>(&Ah[0][0][0]);
        __syncthreads();
        if (wk == 1) {
#pragma unroll
            for (int i = 0; i < WS; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) xch[(((wmi * WN + wn) * WS + i) * 16 + e) * 64 + lane] = acc[i][e];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int i = 0; i < WS; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] += xch[(((wmi * WN + wn) * WS + i) * 16 + e) * 64 + lane];
        }
    }
    const bool writer = WK == 1 || wk == 0;               // (every wave stays for the barriers below)

    // ---- epilogue (C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5))
    const int h = lane >> 5;
    const unsigned out_rows = (unsigned)(g.N * g.H * g.W);
    const unsigned out_bytes = out_rows * (unsigned)Ncols * 4u;
    const __amdgpu_buffer_rsrc_t dst_r = make_rsrc(out + (size_t)blockIdx.z * out_rows * Ncols, out_bytes);
    const bool has_add = MODE == 1 && bias != nullptr && zsplits == 1;
    const __amdgpu_buffer_rsrc_t add_r = make_rsrc(has_add ? bias : out, has_add ? out_bytes : 0u);
    const float bv = (MODE == 0 && bias && zsplits == 1 && colv) ? bias[col] : 0.f;
    const bool want_stats = stats != nullptr && zsplits == 1 && (MODE == 0 || bnb.x != nullptr);
    const bool bstats = MODE == 1 && want_stats, has_mask = bstats && bnb.mask != nullptr;
    const __amdgpu_buffer_rsrc_t bx_r = make_rsrc(bstats ? bnb.x : out, bstats ? out_bytes : 0u);
    const __amdgpu_buffer_rsrc_t bm_r = make_rsrc(has_mask ? (const float *)bnb.mask : out, has_mask ? out_rows * (unsigned)Ncols / 4u : 0u);
    const float bmu = (bstats && colv) ? bnb.mean[col] : 0.f, brs = (bstats && colv) ? bnb.rstd[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const float isa = inv_scale_from_exp(bea), isb = inv_scale_from_exp(beb);
    const bool want_amax = sc.amax_out != nullptr && zsplits == 1;
    unsigned am = 0u;
    if (writer) {
#pragma unroll
    for (int i = 0; i < WS; ++i) {
        unsigned off[16];
        float addv[16], bxv[16];
        unsigned bmb[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ml = (wmi * WS + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int y = y0 + (ml >> 4), x = x0 + (ml & 15);
            const bool ok = colv && y < g.H && x < g.W;
            off[e] = ok ? ((unsigned)((img * g.H + y) * g.W + x) * (unsigned)Ncols + (unsigned)col) * 4u : 0xffffffffu;
        }
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) addv[e] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(add_r, off[e], 0, 0));
            if (bstats) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    bxv[e] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(bx_r, off[e], 0, 0));
                    bmb[e] = has_mask ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(bm_r, off[e] >> 4, 0, 0) & 0xffu : 0xffu;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = (H2 ? acc[i][e] * isa * isb : acc[i][e]) + bv;
            if (MODE == 1) v += addv[e];
            if (MODE == 0 && act == 1 && zsplits == 1) v = v > 0.f ? v : 0.f;
            if (MODE == 0 && act == 2 && zsplits == 1) v = v > 0.f ? v : 0.01f * v;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), dst_r, off[e], 0, 0);
            if (want_amax) am = max(am, off[e] != 0xffffffffu ? abs_bits(v) : 0u);
            if (MODE == 0) {
                const float vs = off[e] != 0xffffffffu ? v : 0.f;
                s1 += vs;
                s2 = fmaf(vs, vs, s2);
            } else if (bstats) {
                const float dz = off[e] != 0xffffffffu ? v * bn_bwd_act(bmb[e], col & 3, bnb.act) : 0.f;
                s1 += dz;
                s2 = fmaf(dz, (bxv[e] - bmu) * brs, s2);
            }
        }
    }
    }
    if (want_amax) amax_commit(am, sc.amax_out);
    if (want_stats) {      // BatchNorm partials of this patch: stats[patch][channels][2] (forward: of y; data gradient: of the BatchNorm backward)
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (WM > 1) {                                         // the waves stacked along the pixels add up in wave order
            float *sx = reinterpret_cast<float *>(&Ah[0][0][0]) + (WK == 2 ? WM * WN * WS * 16 * 64 : 0);
            __syncthreads();                                  // the patch planes are idle (WK == 1: all fragments have been read)
            if (writer && h == 0) {
                sx[((wmi * WN + wn) * 32 + (lane & 31)) * 2] = s1;
                sx[((wmi * WN + wn) * 32 + (lane & 31)) * 2 + 1] = s2;
            }
            __syncthreads();
            s1 = 0.f; s2 = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WM; ++w2) {
                s1 += sx[((w2 * WN + wn) * 32 + (lane & 31)) * 2];
                s2 += sx[((w2 * WN + wn) * 32 + (lane & 31)) * 2 + 1];
            }
        }
        if (writer && wmi == 0 && h == 0 && colv) {
            float *o = stats + ((size_t)tm * Ncols + col) * 2;
            o[0] = s1;
            o[1] = s2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad: one workgroup per (k-tile, c-tile, pixel split, filter tap).  Both operands are transposed on
// their way into LDS (rows = channels, 16 pixels of reduction per slice).
// part[split][k][r][s][c] partial sums; a second kernel adds the splits (deterministic).
// ---------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                         float *__restrict__ part, ConvGeom g, int px_per_split) {
    constexpr int WTM = BM / 2 / 32, WTN = BN / 2 / 32;       // 2x2 waves
    static_assert(WTM >= 1 && WTN >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDP];     // [k][pixel]
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDP];     // [c][pixel]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm0 = (wave >> 1) * (WTM * 32), wn0 = (wave & 1) * (WTN * 32);
    const int ktiles = (g.K + BM - 1) / BM;
    const int k0 = (blockIdx.x % ktiles) * BM, c0 = (blockIdx.x / ktiles) * BN;
    const int split = blockIdx.y, rs = blockIdx.z;
    const int r = rs / g.S, s = rs - r * g.S;
    const int M = g.N * g.Ho * g.Wo;
    const int mbeg = split * px_per_split, mend = min(M, mbeg + px_per_split);
    const int T = (mend - mbeg + BK - 1) / BK;

    // staging: thread -> pixel pp = t & 15 of the slice, float4 group q = t >> 4 (16 groups = 64 channels per pass)
    const int pp = t & 15, q = t >> 4;
    constexpr int A_P = (BM + 63) / 64, B_P = (BN + 63) / 64;
    const __amdgpu_buffer_rsrc_t dy_rsrc = make_rsrc(dy, (unsigned)(g.N * g.Ho * g.Wo * g.K) * 4u);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(x, (unsigned)(g.N * g.H * g.W * g.C) * 4u);
    auto load = [&](int step, float4 *ra, float4 *rb) {
        const int m = mbeg + step * BK + pp;
        const bool mv = m < mend;
        int n = 0, hi = 0, wi = 0;
        if (mv) {
            const int wo = m % g.Wo, t2 = m / g.Wo;
            n = t2 / g.Ho;
            hi = (t2 % g.Ho) * g.stride - g.pad + r;
            wi = wo * g.stride - g.pad + s;
        }
        const bool xin = mv && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int kq = k0 + (q + 16 * i) * 4;
            const bool ok = mv && (q + 16 * i) * 4 < BM && kq < g.K;
            ra[i] = buf_load4(dy_rsrc, ok ? (unsigned)(m * g.K + kq) * 4u : 0xffffffffu);
        }
#pragma unroll
        for (int i = 0; i < B_P; ++i) {
            const int cq = c0 + (q + 16 * i) * 4;
            const bool ok = xin && (q + 16 * i) * 4 < BN && cq < g.C;
            rb[i] = buf_load4(x_rsrc, ok ? (unsigned)(((n * g.H + hi) * g.W + wi) * g.C + cq) * 4u : 0xffffffffu);
        }
    };
    auto store = [&](int buf, const float4 *ra, const float4 *rb) {
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int rr = (q + 16 * i) * 4;
            if (rr < BM) {
                As[buf][rr + 0][pp] = ra[i].x; As[buf][rr + 1][pp] = ra[i].y;
                As[buf][rr + 2][pp] = ra[i].z; As[buf][rr + 3][pp] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < B_P; ++i) {
            const int rr = (q + 16 * i) * 4;
            if (rr < BN) {
                Bs[buf][rr + 0][pp] = rb[i].x; Bs[buf][rr + 1][pp] = rb[i].y;
                Bs[buf][rr + 2][pp] = rb[i].z; Bs[buf][rr + 3][pp] = rb[i].w;
            }
        }
    };
    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float4 ra[A_P], rb[B_P];
    const int row = lane & 31, h = lane >> 5;
    if (T > 0) {
        load(0, ra, rb);
        store(0, ra, rb);
    }
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int cur = step & 1;
        if (step + 1 < T) load(step + 1, ra, rb);
        float af[WTM][8], bf[WTN][8];
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            const float4 v0 = *reinterpret_cast<const float4 *>(&As[cur][wm0 + i * 32 + row][8 * h]);
            const float4 v1 = *reinterpret_cast<const float4 *>(&As[cur][wm0 + i * 32 + row][8 * h + 4]);
            af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
            af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const float4 v0 = *reinterpret_cast<const float4 *>(&Bs[cur][wn0 + j * 32 + row][8 * h]);
            const float4 v1 = *reinterpret_cast<const float4 *>(&Bs[cur][wn0 + j * 32 + row][8 * h + 4]);
            bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
            bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) acc[i][j] = mfma32(af[i][e], bf[j][e], acc[i][j]);
        if (step + 1 < T) store(cur ^ 1, ra, rb);
        __syncthreads();
    }
    float *po = part + (size_t)split * g.K * g.R * g.S * g.C;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int c = c0 + wn0 + j * 32 + row;
        if (c >= g.C) continue;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (k < g.K) po[((size_t)k * g.R * g.S + rs) * g.C + c] = acc[i][j][e];
            }
    }
}

// out[i] = sum over splits of part[s][i].  A block owns 16 float4 columns; its 16 thread groups each add the
// splits s = g, g+16, ... in order, then a fixed-order tree over the groups: deterministic for a given plan.
__global__ __launch_bounds__(256) void split_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, size_t n,
                                                           int splits) {
    __shared__ float4 red[16][16];
    sqd::split_reduce_block(part, out, n, splits, blockIdx.x, red);
}
// ... and a second sum over the same splits in the same launch (a weight gradient's filter and bias partials): blocks [0, nb1) own the
// first, the others the second — the arithmetic of two split_reduce_kernel launches
__global__ __launch_bounds__(256) void split_reduce2_kernel(const float *__restrict__ part, float *__restrict__ out, size_t n,
                                                            const float *__restrict__ part2, float *__restrict__ out2, size_t n2, int splits,
                                                            int nb1) {
    __shared__ float4 red[16][16];
    if ((int)blockIdx.x < nb1) sqd::split_reduce_block(part, out, n, splits, blockIdx.x, red);       // (workgroup-uniform)
    else sqd::split_reduce_block(part2, out2, n2, splits, (int)blockIdx.x - nb1, red);
}

// ---------------------------------------------------------------------------------------------------
// wgrad, operands straight from memory (no LDS staging): v_mfma_f32_16x16x4_f32 takes A[i][p] from lane
// (i = lane%16, p = lane/16) and B[p][j] likewise, and with channels-last tensors "channel i of pixel p" for
// 16 x 4 lanes is 4 contiguous 64-byte runs of dy (resp. x) — a coalesced load IS the operand fetch.
// A lane loads KT (CT) consecutive channels with one dwordxKT load and feeds KT x CT interleaved 16x16 tiles:
//   tile (q, cq) holds dW[k0 + KT*i + q][c0 + CT*j + cq].
// One wave = one (k-group, c-group, tap, pixel range); the 4 waves of a workgroup take 4 pixel ranges and add
// their accumulators through LDS before writing one partial (part[split][k][r][s][c]).
// ---------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV> struct VecLoad;
template <> struct VecLoad<1> {
    static __device__ __forceinline__ void ld(const float *p, float *v) { v[0] = *p; }
};
template <> struct VecLoad<2> {
    static __device__ __forceinline__ void ld(const float *p, float *v) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <> struct VecLoad<4> {
    static __device__ __forceinline__ void ld(const float *p, float *v) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

// TP = filter taps per wave: 1, or the S = 3 taps of one filter row — the wave then fetches dy once for three
// products (x at wi-1, wi, wi+1: the neighbouring pixel's channels, C floats further) and issues 3x the MFMAs per
// step of address arithmetic.
__device__ __forceinline__ void ldv(const float *__restrict__ base, unsigned byte_off, float (&v)[1]) {
    v[0] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void ldv(const float *__restrict__ base, unsigned byte_off, float (&v)[2]) {
    const float2 t = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + byte_off);
    v[0] = t.x; v[1] = t.y;
}
__device__ __forceinline__ void ldv(const float *__restrict__ base, unsigned byte_off, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

// the same fetches as raw buffer loads: an offset of 0xffffffff (pixel beyond the range, padding tap) returns zeros, no branch
__device__ __forceinline__ void ldb(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float (&v)[1]) {
    v[0] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void ldb(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float (&v)[2]) {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
    v[0] = __int_as_float(t.x); v[1] = __int_as_float(t.y);
}
__device__ __forceinline__ void ldb(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float (&v)[4]) {
    const float4 t = buf_load4(r, byte_off);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

template <int KT, int CT, int TP>
__global__ __launch_bounds__(256) void conv_wgrad_direct_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                float *__restrict__ part, ConvGeom g, int px_per_wave, int kgroups,
                                                                int nsplits, float *__restrict__ bias_part) {
    constexpr int UB = TP == 3 ? 2 : 4;                     // MFMA steps (of 4 pixels) per load batch
    constexpr int NT = KT * CT;
    __shared__ float red[4][NT * 4][64];                    // cross-wave add, one tap at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order: tap groups fastest, then channel groups, then pixel ranges — the taps of a 3x3 filter read the
    // same dy rows and overlapping x rows out of one L2
    const int tgroups = g.R * g.S / TP, groups = kgroups * (g.C / (16 * CT));
    const int logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (logical >= tgroups * groups * nsplits) return;
    const int tg = logical % tgroups, lg = logical / tgroups, grp = lg % groups, split = lg / groups;
    const int kg = grp % kgroups, cg = grp / kgroups;
    const int rs0 = tg * TP, r = rs0 / g.S, s0 = rs0 - r * g.S;
    const int M = g.N * g.Ho * g.Wo;
    const int i16 = lane & 15, pq = lane >> 4;
    const int kl = kg * 16 * KT + KT * i16, cl = cg * 16 * CT + CT * i16;
    const int mbeg = (split * 4 + wave) * px_per_wave;
    const int mend = min(M, mbeg + px_per_wave);
    // this lane's pixel: m = mbeg + pq, advancing by 4 per step; (n, ho, wo) and the byte offsets kept incrementally
    int m = mbeg + pq;
    int wo = m % g.Wo, t2 = m / g.Wo, ho = t2 % g.Ho, n = t2 / g.Ho;
    unsigned dyoff = ((unsigned)m * g.K + kl) * 4u;
    const unsigned dystep = 4u * g.K * 4u;
    int hi = ho * g.stride - g.pad + r;
    unsigned xrow = ((unsigned)(n * g.H + hi) * g.W) * (unsigned)g.C;      // element offset of (n, hi, 0, 0); unused when hi is outside

    f32x4 acc[TP][KT][CT];
#pragma unroll
    for (int t = 0; t < TP; ++t)
#pragma unroll
        for (int q = 0; q < KT; ++q)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[t][q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of dy: the workgroups of the first channel group / tap group add up the dy values they
    // fetch anyway (bias_part [nsplits][K], summed over the splits by the caller)
    const bool do_bias = bias_part != nullptr && cg == 0 && tg == 0;
    float bsum[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) bsum[q] = 0.f;

    const __amdgpu_buffer_rsrc_t dy_rsrc = make_rsrc(dy, (unsigned)(g.N * g.Ho * g.Wo * g.K) * 4u);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(x, (unsigned)(g.N * g.H * g.W * g.C) * 4u);
    auto load_batch = [&](float (*a)[KT], float (*b)[TP][CT]) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const bool mv = m < mend;
            const bool hv = mv && (unsigned)hi < (unsigned)g.H;
            const int wi0 = wo * g.stride - g.pad + s0;
            ldb(dy_rsrc, mv ? dyoff : 0xffffffffu, a[u]);
#pragma unroll
            for (int t = 0; t < TP; ++t)
                ldb(x_rsrc, (hv && (unsigned)(wi0 + t) < (unsigned)g.W) ? (xrow + (unsigned)(wi0 + t) * g.C + cl) * 4u : 0xffffffffu, b[u][t]);
            m += 4;
            dyoff += dystep;
            wo += 4;
            while (wo >= g.Wo) {                            // next output row (twice when Wo < 4)
                wo -= g.Wo;
                if (++ho >= g.Ho) { ho = 0; ++n; }
                hi = ho * g.stride - g.pad + r;
                xrow = ((unsigned)(n * g.H + hi) * g.W) * (unsigned)g.C;
            }
        }
    };
    auto mma_batch = [&](float (*a)[KT], float (*b)[TP][CT]) {
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int t = 0; t < TP; ++t)
#pragma unroll
                for (int q = 0; q < KT; ++q)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[t][q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][q], b[u][t][c], acc[t][q][c], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int q = 0; q < KT; ++q) bsum[q] += a[u][q];
        }
    };
    float a0[UB][KT], b0[UB][TP][CT], a1[UB][KT], b1[UB][TP][CT];
    const int nb = (mend - mbeg + 4 * UB - 1) / (4 * UB);    // batches (loads past mend are masked to zero)
    if (nb > 0) load_batch(a0, b0);
    for (int bi = 0; bi < nb; bi += 2) {
        load_batch(a1, b1);
        mma_batch(a0, b0);
        load_batch(a0, b0);
        mma_batch(a1, b1);
    }
    // ---- add the 4 waves' accumulators (fixed order), wave w writes quarter w of the tile registers; one tap per round
    // (a slot per wave: a single shared image with the waves adding in turn needs 4x less LDS but serialises the
    // tail behind 4 barriers — measured 9 % slower over the config-B layers)
    float *po = part + (size_t)split * g.K * g.R * g.S * g.C;
#pragma unroll
    for (int t = 0; t < TP; ++t) {
        if (t > 0) __syncthreads();
#pragma unroll
        for (int q = 0; q < KT; ++q)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[wave][(q * CT + c) * 4 + v][lane] = acc[t][q][c][v];
        __syncthreads();
        for (int idx = wave; idx < NT * 4; idx += 4) {
            const float sum = ((red[0][idx][lane] + red[1][idx][lane]) + red[2][idx][lane]) + red[3][idx][lane];
            const int v = idx & 3, c = (idx >> 2) % CT, q = (idx >> 2) / CT;
            const int k = kg * 16 * KT + KT * (4 * pq + v) + q, cc = cg * 16 * CT + CT * i16 + c;
            po[((size_t)k * g.R * g.S + rs0 + t) * g.C + cc] = sum;
        }
    }
    if (do_bias) {                                           // workgroup-uniform
        __shared__ float bred[4][16 * KT];
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            float v = bsum[q];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (pq == 0) bred[wave][i16 * KT + q] = v;
        }
        __syncthreads();
        if (threadIdx.x < 16 * KT)
            bias_part[(size_t)split * g.K + kg * 16 * KT + threadIdx.x] =
                ((bred[0][threadIdx.x] + bred[1][threadIdx.x]) + bred[2][threadIdx.x]) + bred[3][threadIdx.x];
    }
}


// ---------------------------------------------------------------------------------------------------
// wgrad, shared-operand kernel: the four waves of a workgroup own four register tiles of ONE (WK*KT*16) x (WC*CT*16) block of
// dw[., tap, .] and walk the same pixel range together; the dy / x rows of 32 pixels are staged once per workgroup in LDS in
// their memory order ([pixel][channel]: the 16x16x4 fp32 MFMA takes one value per lane, rows = channels, k = pixel, so a lane's
// KT (CT) consecutive channels are one 16-byte LDS read and nothing is transposed).  The direct-operand kernel above gives every
// wave its own 64 x 64 tile and pixel range, i.e. its own 512 bytes of operands per 4 pixels: at 16 B/clk/wave the L1 path, not
// the matrix core, sets its pace (43 % MFMA busy).  Here a 128 x 128 block costs 1 KB per 4 pixels for four waves — a quarter of
// the fetch traffic per multiply — and there is no cross-wave reduction at the end.
// part[split][k][r][s][c] partial sums as for the other kernels (split_reduce_kernel adds the splits).
// ---------------------------------------------------------------------------------------------------
template <int N> struct FVec;
template <> struct FVec<2> { typedef float2 type; };
template <> struct FVec<4> { typedef float4 type; };
template <int WK, int WC, int KT, int CT>
__global__ __launch_bounds__(256) void conv_wgrad_shared_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                float *__restrict__ part, ConvGeom g, int px_per_split, int nsplits) {
    static_assert(WK * WC == 4 && (KT == 2 || KT == 4) && (CT == 2 || CT == 4), "4 waves, register tiles of 32 / 64 channels");
    constexpr int TK = WK * KT * 16, TC = WC * CT * 16, PS = 32;    // block of filters x channels, pixels per slab
    constexpr int CHA = TK / 4, CHB = TC / 4;                       // float4 per staged pixel row
    constexpr int PPA = 256 / CHA, PPB = 256 / CHB;                 // pixel rows per staging pass
    constexpr int NPA = PS / PPA, NPB = PS / PPB;                   // passes per slab
    static_assert(NPA >= 1 && NPB >= 1, "tile too narrow for 256 staging threads");
    __shared__ __attribute__((aligned(16))) float sA[PS][TK];
    __shared__ __attribute__((aligned(16))) float sB[PS][TC];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wk = wave / WC, wc = wave % WC;
    const int i16 = lane & 15, pq = lane >> 4;
    const int RS = g.R * g.S, tk = g.K / TK, tc = g.C / TC;
    const int logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // taps fastest inside an XCD's range
    if (logical >= RS * tk * tc * nsplits) return;
    const int tap = logical % RS, lg = logical / RS, grp = lg % (tk * tc), split = lg / (tk * tc);
    const int k0 = (grp % tk) * TK, c0 = (grp / tk) * TC, r = tap / g.S, s = tap - r * g.S;
    const int M = g.N * g.Ho * g.Wo;
    const int mbeg = split * px_per_split, mend = min(M, mbeg + px_per_split);
    const int chA = t % CHA, ppA = t / CHA, chB = t % CHB, ppB = t / CHB;
    // x rows: (n, ho, wo) of this thread's pixel of every pass, advanced by PS per slab
    int wo[NPB], ho[NPB], nn[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int m = mbeg + i * PPB + ppB;
        wo[i] = m % g.Wo;
        const int t2 = m / g.Wo;
        ho[i] = t2 % g.Ho;
        nn[i] = t2 / g.Ho;
    }
    const __amdgpu_buffer_rsrc_t dy_rsrc = make_rsrc(dy, (unsigned)(g.N * g.Ho * g.Wo * g.K) * 4u);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(x, (unsigned)(g.N * g.H * g.W * g.C) * 4u);
    int mslab = mbeg;
    float4 ra[NPA], rb[NPB];
    auto load = [&]() {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int m = mslab + i * PPA + ppA;
            ra[i] = buf_load4(dy_rsrc, m < mend ? ((unsigned)m * g.K + k0 + chA * 4) * 4u : 0xffffffffu);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int m = mslab + i * PPB + ppB;
            const int hi = ho[i] * g.stride - g.pad + r, wi = wo[i] * g.stride - g.pad + s;
            const bool ok = m < mend && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            rb[i] = buf_load4(x_rsrc, ok ? ((unsigned)((nn[i] * g.H + hi) * g.W + wi) * g.C + c0 + chB * 4) * 4u : 0xffffffffu);
            wo[i] += PS;
            while (wo[i] >= g.Wo) {
                wo[i] -= g.Wo;
                if (++ho[i] >= g.Ho) { ho[i] = 0; ++nn[i]; }
            }
        }
        mslab += PS;
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < NPA; ++i) *reinterpret_cast<float4 *>(&sA[i * PPA + ppA][chA * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i) *reinterpret_cast<float4 *>(&sB[i * PPB + ppB][chB * 4]) = rb[i];
    };
    f32x4 acc[KT][CT];
#pragma unroll
    for (int q = 0; q < KT; ++q)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nslab = (mend - mbeg + PS - 1) / PS;
    if (nslab > 0) {
        load();
        store();
    }
    __syncthreads();
    typedef typename FVec<KT>::type AV;
    typedef typename FVec<CT>::type BV;
    for (int sl = 0; sl < nslab; ++sl) {
        if (sl + 1 < nslab) load();                     // next slab into registers while this one is multiplied
#pragma unroll
        for (int ks = 0; ks < PS / 4; ++ks) {
            const AV av = *reinterpret_cast<const AV *>(&sA[ks * 4 + pq][wk * KT * 16 + KT * i16]);
            const BV bv = *reinterpret_cast<const BV *>(&sB[ks * 4 + pq][wc * CT * 16 + CT * i16]);
            const float *a = reinterpret_cast<const float *>(&av), *b = reinterpret_cast<const float *>(&bv);
#pragma unroll
            for (int q = 0; q < KT; ++q)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[c], acc[q][c], 0, 0, 0);
        }
        __syncthreads();
        if (sl + 1 < nslab) store();
        __syncthreads();
    }
    // C/D layout of the 16x16 MFMA: row = 4 * pq + v, column = i16; register tile q / c interleaves the channels (see the loads)
    float *po = part + (size_t)split * g.K * RS * g.C;
#pragma unroll
    for (int q = 0; q < KT; ++q)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int k = k0 + wk * KT * 16 + KT * (4 * pq + v) + q, cc = c0 + wc * CT * 16 + CT * i16;
            BV o;
            float *of = reinterpret_cast<float *>(&o);
#pragma unroll
            for (int c = 0; c < CT; ++c) of[c] = acc[q][c][v];
            *reinterpret_cast<BV *>(&po[((size_t)k * RS + tap) * g.C + cc]) = o;
        }
}

// ---------------------------------------------------------------------------------------------------
// wgrad, row-window kernel (impl 4): for the high-resolution layers with FEW channels (K = 16 KT <= 64 filters, C = 16 CT channels,
// R x R taps, stride 1) the kernels above spend their time fetching: every tap re-reads the x rows (9 - 16 times), and with 16
// filters a dy fetch feeds one or two MFMAs.  Here a workgroup holds the WHOLE dw [K][R][R][C] in the accumulators of its waves and
// walks output rows in chunks of PXW pixels: the R input rows of the window (PXW + R - 1 pixels, all channels) sit in an LDS ring of
// R + 1 slots in memory order, one new row per step (fetched into registers while the current step multiplies), the dy row in a
// double buffer; every tap is a shifted read of the same rows.  The IG x PG waves split the (tap, channel tile) items IG ways (wave
// ig owns items ig, ig + IG, ... for all KT filter tiles: per 4 pixels KT + NB LDS reads feed KT * NB MFMAs) and the 4-pixel groups
// of a step PG ways; the PG pixel groups add their accumulators through LDS once, at the end.  Row strides are = 16 (mod 32) floats
// so that the pixel groups of a read fall on disjoint banks.  Steps (image, column chunk, row) are dealt to the workgroups as equal
// contiguous ranges; part[split = workgroup][k][r][s][c] as for the other kernels.
// ---------------------------------------------------------------------------------------------------
template <int KT, int CT, int R, int IG, int PG, int PXW>
__global__ __launch_bounds__(IG * PG * 64, IG * PG == 8 ? 1 : 2) void conv_wgrad_rows_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                                             float *__restrict__ part, ConvGeom g, int steps_per_wg,
                                                                                             int nchunks, int nsplits, float *__restrict__ bias_part) {
    constexpr int NTH = IG * PG * 64, K = 16 * KT, C = 16 * CT, XW = PXW + R - 1;
    constexpr int KS = K + (K % 32 == 16 ? 0 : 16), CS = C + (C % 32 == 16 ? 0 : 16);     // LDS row strides (floats)
    constexpr int NI = CT * R * R, NB = (NI + IG - 1) / IG;                                // (tap, channel tile) items; per wave
    constexpr int XV = (XW * C / 4 + NTH - 1) / NTH, DV = (PXW * K / 4 + NTH - 1) / NTH;   // float4 fetches per thread and row
    constexpr int NKS = PXW / 4 / PG;                                                      // 4-pixel groups per wave and step
    static_assert(PXW % (4 * PG) == 0 && (IG * PG == 4 || IG * PG == 8), "wave grid");
    constexpr int RING = (R + 1) * XW * CS, DYB = 2 * PXW * KS, RED = IG * NB * KT * 256;
    constexpr int LDSF = (RING + DYB > RED ? RING + DYB : RED) + PG * K;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    float *xs = lds, *dys = lds + RING, *red = lds, *bred = lds + LDSF - PG * K;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ig = wave % IG, pg = wave / IG;
    const int i16 = lane & 15, pq = lane >> 4;
    const int split = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);             // neighbouring step ranges on one XCD
    if (split >= nsplits) return;
    const int total = g.N * nchunks * g.Ho;
    const int s_beg = split * steps_per_wg, s_end = min(total, s_beg + steps_per_wg);
    const __amdgpu_buffer_rsrc_t dy_rsrc = make_rsrc(dy, (unsigned)(g.N * g.Ho * g.Wo * g.K) * 4u);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(x, (unsigned)(g.N * g.H * g.W * g.C) * 4u);

    // this wave's items: LDS offset of (tap column s, channel tile) inside a row slot, and the tap row
    int ioff[NB], irow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int it = min(ig + IG * b, NI - 1);                // (a missing last item repeats the previous one and is not stored)
        const int ct = it % CT, tap = it / CT, r = tap / R, sx = tap - r * R;
        ioff[b] = sx * CS + ct * 16;
        irow[b] = r;
    }
    float4 rx[XV], rd[DV];
    auto fetch_x = [&](int n, int hi, int w0) {                // input row hi of image n, pixels w0 - pad .. + XW
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = t + i * NTH, px = idx / (C / 4), c4 = idx - px * (C / 4);
            const int wi = w0 - g.pad + px;
            const bool ok = idx < XW * C / 4 && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            rx[i] = buf_load4(x_rsrc, ok ? ((unsigned)((n * g.H + hi) * g.W + wi) * C + c4 * 4) * 4u : 0xffffffffu);
        }
    };
    auto store_x = [&](int slot) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int idx = t + i * NTH, px = idx / (C / 4), c4 = idx - px * (C / 4);
            if (idx < XW * C / 4) *reinterpret_cast<float4 *>(&xs[slot * (XW * CS) + px * CS + c4 * 4]) = rx[i];
        }
    };
    auto fetch_dy = [&](int n, int ho, int w0) {
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            const int idx = t + i * NTH, px = idx / (K / 4), k4 = idx - px * (K / 4);
            const bool ok = idx < PXW * K / 4 && w0 + px < g.Wo;
            rd[i] = buf_load4(dy_rsrc, ok ? ((unsigned)((n * g.Ho + ho) * g.Wo + w0 + px) * K + k4 * 4) * 4u : 0xffffffffu);
        }
    };
    auto store_dy = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            const int idx = t + i * NTH, px = idx / (K / 4), k4 = idx - px * (K / 4);
            if (idx < PXW * K / 4) *reinterpret_cast<float4 *>(&dys[buf * (PXW * KS) + px * KS + k4 * 4]) = rd[i];
        }
    };
    auto slot_of = [&](int hi) { return (hi + g.pad) % (R + 1); };     // ring slot of input row hi (hi >= -pad)

    f32x4 acc[NB][KT];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < KT; ++q) acc[b][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) bsum[q] = 0.f;

    int step = s_beg;
    while (step < s_end) {
        // ---- a column run: (n, chunk) fixed, rows ho0 .. ho1 - 1
        const int ho0 = step % g.Ho, col = step / g.Ho, chunk = col % nchunks, n = col / nchunks;
        const int ho1 = min(g.Ho, ho0 + (s_end - step));
        const int w0 = chunk * PXW;
        __syncthreads();                                       // the previous run's last step still reads the ring
        for (int r = 0; r < R; ++r) {                          // the first window, row by row through the fetch registers
            const int hi = ho0 - g.pad + r;
            fetch_x(n, hi, w0);
            if (r == 0) fetch_dy(n, ho0, w0);
            store_x(slot_of(hi));
        }
        store_dy(ho0 & 1);
        __syncthreads();
        for (int ho = ho0; ho < ho1; ++ho) {
            const bool more = ho + 1 < ho1;
            if (more) {                                        // next step's new row and dy row into registers
                fetch_x(n, ho + 1 - g.pad + R - 1, w0);
                fetch_dy(n, ho + 1, w0);
            }
            int rowbase[R];
#pragma unroll
            for (int r = 0; r < R; ++r) rowbase[r] = slot_of(ho - g.pad + r) * (XW * CS);
            int bbase[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                int rb = rowbase[0];
#pragma unroll
                for (int r = 1; r < R; ++r) rb = irow[b] == r ? rowbase[r] : rb;
                bbase[b] = rb + ioff[b] + (pg * NKS * 4 + pq) * CS + i16;
            }
            const float *dl = &dys[(ho & 1) * (PXW * KS) + (pg * NKS * 4 + pq) * KS + i16];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                float a[KT];
#pragma unroll
                for (int q = 0; q < KT; ++q) a[q] = dl[ks * 4 * KS + q * 16];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float bv = xs[bbase[b] + ks * 4 * CS];
#pragma unroll
                    for (int q = 0; q < KT; ++q) acc[b][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], bv, acc[b][q], 0, 0, 0);
                }
                if (bias_part != nullptr) {
#pragma unroll
                    for (int q = 0; q < KT; ++q) bsum[q] += a[q];
                }
                // half way through: the next step's rows have arrived; their LDS stores run under the remaining MFMAs (the slots
                // written are not read by this step: ring of R + 1, dy double buffer)
                if (ks == (NKS - 1) / 2 && more) {
                    store_x(slot_of(ho + 1 - g.pad + R - 1));
                    store_dy((ho + 1) & 1);
                }
            }
            __syncthreads();
        }
        step += ho1 - ho0;
    }
    // ---- the PG pixel groups add up (fixed order: pg = PG - 1 first) through LDS; the pg = 0 waves then hold the sums
    if (bias_part != nullptr && ig == 0) {
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            float v = bsum[q];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (pq == 0) bred[pg * K + q * 16 + i16] = v;
        }
    }
#pragma unroll 1
    for (int p = PG - 1; p >= 1; --p) {
        __syncthreads();                                       // (first round: the last step's reads of the ring are done)
        if (pg == p) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < KT; ++q)
                    *reinterpret_cast<f32x4 *>(&red[((ig * NB + b) * KT + q) * 256 + lane * 4]) = acc[b][q];
        }
        __syncthreads();
        if (pg == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < KT; ++q) acc[b][q] += *reinterpret_cast<const f32x4 *>(&red[((ig * NB + b) * KT + q) * 256 + lane * 4]);
        }
    }
    if (PG == 1) __syncthreads();
    // ---- C/D layout of the 16x16 MFMA: row (filter) = 4 * pq + v, column (channel) = i16
    if (pg == 0) {
        float *po = part + (size_t)split * K * R * R * C;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int it = ig + IG * b;
            if (it < NI) {
                const int ct = it % CT, tap = it / CT;
#pragma unroll
                for (int q = 0; q < KT; ++q)
#pragma unroll
                    for (int v = 0; v < 4; ++v) po[((size_t)(q * 16 + 4 * pq + v) * R * R + tap) * C + ct * 16 + i16] = acc[b][q][v];
            }
        }
    }
    if (bias_part != nullptr && t < K) {                       // column sums of dy (every item group read all of it: group 0 reports)
        float v = bred[t];
#pragma unroll
        for (int p = 1; p < PG; ++p) v += bred[p * K + t];
        bias_part[(size_t)split * K + t] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// wgrad on the bf16 matrix cores with three-term operands (the arithmetic of the bk + 1024 forward / data-gradient plans: every
// fp32 value as the exact sum of three bf16 terms, 6 of the 9 partial products, fp32 accumulation — fp32-level accuracy at 6/16 of
// the fp32 kernels' matrix-pipe time).  v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction elements — here: pixels — per lane,
// while memory has the channels contiguous.  So the staging turns the tile: a wavefront item is 64 channels (one per lane) x 8
// consecutive pixels, fetched as 8 dword loads of 256 contiguous bytes each; the lane converts its 8 values and writes them as
// one 16-byte LDS store per term into [term][channel][pixel] (row pitch 80 bytes: conflict-free 16-byte reads and writes).  The
// pixel arithmetic of an item (row wrap, padding, stride) is wave-uniform, i.e. scalar.  Block / wave tiling, split plan and
// partial layout as conv_wgrad_shared_kernel.
// ---------------------------------------------------------------------------------------------------
template <int WK, int WC, int WTK, int WTC>
__global__ __launch_bounds__(256) void conv_wgrad_split3_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                                float *__restrict__ part, ConvGeom g, int px_per_split, int nsplits) {
    static_assert(WK * WC == 4 && WTK >= 1 && WTK <= 2 && WTC >= 1 && WTC <= 2, "4 waves, 32 / 64 channels per wave and side");
    constexpr int TK = WK * WTK * 32, TC = WC * WTC * 32, PS = 32, LDHW = PS + 8;     // block, pixels per slab, bf16 per LDS row
    constexpr int ITEMS_A = TK / 16, ITEMS = (TK + TC) / 16, NI = ITEMS / 4;            // wave-items (64 channels x 8 pixels) per slab
    static_assert(TK % 64 == 0 && TC % 64 == 0 && ITEMS % 4 == 0, "blocks of 64 channels");
    __shared__ __attribute__((aligned(16))) unsigned short Ah[3][TK][LDHW];
    __shared__ __attribute__((aligned(16))) unsigned short Bh[3][TC][LDHW];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wk = wave / WC, wc = wave % WC;
    const int row = lane & 31, hh = lane >> 5;
    const int RS = g.R * g.S, tk = g.K / TK, tc = g.C / TC;
    const int logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (logical >= RS * tk * tc * nsplits) return;
    const int tap = logical % RS, lg = logical / RS, grp = lg % (tk * tc), split = lg / (tk * tc);
    const int k0 = (grp % tk) * TK, c0 = (grp / tk) * TC, r = tap / g.S, s = tap - r * g.S;
    const int M = g.N * g.Ho * g.Wo;
    const int mbeg = split * px_per_split, mend = min(M, mbeg + px_per_split);
    const __amdgpu_buffer_rsrc_t dy_rsrc = make_rsrc(dy, (unsigned)(g.N * g.Ho * g.Wo * g.K) * 4u);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(x, (unsigned)(g.N * g.H * g.W * g.C) * 4u);
    int mslab = mbeg;
    float vA[NI][8], vB[NI][8];            // two slabs in flight: a slab's loads are issued two multiply phases before it is staged
    auto load = [&](float (&v)[NI][8]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int it = wave + 4 * j;                              // wave-uniform
            const bool isA = it < ITEMS_A;
            const int idx = isA ? it : it - ITEMS_A, chalf = idx >> 2, pg = idx & 3;
            const int m0 = mslab + 8 * pg;
            if (isA) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int m = m0 + jj;
                    v[j][jj] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        dy_rsrc, m < mend ? ((unsigned)m * g.K + k0 + chalf * 64 + lane) * 4u : 0xffffffffu, 0, 0));
                }
            } else {
                int wo = m0 % g.Wo, t2 = m0 / g.Wo, ho = t2 % g.Ho, n = t2 / g.Ho;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int hi = ho * g.stride - g.pad + r, wi = wo * g.stride - g.pad + s;
                    const bool ok = m0 + jj < mend && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                    v[j][jj] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        x_rsrc, ok ? ((unsigned)((n * g.H + hi) * g.W + wi) * g.C + c0 + chalf * 64 + lane) * 4u : 0xffffffffu, 0, 0));
                    if (++wo >= g.Wo) {
                        wo = 0;
                        if (++ho >= g.Ho) { ho = 0; ++n; }
                    }
                }
            }
        }
        mslab += PS;
    };
    auto store = [&](const float (&v)[NI][8]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int it = wave + 4 * j;
            const bool isA = it < ITEMS_A;
            const int idx = isA ? it : it - ITEMS_A, chalf = idx >> 2, pg = idx & 3;
            const Split4 s0 = split3(make_float4(v[j][0], v[j][1], v[j][2], v[j][3]));
            const Split4 s1 = split3(make_float4(v[j][4], v[j][5], v[j][6], v[j][7]));
#pragma unroll
            for (int tm = 0; tm < 3; ++tm) {
                u32x4 o;
                o.x = s0.t[tm].x; o.y = s0.t[tm].y; o.z = s1.t[tm].x; o.w = s1.t[tm].y;
                unsigned short *dst = isA ? &Ah[tm][chalf * 64 + lane][8 * pg] : &Bh[tm][chalf * 64 + lane][8 * pg];
                *reinterpret_cast<u32x4 *>(dst) = o;
            }
        }
    };
    f32x16 acc[WTK][WTC];
#pragma unroll
    for (int i = 0; i < WTK; ++i)
#pragma unroll
        for (int j = 0; j < WTC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nslab = (mend - mbeg + PS - 1) / PS;
    if (nslab > 0) {
        load(vA);
        store(vA);
    }
    __syncthreads();
    if (nslab > 1) load(vA);                // slab 1
    if (nslab > 2) load(vB);                // slab 2
    auto multiply = [&]() {
#pragma unroll
        for (int ks = 0; ks < PS / 16; ++ks) {
            u32x4 a[WTK][3], b[WTC][3];
#pragma unroll
            for (int i = 0; i < WTK; ++i)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) a[i][tm] = *reinterpret_cast<const u32x4 *>(&Ah[tm][(wk * WTK + i) * 32 + row][ks * 16 + hh * 8]);
#pragma unroll
            for (int j = 0; j < WTC; ++j)
#pragma unroll
                for (int tm = 0; tm < 3; ++tm) b[j][tm] = *reinterpret_cast<const u32x4 *>(&Bh[tm][(wc * WTC + j) * 32 + row][ks * 16 + hh * 8]);
#pragma unroll
            for (int i = 0; i < WTK; ++i)
#pragma unroll
                for (int j = 0; j < WTC; ++j) {
                    f32x16 c = acc[i][j];                            // smallest terms first
                    c = mfma_bf(a[i][0], b[j][2], c);
                    c = mfma_bf(a[i][2], b[j][0], c);
                    c = mfma_bf(a[i][1], b[j][1], c);
                    c = mfma_bf(a[i][0], b[j][1], c);
                    c = mfma_bf(a[i][1], b[j][0], c);
                    acc[i][j] = mfma_bf(a[i][0], b[j][0], c);
                }
        }
    };
    for (int sl = 0; sl < nslab; sl += 2) {
        multiply();                                       // slab sl
        __syncthreads();
        if (sl + 1 < nslab) store(vA);                    // slab sl + 1
        __syncthreads();
        if (sl + 3 < nslab) load(vA);
        if (sl + 1 < nslab) {
            multiply();                                   // slab sl + 1
            __syncthreads();
            if (sl + 2 < nslab) store(vB);                // slab sl + 2
            __syncthreads();
            if (sl + 4 < nslab) load(vB);
        }
    }
    // C/D layout of the 32x32 MFMA: column (c) = lane & 31, row (k) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    float *po = part + (size_t)split * g.K * RS * g.C;
#pragma unroll
    for (int i = 0; i < WTK; ++i)
#pragma unroll
        for (int j = 0; j < WTC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + (wk * WTK + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh, cc = c0 + (wc * WTC + j) * 32 + row;
                po[((size_t)k * RS + tap) * g.C + cc] = acc[i][j][e];
            }
}

// ---------------------------------------------------------------------------------------------------
// wgrad, three-term bf16 operands straight from memory (impl 6, round 4).  The two kernels above that run on the bf16 matrix cores
// turn the tile in LDS (pixels must be the 8 consecutive reduction elements of a lane, memory has the channels contiguous) and pay
// for it with two barriers per 32 pixels and a workgroup-wide staging phase the matrix pipe waits behind.  But the MFMA only asks
// that A and B agree on WHICH pixel sits in reduction slot kk, and that the 32 rows of a tile are 32 different channels — not which:
//   * slot kk = 8 h + j (h = lane / 32) holds pixel p0 + 2 j + h: the two halves of the wave read the even / the odd pixel of a
//     pair, so every pixel quantity is wave-uniform (scalar unit) and the odd half's address is the even one + a lane constant;
//   * row i of tile ta holds filter k0 + KT i + ta (tile tb: channel c0 + CT i + tb): lane i fetches its KT (CT) consecutive
//     channels of a pixel as ONE 8 / 16-byte load (256 / 512 contiguous bytes per half wave), and the value it needs for tile ta
//     is element ta of that load — nothing is transposed, nothing goes through LDS, no barrier until the epilogue.
// A step = 16 pixels = 8 loads of dy + 8 of x per lane, 5.5 VALU instructions per value for the exact split into three bf16 terms,
// 6 x KT x CT v_mfma_f32_32x32x16_bf16 (smallest terms first).  The four waves of a workgroup walk four pixel ranges of one
// (filters, channels, tap) tile and add their accumulators through LDS in a fixed order.  Needs an even Wo (a pair never straddles
// an output row) unless the convolution is a plain 1x1 (x pixel = output pixel).  part[split][k][r][s][c] as for the others.
// ---------------------------------------------------------------------------------------------------
template <int KT, int CT, bool FLAT, int NW = 4, bool H2 = false>
__global__ __launch_bounds__(NW * 64, (KT * CT <= 4 || NW == 8 ? 2 : 1)) void conv_wgrad_direct3_kernel(const float *__restrict__ dy,
                                                                                                      const float *__restrict__ x,
                                                                                                      float *__restrict__ part, ConvGeom g,
                                                                                                      int px_per_wave, int nsplits, OpScale sc) {
    constexpr int NTM = H2 ? 2 : 3;                         // operand terms: three bf16, or (H2, impl 7) two fp16 of the scaled operands
    const unsigned bea = H2 ? scale_exp(sc.amax_a) : 127u, beb = H2 ? scale_exp(sc.amax_b) : 127u;
    const float sca = scale_from_exp(bea), scb = scale_from_exp(beb);
#ifdef SQD_WGRAD_TRACE
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    constexpr int NR = KT * CT * 16;                        // accumulator registers per lane
    static_assert((NW == 4 || NW == 8) && (256 / NW) % CT == 0 && (256 / NW / NW) % CT == 0, "4 or 8 waves");
    constexpr int PR = (NR < 256 / NW ? NR : 256 / NW);     // ... of which one pass of the final cross-wave sum takes PR (64 KB of LDS)
    constexpr bool TWO_RAW = KT * CT <= 4;                  // small tiles: two steps of loads in flight; large ones: the converted terms are the second buffer
    __shared__ float red[NW][PR][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int RS = g.R * g.S, tk = g.K / (32 * KT), tc = g.C / (32 * CT);
    const int logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // taps fastest inside an XCD's range
    if (logical >= RS * tk * tc * nsplits) return;
    const int tap = logical % RS, lg = logical / RS, grp = lg % (tk * tc), split = lg / (tk * tc);
    const int k0 = (grp % tk) * 32 * KT, c0 = (grp / tk) * 32 * CT, r = tap / g.S, s = tap - r * g.S;
    const int M = g.N * g.Ho * g.Wo;
    const int mbeg = (split * NW + wave) * px_per_wave, mend = min(M, mbeg + px_per_wave);
    const __amdgpu_buffer_rsrc_t dy_rsrc = make_rsrc(dy, (unsigned)(g.N * g.Ho * g.Wo * g.K) * 4u);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(x, (unsigned)(g.N * g.H * g.W * g.C) * 4u);
    // lane constants: its channels, + one pixel for the odd half
    const unsigned dyl = (unsigned)(k0 + KT * i32) * 4u + (h ? (unsigned)g.K * 4u : 0u);
    const unsigned xl = (unsigned)(c0 + CT * i32) * 4u + (h ? (unsigned)(FLAT ? g.C : g.stride * g.C) * 4u : 0u);
    // the even pixel of the next pair (wave-uniform)
    int m = mbeg;
    int wo = 0, ho = 0, n = 0;
    if (!FLAT) {
        wo = m % g.Wo;
        const int t2 = m / g.Wo;
        ho = t2 % g.Ho;
        n = t2 / g.Ho;
    }
    f32x16 acc[KT][CT];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    auto load_step = [&](float (*a)[KT], float (*b)[CT]) {
#ifdef SQD_WG_NOLOAD
        if (m > mbeg) {
            m += 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int q = 0; q < KT; ++q) asm volatile("" : "+v"(a[j][q]));
#pragma unroll
                for (int q = 0; q < CT; ++q) asm volatile("" : "+v"(b[j][q]));
            }
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ev = m < mend, od = m + 1 < mend;
            ldb(dy_rsrc, (h ? od : ev) ? dyl + (unsigned)m * (unsigned)g.K * 4u : 0xffffffffu, a[j]);
            if (FLAT) {
                ldb(x_rsrc, (h ? od : ev) ? xl + (unsigned)m * (unsigned)g.C * 4u : 0xffffffffu, b[j]);
            } else {
                const int hi = ho * g.stride - g.pad + r, wi = wo * g.stride - g.pad + s;
                const bool row = (unsigned)hi < (unsigned)g.H;
                const bool xe = ev && row && (unsigned)wi < (unsigned)g.W, xo = od && row && (unsigned)(wi + g.stride) < (unsigned)g.W;
                const unsigned so = (unsigned)(((n * g.H + hi) * g.W + wi) * g.C) * 4u;      // (may wrap below 0 for wi = -1: only the odd half adds to it)
                ldb(x_rsrc, (h ? xo : xe) ? xl + so : 0xffffffffu, b[j]);
                wo += 2;
                if (wo >= g.Wo) {
                    wo -= g.Wo;
                    if (++ho >= g.Ho) { ho = 0; ++n; }
                }
            }
            m += 2;
        }
    };
    auto convert = [&](float (*a)[KT], float (*b)[CT], u32x4 (*A)[NTM], u32x4 (*B)[NTM]) {
#ifdef SQD_WG_NOCVT
#pragma unroll
        for (int q = 0; q < KT; ++q)
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
                A[q][tm] = (u32x4){__float_as_uint(a[0 + tm][q]), __float_as_uint(a[1 + tm][q]), __float_as_uint(a[2 + tm][q]), __float_as_uint(a[3 + tm][q])};
#pragma unroll
        for (int q = 0; q < CT; ++q)
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
                B[q][tm] = (u32x4){__float_as_uint(b[0 + tm][q]), __float_as_uint(b[1 + tm][q]), __float_as_uint(b[2 + tm][q]), __float_as_uint(b[3 + tm][q])};
        return;
#endif
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            const SplitT<NTM> s0 = split_terms<NTM, H2>(make_float4(a[0][q], a[1][q], a[2][q], a[3][q]), sca),
                              s1 = split_terms<NTM, H2>(make_float4(a[4][q], a[5][q], a[6][q], a[7][q]), sca);
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) A[q][tm] = (u32x4){s0.t[tm].x, s0.t[tm].y, s1.t[tm].x, s1.t[tm].y};
        }
#pragma unroll
        for (int q = 0; q < CT; ++q) {
            const SplitT<NTM> s0 = split_terms<NTM, H2>(make_float4(b[0][q], b[1][q], b[2][q], b[3][q]), scb),
                              s1 = split_terms<NTM, H2>(make_float4(b[4][q], b[5][q], b[6][q], b[7][q]), scb);
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) B[q][tm] = (u32x4){s0.t[tm].x, s0.t[tm].y, s1.t[tm].x, s1.t[tm].y};
        }
    };
    auto multiply = [&](u32x4 (*A)[NTM], u32x4 (*B)[NTM]) {
#pragma unroll
        for (int q = 0; q < KT; ++q)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[q][c] = mma_terms<NTM, H2>(A[q], B[c], acc[q][c]);      // smallest terms first
    };
    auto mma_step = [&](float (*a)[KT], float (*b)[CT]) {
        u32x4 A[KT][NTM], B[CT][NTM];
        convert(a, b, A, B);
        multiply(A, B);
    };
    const int nst = (mend - mbeg + 15) / 16;                  // steps (loads past mend return zeros)
    if constexpr (TWO_RAW) {
        float a0[8][KT], b0[8][CT], a1[8][KT], b1[8][CT];
        if (nst > 0) load_step(a0, b0);
        for (int st = 0; st < nst; st += 2) {
            load_step(a1, b1);
            mma_step(a0, b0);
            load_step(a0, b0);
            mma_step(a1, b1);
        }
    } else {
        float a0[8][KT], b0[8][CT];
        if (nst > 0) load_step(a0, b0);
        for (int st = 0; st < nst; ++st) {
            u32x4 A[KT][NTM], B[CT][NTM];
            convert(a0, b0, A, B);
            load_step(a0, b0);                                // the next step's values arrive under this step's products
            multiply(A, B);
        }
    }
#ifdef SQD_WGRAD_TRACE
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    // ---- the four waves' accumulators added in a fixed order, PR registers per pass; register q = (ta * 16 + e) * CT + tb, wave w sums
    // q in [w PR/4, (w+1) PR/4) of a pass: CT consecutive channels per store.  C/D layout of the 32x32 MFMA: column = lane & 31,
    // row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    float *po = part + (size_t)split * g.K * RS * g.C;
    const float isa = inv_scale_from_exp(bea), isb = inv_scale_from_exp(beb);     // H2: the accumulators hold s_dy s_x times the partial sums
#pragma unroll
    for (int pass = 0; pass < NR / PR; ++pass) {
        if (pass > 0) __syncthreads();
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int b = 0; b < CT; ++b) {
                    const int q = (a * 16 + e) * CT + b;
                    if (q / PR == pass) red[wave][q % PR][lane] = acc[a][b][e];
                }
        __syncthreads();
        for (int q0 = wave * (PR / NW); q0 < (wave + 1) * (PR / NW); q0 += CT) {
            float v[CT];
#pragma unroll
            for (int b = 0; b < CT; ++b) {
                v[b] = ((red[0][q0 + b][lane] + red[1][q0 + b][lane]) + red[2][q0 + b][lane]) + red[3][q0 + b][lane];
                if (NW == 8) v[b] += ((red[4 % NW][q0 + b][lane] + red[5 % NW][q0 + b][lane]) + red[6 % NW][q0 + b][lane]) + red[7 % NW][q0 + b][lane];
                if (H2) v[b] = v[b] * isa * isb;
            }
            const int ae = (pass * PR + q0) / CT, a = ae >> 4, e = ae & 15;
            const int k = k0 + KT * ((e & 3) + 8 * (e >> 2) + 4 * h) + a, cc = c0 + CT * i32;
            float *dst = po + ((size_t)k * RS + tap) * g.C + cc;
            if (CT == 2) *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[1 % CT]);
            else if (CT == 4) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1 % CT], v[2 % CT], v[3 % CT]);
            else dst[0] = v[0];
        }
    }
#ifdef SQD_WGRAD_TRACE
    if (lane == 0) {
        unsigned long long *tr = reinterpret_cast<unsigned long long *>(part + (size_t)nsplits * g.K * RS * g.C) + ((size_t)blockIdx.x * NW + wave) * 4;
        tr[0] = t_entry; tr[1] = t_loop; tr[2] = __builtin_amdgcn_s_memtime(); tr[3] = (unsigned long long)logical;
    }
#endif
}

// sum of the split-K partial tiles (+ bias, + activation): part [Z][M*Ncols] -> out [M*Ncols]
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                          const float *__restrict__ addend, float *__restrict__ out, size_t n, int Z,
                                                          int Ncols, int act, unsigned *__restrict__ amax_out) {
    unsigned am = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4 *>(part)[i];
        for (int z = 1; z < Z; ++z) {
            const float4 b = reinterpret_cast<const float4 *>(part + (size_t)z * n)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        if (bias) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + (i * 4) % Ncols);
            a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
        }
        if (addend) {
            const float4 bv = reinterpret_cast<const float4 *>(addend)[i];
            a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
        }
        if (act == 1) {
            a.x = a.x > 0.f ? a.x : 0.f; a.y = a.y > 0.f ? a.y : 0.f; a.z = a.z > 0.f ? a.z : 0.f; a.w = a.w > 0.f ? a.w : 0.f;
        } else if (act == 2) {
            a.x = a.x > 0.f ? a.x : 0.01f * a.x; a.y = a.y > 0.f ? a.y : 0.01f * a.y;
            a.z = a.z > 0.f ? a.z : 0.01f * a.z; a.w = a.w > 0.f ? a.w : 0.01f * a.w;
        }
        reinterpret_cast<float4 *>(out)[i] = a;
        am = max(max(am, abs_bits(a.x)), max(abs_bits(a.y), max(abs_bits(a.z), abs_bits(a.w))));
    }
    amax_commit(am, amax_out);
}

// The same sum for a convolution whose output feeds a BatchNorm (forward, MODE 0) or whose data gradient is a BatchNorm's complete
// output gradient (MODE 1) — with that BatchNorm's per-channel partial sums taken on the way, as the unsplit kernels' epilogues do:
// a split-K plan no longer costs the BatchNorm a reduction pass of its own over the tensor (bn_reduce_kernel, one launch and one
// more read of the output).  A workgroup owns RED_ROWS rows x CH channels: thread (cg = t % TPR, rr = t / TPR) walks rows
// rr, rr + RP, ...; fixed-order LDS sum over the row groups; stats[row block][channel][2] (bn_reduce_kernel's layout).
// MODE 0: (sum y, sum y^2), y = sum_z part + bias.   MODE 1: (sum dz, sum dz * xhat), dx = sum_z part + addend, dz = dx * act'(mask).
constexpr int RED_ROWS = 64;
template <int MODE>
__global__ __launch_bounds__(256) void gemm_reduce_stats_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                                const float *__restrict__ addend, float *__restrict__ out, int M, int Z,
                                                                int Ncols, int act, float *__restrict__ stats, BnBwdSrc bnb, int CH,
                                                                unsigned *__restrict__ amax_out) {
    __shared__ float4 sh[2][256];
    unsigned am = 0u;
    const int t = threadIdx.x;
    const int chunk = min(CH, Ncols - (int)blockIdx.y * CH);            // channels of this workgroup (multiple of 4; CH = 256 .. 32)
    const int TPR = chunk / 4, RP = 256 / TPR;
    const int cg0 = t % TPR, rr = t / TPR;
    const int c = blockIdx.y * CH + cg0 * 4;
    const int r0 = blockIdx.x * RED_ROWS, r1 = min(M, r0 + RED_ROWS);
    const size_t n = (size_t)M * Ncols;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (rr < RP) {
        float4 bv = a, mu = a, rs = a;
        if (MODE == 0 && bias) bv = *reinterpret_cast<const float4 *>(bias + c);
        if (MODE == 1) {
            mu = *reinterpret_cast<const float4 *>(bnb.mean + c);
            rs = *reinterpret_cast<const float4 *>(bnb.rstd + c);
        }
        for (int row = r0 + rr; row < r1; row += RP) {
            const size_t o = (size_t)row * Ncols + c;
            float4 v = *reinterpret_cast<const float4 *>(part + o);
            for (int z = 1; z < Z; ++z) {
                const float4 w = *reinterpret_cast<const float4 *>(part + (size_t)z * n + o);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            if (MODE == 0) {
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (act == 1) {
                    v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                } else if (act == 2) {
                    v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                    v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
                }
            } else if (addend) {
                const float4 w = *reinterpret_cast<const float4 *>(addend + o);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            *reinterpret_cast<float4 *>(out + o) = v;
            am = max(max(am, abs_bits(v.x)), max(abs_bits(v.y), max(abs_bits(v.z), abs_bits(v.w))));
            if (MODE == 0) {
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                b.x = fmaf(v.x, v.x, b.x); b.y = fmaf(v.y, v.y, b.y); b.z = fmaf(v.z, v.z, b.z); b.w = fmaf(v.w, v.w, b.w);
            } else {
                const float4 xv = *reinterpret_cast<const float4 *>(bnb.x + o);
                const unsigned bits = bnb.mask ? bnb.mask[o / 4] : 0xfu;
                const float d0 = v.x * bn_bwd_act(bits, 0, bnb.act), d1 = v.y * bn_bwd_act(bits, 1, bnb.act);
                const float d2 = v.z * bn_bwd_act(bits, 2, bnb.act), d3 = v.w * bn_bwd_act(bits, 3, bnb.act);
                a.x += d0; a.y += d1; a.z += d2; a.w += d3;
                b.x = fmaf(d0, (xv.x - mu.x) * rs.x, b.x); b.y = fmaf(d1, (xv.y - mu.y) * rs.y, b.y);
                b.z = fmaf(d2, (xv.z - mu.z) * rs.z, b.z); b.w = fmaf(d3, (xv.w - mu.w) * rs.w, b.w);
            }
        }
    }
    amax_commit(am, amax_out);
    sh[0][t] = a;
    sh[1][t] = b;
    __syncthreads();
    if (rr == 0) {
        for (int k = 1; k < RP; ++k) {
            const float4 a2 = sh[0][k * TPR + cg0], b2 = sh[1][k * TPR + cg0];
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
        }
        float *o = stats + ((size_t)blockIdx.x * Ncols + c) * 2;
        reinterpret_cast<float4 *>(o)[0] = make_float4(a.x, b.x, a.y, b.y);
        reinterpret_cast<float4 *>(o)[1] = make_float4(a.z, b.z, a.w, b.w);
    }
}

// gradient through the activation a convolution / linear layer applied in its epilogue, from the OUTPUT y (sign(y) == sign(pre-
// activation) for both): ReLU g * (y > 0), LeakyReLU(0.01) g * (y > 0 ? 1 : 0.01)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ g, const float *__restrict__ y, float *__restrict__ out,
                                                      size_t n4, size_t n, int act, unsigned *__restrict__ amax_out) {
    const float neg = act == 2 ? 0.01f : 0.f;
    unsigned am = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4 *>(g)[i], b = reinterpret_cast<const float4 *>(y)[i];
        const float4 o = make_float4(b.x > 0.f ? a.x : neg * a.x, b.y > 0.f ? a.y : neg * a.y, b.z > 0.f ? a.z : neg * a.z, b.w > 0.f ? a.w : neg * a.w);
        reinterpret_cast<float4 *>(out)[i] = o;
        am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float o = y[i] > 0.f ? g[i] : neg * g[i];
        out[i] = o;
        am = max(am, abs_bits(o));
    }
    amax_commit(am, amax_out);
}

// column sums of a [M, K] matrix (bias gradient), deterministic: block b adds rows [b*rpb, (b+1)*rpb) of a band of up
// to 256 columns (blockIdx.y): thread (col4 = t % cpb, grp = t / cpb) adds rows grp, grp + 256/cpb, ..., then a
// fixed-order LDS tree over the groups.  Applied twice: dy -> part [nblk][K] -> out [K].
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ src, float *__restrict__ dst, int M, int K, int rpb,
                                                     int cpb) {
    __shared__ float4 red[256];
    const int t = threadIdx.x, col = t % cpb, grp = t / cpb, ngrp = 256 / cpb;
    const int c4 = blockIdx.y * 64 + col;                      // float4 column
    const int r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 * 4 < K)
        for (int m = r0 + grp; m < r1; m += ngrp) {
            const float4 b = *reinterpret_cast<const float4 *>(src + (size_t)m * K + c4 * 4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    red[t] = a;
    __syncthreads();
    for (int w = ngrp >> 1; w >= 1; w >>= 1) {
        if (grp < w) {
            const float4 b = red[t + w * cpb];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            red[t] = a;
        }
        __syncthreads();
    }
    if (grp == 0 && c4 * 4 < K) *reinterpret_cast<float4 *>(dst + (size_t)blockIdx.x * K + c4 * 4) = a;
}

int check_geom(const char *who, const ConvGeom &g) {
    SQD_CHECK_ARG(g.N > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.K > 0 && g.R > 0 && g.S > 0 && g.stride > 0 && g.pad >= 0,
                  "%s: bad geometry", who);
    SQD_CHECK_ARG((long long)g.N * g.H * g.W * g.C < (1ll << 30) && (long long)g.N * g.Ho * g.Wo * g.K < (1ll << 30) &&
                      (long long)g.K * g.R * g.S * g.C < (1ll << 30),
                  "%s: tensors of 4 GiB or more are not supported (32-bit byte offsets of the buffer loads)", who);
    // (Ho, Wo) may be SMALLER than the full output extent: the top-left Ho x Wo outputs are computed (the 7x7/2 stems run as a
    // 4x4/1 convolution on a space-to-depth input, whose symmetric padding yields one surplus row and column)
    SQD_CHECK_ARG(g.Ho >= 1 && g.Wo >= 1 && g.Ho <= (g.H + 2 * g.pad - g.R) / g.stride + 1 &&
                      g.Wo <= (g.W + 2 * g.pad - g.S) / g.stride + 1,
                  "%s: Ho/Wo inconsistent with H/W/R/S/stride/pad", who);
    return SQD_OK;
}
}  // namespace

extern "C" int sqd_conv_supported(int C, int K) { return (C % 4 == 0 && K % 4 == 0) ? 1 : 0; }

// 0: fp32 MFMA (default); 1: split-precision bf16 MFMA (three terms, fp32-level accuracy); 2: plain bf16 operands with fp32
// accumulation — for forward / data gradient (sqd_conv_set_precision); the weight gradient stays on the fp32 MFMA kernels
static int &conv_precision() {
    static int prec = 0;
    return prec;
}

struct GemmPlan {
    int bm, bn, z, bk;
    int waves = 4;                  // wavefronts per workgroup: 4, or 8 on the >= 128x64 tiles (one 32x32 tile per wave)
    int single = 0;                 // 1: single-buffered LDS variant (bk + 512 in sqd_conv_set_plan)
    int split3 = 0;                 // 1: three-term bf16 operands on the bf16 matrix cores, single LDS buffer, 32-channel slices (bk + 1024)
    int halo = 0;                   // 1: conv3x3_halo_kernel (bk + 2048; 3x3 / stride 1 / pad 1 only): bm = pixels of a patch, bn = channels of a tile
    int h2 = 0;                     // 1: with split3 (and halo): two fp16 terms of the scaled operands instead of three bf16 terms (bk + 4096); needs the operands' max |.|
    int64_t ws_floats;
};
// measured plans registered through sqd_conv_set_plan: (mode, geometry) -> (bm, bn, z)
typedef std::tuple<int, int, int, int, int, int, int, int, int, int> PlanKey;
static std::map<PlanKey, std::tuple<int, int, int, int>> &plan_table() {
    static std::map<PlanKey, std::tuple<int, int, int, int>> t;
    return t;
}
static std::mutex &plan_mutex() {
    static std::mutex m;
    return m;
}
static PlanKey plan_key(int mode, const ConvGeom &g) {
    return PlanKey(mode, g.N, g.H, g.W, g.C, g.K, g.R, g.S, g.stride, g.pad);
}
// tile / split-K choice: enough workgroups to cover 256 CUs a few times over, partial workspace <= 64 MB
static GemmPlan plan_gemm(int mode, const ConvGeom &g) {
    int ncls = 1;                                 // stride classes that have at least one tap (the others only zero-fill)
    if (mode == 1) {
        int ah = 0, aw = 0;
        for (int p = 0; p < g.stride; ++p) {
            ah += (p + g.pad) % g.stride < g.R;
            aw += (p + g.pad) % g.stride < g.S;
        }
        ncls = ah * aw > 0 ? ah * aw : 1;
    }
    const int Mcls = mode == 0 ? g.N * g.Ho * g.Wo : g.N * ((g.H + g.stride - 1) / g.stride) * ((g.W + g.stride - 1) / g.stride);
    const int Mrows = mode == 0 ? g.N * g.Ho * g.Wo : g.N * g.H * g.W;     // output rows (workspace size)
    const int Ncols = mode == 0 ? g.K : g.C;
    const int taps = mode == 0 ? g.R * g.S : ((g.R + g.stride - 1) / g.stride) * ((g.S + g.stride - 1) / g.stride);
    const int T = taps * (((mode == 0 ? g.C : g.K) + BK - 1) / BK);
    GemmPlan p;
    // Tile and split-K by a small cost model (cycles on the busiest CU), calibrated on the config-B layers
    // (profiles/r01c_conv_layers.md): a workgroup-step costs its MFMA cycles (8 x 64 per 32x32 wave tile) and, when too
    // few workgroups share a CU to hide it, ~400 cycles of load/barrier latency; split-K pays for writing and
    // re-reading z partial images.
    static const int cand[5][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}, {128, 32}};
    static const int zs[9] = {1, 2, 3, 4, 6, 8, 10, 12, 16};
    double best = 1e30;
    p.bm = 128; p.bn = 32; p.z = 1; p.bk = 16;
    for (int ci = 0; ci < 5; ++ci) {
        const int bm = cand[ci][0], bn = cand[ci][1];
        if (bn > 32 && bn >= 2 * Ncols) continue;                       // a tile wider than twice the channel count
        if (bn == 32 && Ncols > 32) continue;
        const long long tiles = (long long)ncls * ((Mcls + bm - 1) / bm) * ((Ncols + bn - 1) / bn);
        const double mfma = (bm / 32) * (bn / 32) / 4.0 * 512.0;
        const int occ_max = bm * bn >= 128 * 128 ? 2 : 4;
        for (int zi = 0; zi < 9; ++zi) {
            const int z = zs[zi];
            if (z > 1 && (z > T / 4 || (int64_t)z * Mrows * Ncols * 4 > (64ll << 20) || ((int64_t)Mrows * Ncols) % 4 != 0)) continue;
            const long long wgs = tiles * z;
            const long long per_cu = (wgs + 255) / 256;
            const int occ = per_cu < occ_max ? (int)per_cu : occ_max;
            const double steps = (double)(T + z - 1) / z;
            const double rounds = (double)((per_cu + occ - 1) / occ);
            const double step_cyc = occ * mfma > mfma + 400.0 ? occ * mfma : mfma + 400.0;
            double cyc = rounds * (steps * step_cyc + 1500.0);         // + prologue / epilogue per round
            if (z > 1) cyc += 2.4e9 * ((z + 1.0) * Mrows * Ncols * 4.0 / 3.0e12 + 3.0e-6);
            if (cyc < best) { best = cyc; p.bm = bm; p.bn = bn; p.z = z; }
        }
    }
    {   // a plan measured by the caller (sqd_conv_set_plan) overrides the model
        std::lock_guard<std::mutex> lk(plan_mutex());
        auto it = plan_table().find(plan_key(mode, g));
        if (it != plan_table().end()) {
            p.bm = std::get<0>(it->second);
            p.bn = std::get<1>(it->second);
            p.z = std::get<2>(it->second);
            p.bk = std::get<3>(it->second) & 255;
            p.waves = (std::get<3>(it->second) & 256) ? 8 : 4;
            p.single = (std::get<3>(it->second) & 512) ? 1 : 0;
            p.split3 = (std::get<3>(it->second) & 1024) ? 1 : 0;
            p.halo = (std::get<3>(it->second) & 2048) ? 1 : 0;
            p.h2 = (std::get<3>(it->second) & 4096) ? 1 : 0;
        }
    }
    if (conv_precision() != 0) {
        // the operand-precision modes run ONE kernel family (DISPATCH_GEMM_BF): the plan describes the tile that is launched,
        // so that the BatchNorm partial rows sqd_conv_fwd_stats_rows reports are the rows the kernel writes
        p.halo = p.split3 = p.single = p.h2 = 0;
        p.waves = 4;
        if (p.bm == 128 && p.bn >= 64) p.bn = 64;
        else if (p.bm == 128) p.bn = 32;
        else if (p.bn == 128) p.bm = 64;
        else { p.bm = 64; p.bn = 64; }
        if (!(p.bm == 64 && p.bn == 64 && p.bk == 32)) p.bk = 16;
    }
    int z = p.z;
    if (p.halo) z = z <= ((mode == 0 ? g.C : g.K) + 31) / 32 ? z : 1;          // split over 32-channel chunks
    else if (p.bk >= 32) z = z <= T / (p.bk / 8) ? z : 1;        // (slices are 2x / 4x as wide)
    p.z = z;
    p.ws_floats = z > 1 ? (int64_t)z * Mrows * Ncols : 0;
    return p;
}

#define LAUNCH_GEMM_P(MODE, BM, BN, WGM, WGN, BKT, PREC)                                                                 \
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, BM, BN, WGM, WGN, BKT, PREC>),                                           \
                       dim3((ncls * ((Mcls + BM - 1) / BM) * ((Ncols + BN - 1) / BN) + 7) / 8 * 8, 1, p.z), dim3(256), 0, st, \
                       a_src, w, bias, dst, g, act, p.z, order, stats, bnb, sc)
#define LAUNCH_GEMM(MODE, BM, BN, WGM, WGN, BKT) LAUNCH_GEMM_P(MODE, BM, BN, WGM, WGN, BKT, 0)
#define LAUNCH_GEMM8(MODE, BM, BN, WGM, WGN, BKT)                                                                       \
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, BM, BN, WGM, WGN, BKT, 0>),                                              \
                       dim3((ncls * ((Mcls + BM - 1) / BM) * ((Ncols + BN - 1) / BN) + 7) / 8 * 8, 1, p.z), dim3(512), 0, st, \
                       a_src, w, bias, dst, g, act, p.z, order, stats, bnb, sc)
#define LAUNCH_GEMM8_P(MODE, BM, BN, WGM, WGN, BKT, PREC)                                                                \
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, BM, BN, WGM, WGN, BKT, PREC>),                                           \
                       dim3((ncls * ((Mcls + BM - 1) / BM) * ((Ncols + BN - 1) / BN) + 7) / 8 * 8, 1, p.z), dim3(512), 0, st, \
                       a_src, w, bias, dst, g, act, p.z, order, stats, bnb, sc)
#define LAUNCH_HALO4(WTM, WM, WN, WK)                                                                                      \
    if (p.h2)                                                                                                              \
        hipLaunchKernelGGL((conv3x3_halo_kernel<0, WTM, WM, WN, WK, 4, true>),                                                   \
                       dim3(g.N * ((g.H + 2 * WTM - 1) / (2 * WTM)) * ((g.W + 15) / 16) * ((Ncols + WN * 32 - 1) / (WN * 32)), 1, p.z), \
                       dim3(WM * WN * WK * 64), 0, st, a_src, w, bias, dst, g, act, p.z, stats, bnb, sc);                          \
    else                                                                                                                   \
    hipLaunchKernelGGL((conv3x3_halo_kernel<0, WTM, WM, WN, WK, 4>),                                                         \
                       dim3(g.N * ((g.H + 2 * WTM - 1) / (2 * WTM)) * ((g.W + 15) / 16) * ((Ncols + WN * 32 - 1) / (WN * 32)), 1, p.z), \
                       dim3(WM * WN * WK * 64), 0, st, a_src, w, bias, dst, g, act, p.z, stats, bnb, sc)
#define LAUNCH_HALO(MODE, WTM, WM, WN, WK)                                                                                 \
    if (p.h2)                                                                                                              \
        hipLaunchKernelGGL((conv3x3_halo_kernel<MODE, WTM, WM, WN, WK, 3, true>),                                                \
                       dim3(g.N * ((g.H + 2 * WTM - 1) / (2 * WTM)) * ((g.W + 15) / 16) * ((Ncols + WN * 32 - 1) / (WN * 32)), 1, p.z), \
                       dim3(WM * WN * WK * 64), 0, st, a_src, w, bias, dst, g, act, p.z, stats, bnb, sc);                          \
    else                                                                                                                   \
    hipLaunchKernelGGL((conv3x3_halo_kernel<MODE, WTM, WM, WN, WK>),                                                         \
                       dim3(g.N * ((g.H + 2 * WTM - 1) / (2 * WTM)) * ((g.W + 15) / 16) * ((Ncols + WN * 32 - 1) / (WN * 32)), 1, p.z), \
                       dim3(WM * WN * WK * 64), 0, st, a_src, w, bias, dst, g, act, p.z, stats, bnb, sc)
#define DISPATCH_GEMM(MODE)                                                      \
    if (p.halo && g.R == 4) {                /* 4x4 taps: the space-to-depth stems (forward only); few input channels: waves split pixels */ \
        if (MODE == 0) {                                                         \
            if (p.bm == 128 && p.bn == 64) LAUNCH_HALO4(4, 2, 2, 1);             \
            else if (p.bm == 128) LAUNCH_HALO4(4, 4, 1, 1);                      \
            else if (p.bn == 64) LAUNCH_HALO4(2, 2, 2, 1);                       \
            else LAUNCH_HALO4(2, 2, 1, 2);                                       \
        }                                                                        \
    } else if (p.halo && p.waves == 8) {                                         \
        if (p.bm == 128 && p.bn == 128) LAUNCH_HALO(MODE, 4, 2, 4, 1);           \
        else if (p.bm == 128 && p.bn == 64) LAUNCH_HALO(MODE, 4, 2, 2, 2);       \
        else if (p.bn == 128) LAUNCH_HALO(MODE, 2, 2, 4, 1);                     \
        else LAUNCH_HALO(MODE, 2, 2, 2, 2);                                      \
    } else if (p.halo) {                                                         \
        if (p.bm == 128 && p.bn == 128) LAUNCH_HALO(MODE, 4, 1, 4, 1);           \
        else if (p.bm == 128 && p.bn == 64) LAUNCH_HALO(MODE, 4, 1, 2, 2);       \
        else if (p.bm == 128) LAUNCH_HALO(MODE, 4, 2, 1, 2);                     \
        else if (p.bn == 128) LAUNCH_HALO(MODE, 2, 1, 4, 1);                     \
        else if (p.bn == 64) LAUNCH_HALO(MODE, 2, 1, 2, 2);                      \
        else LAUNCH_HALO(MODE, 2, 2, 1, 2);                                      \
    } else if (p.split3 && p.h2 && p.waves == 8) {                                     \
        if (p.bm == 128 && p.bn == 128) LAUNCH_GEMM8_P(MODE, 128, 128, 4, 2, 32, 5);     \
        else if (p.bm == 128) LAUNCH_GEMM8_P(MODE, 128, 64, 4, 2, 32, 5);               \
        else LAUNCH_GEMM8_P(MODE, 64, 128, 2, 4, 32, 5);                                \
    } else if (p.split3 && p.h2) {                                                      \
        if (p.bm == 128 && p.bn == 128) LAUNCH_GEMM_P(MODE, 128, 128, 2, 2, 32, 5);      \
        else if (p.bm == 128 && p.bn == 32) LAUNCH_GEMM_P(MODE, 128, 32, 4, 1, 32, 5);   \
        else if (p.bm == 128) LAUNCH_GEMM_P(MODE, 128, 64, 2, 2, 32, 5);         \
        else if (p.bn == 128) LAUNCH_GEMM_P(MODE, 64, 128, 2, 2, 32, 5);         \
        else LAUNCH_GEMM_P(MODE, 64, 64, 2, 2, 32, 5);                           \
    } else if (p.split3 && p.waves == 8) {                                             \
        if (p.bm == 128 && p.bn == 128) LAUNCH_GEMM8_P(MODE, 128, 128, 4, 2, 32, 4);     \
        else if (p.bm == 128) LAUNCH_GEMM8_P(MODE, 128, 64, 4, 2, 32, 4);               \
        else LAUNCH_GEMM8_P(MODE, 64, 128, 2, 4, 32, 4);                                \
    } else if (p.split3) {                                                              \
        if (p.bm == 128 && p.bn == 128) LAUNCH_GEMM_P(MODE, 128, 128, 2, 2, 32, 4);      \
        else if (p.bm == 128 && p.bn == 32) LAUNCH_GEMM_P(MODE, 128, 32, 4, 1, 32, 4);   \
        else if (p.bm == 128) LAUNCH_GEMM_P(MODE, 128, 64, 2, 2, 32, 4);         \
        else if (p.bn == 128) LAUNCH_GEMM_P(MODE, 64, 128, 2, 2, 32, 4);         \
        else LAUNCH_GEMM_P(MODE, 64, 64, 2, 2, 32, 4);                           \
    } else if (p.single && p.bm == 64 && p.bn == 64) {                                  \
        if (p.bk == 64) LAUNCH_GEMM_P(MODE, 64, 64, 2, 2, 64, 2);                \
        else if (p.bk == 32) LAUNCH_GEMM_P(MODE, 64, 64, 2, 2, 32, 2);           \
        else LAUNCH_GEMM_P(MODE, 64, 64, 2, 2, 16, 2);                           \
    } else if (p.single && p.bm == 128 && p.bn == 64 && p.bk == 32) {            \
        LAUNCH_GEMM_P(MODE, 128, 64, 2, 2, 32, 2);                               \
    } else if (p.single && p.bm == 64 && p.bn == 128 && p.bk == 32) {            \
        LAUNCH_GEMM_P(MODE, 64, 128, 2, 2, 32, 2);                               \
    } else if (p.single && p.bm == 128 && p.bn == 128) {                         \
        LAUNCH_GEMM_P(MODE, 128, 128, 2, 2, 16, 2);                              \
    } else if (p.single && p.bm == 128 && p.bn == 32) {                          \
        if (p.bk == 32) LAUNCH_GEMM_P(MODE, 128, 32, 4, 1, 32, 2);               \
        else LAUNCH_GEMM_P(MODE, 128, 32, 4, 1, 16, 2);                          \
    } else if (p.waves == 8 && p.bm == 128 && p.bn == 128) LAUNCH_GEMM8(MODE, 128, 128, 4, 2, 16); \
    else if (p.waves == 8 && p.bm == 128 && p.bn == 64) {                        \
        if (p.bk == 32) LAUNCH_GEMM8(MODE, 128, 64, 4, 2, 32);                   \
        else LAUNCH_GEMM8(MODE, 128, 64, 4, 2, 16);                              \
    } else if (p.waves == 8 && p.bm == 64 && p.bn == 128 && p.bk == 32) {        \
        LAUNCH_GEMM8(MODE, 64, 128, 2, 4, 32);                                   \
    } else if (p.bm == 128 && p.bn == 128) LAUNCH_GEMM(MODE, 128, 128, 2, 2, 16);       \
    else if (p.bm == 128 && p.bn == 64) {                                        \
        if (p.bk == 32) LAUNCH_GEMM(MODE, 128, 64, 2, 2, 32);                    \
        else LAUNCH_GEMM(MODE, 128, 64, 2, 2, 16);                               \
    } else if (p.bm == 128) {                                                    \
        if (p.bk == 32) LAUNCH_GEMM(MODE, 128, 32, 4, 1, 32);                    \
        else LAUNCH_GEMM(MODE, 128, 32, 4, 1, 16);                               \
    } else if (p.bn == 128) {                                                    \
        if (p.bk == 32) LAUNCH_GEMM(MODE, 64, 128, 2, 2, 32);                    \
        else LAUNCH_GEMM(MODE, 64, 128, 2, 2, 16);                               \
    } else {                                                                     \
        if (p.bk == 32) LAUNCH_GEMM(MODE, 64, 64, 2, 2, 32);                     \
        else LAUNCH_GEMM(MODE, 64, 64, 2, 2, 16);                                \
    }
// split-precision variants: the three bf16 planes need 1.5x the LDS of the fp32 tile, so 128x128 runs as 128x64 and only the
// 64x64 tile keeps the 32-channel slice
#define DISPATCH_GEMM_BF(MODE, PR)                                               \
    if (p.bm == 128 && p.bn >= 64) LAUNCH_GEMM_P(MODE, 128, 64, 2, 2, 16, PR);   \
    else if (p.bm == 128) LAUNCH_GEMM_P(MODE, 128, 32, 4, 1, 16, PR);            \
    else if (p.bn == 128) LAUNCH_GEMM_P(MODE, 64, 128, 2, 2, 16, PR);            \
    else if (p.bk == 32) LAUNCH_GEMM_P(MODE, 64, 64, 2, 2, 32, PR);              \
    else LAUNCH_GEMM_P(MODE, 64, 64, 2, 2, 16, PR);

static int launch_gemm(int mode, const float *a_src, const float *w, const float *bias, float *out, float *ws, const ConvGeom &g,
                       int act, void *stream, float *stats = nullptr, BnBwdSrc bnb = BnBwdSrc{nullptr, nullptr, nullptr, nullptr, 0},
                       OpScale sc = OpScale{nullptr, nullptr, nullptr}) {
    const int ncls = mode == 0 ? 1 : g.stride * g.stride;
    const int Mcls = mode == 0 ? g.N * g.Ho * g.Wo : g.N * ((g.H + g.stride - 1) / g.stride) * ((g.W + g.stride - 1) / g.stride);
    const int Mrows = mode == 0 ? g.N * g.Ho * g.Wo : g.N * g.H * g.W;
    const int Ncols = mode == 0 ? g.K : g.C;
    const GemmPlan p = plan_gemm(mode, g);
    if (p.z > 1 && !ws) {
        sqd::set_error("sqd_conv: this shape needs a split-K workspace of %lld floats (sqd_conv_plan)", (long long)p.ws_floats);
        return SQD_EINVAL;
    }
    if (p.h2 && !(sc.amax_a && sc.amax_b)) {
        sqd::set_error("sqd_conv: the plan registered for this geometry runs on two-term fp16 operands and needs the max |.| of both operands "
                       "(sqd_conv_fwd_scaled / sqd_conv_dgrad_scaled; sqd_amax)");
        return SQD_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const int order = 2;                                   // N-tiles fastest, no XCD chunking: measured best on MI355X (profiles/r01c_conv_layers.md)
    float *dst = p.z > 1 ? ws : out;
    (void)hipGetLastError();
    if (conv_precision() == 1) {
        if (mode == 0) { DISPATCH_GEMM_BF(0, 1) } else { DISPATCH_GEMM_BF(1, 1) }
    } else if (conv_precision() == 2) {
        if (mode == 0) { DISPATCH_GEMM_BF(0, 3) } else { DISPATCH_GEMM_BF(1, 3) }
    } else if (mode == 0) { DISPATCH_GEMM(0) } else { DISPATCH_GEMM(1) }
    if (p.z > 1 && stats && (mode == 0 || bnb.x != nullptr)) {
        // split reduction + the following BatchNorm's partial sums in one pass: stats [ceil(Mrows / RED_ROWS)][Ncols][2]
        // the channels of a row block are cut into chunks of 256 .. 32 until the launch has ~1000 workgroups (split plans are the
        // few-pixel layers: 1440 or 5760 rows)
        const int mb = (Mrows + RED_ROWS - 1) / RED_ROWS;
        int CH = 256;
        while (CH > 32 && mb * ((Ncols + CH - 1) / CH) < 1024) CH >>= 1;
        const dim3 grid(mb, (Ncols + CH - 1) / CH);
        if (mode == 0)
            hipLaunchKernelGGL((gemm_reduce_stats_kernel<0>), grid, dim3(256), 0, st, ws, bias, (const float *)nullptr, out, Mrows, p.z, Ncols, act, stats, bnb, CH, sc.amax_out);
        else
            hipLaunchKernelGGL((gemm_reduce_stats_kernel<1>), grid, dim3(256), 0, st, ws, (const float *)nullptr, bias, out, Mrows, p.z, Ncols, 0, stats, bnb, CH, sc.amax_out);
    } else if (p.z > 1) {
        const size_t n = (size_t)Mrows * Ncols;
        size_t nb = (n / 4 + 255) / 256;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, st, ws, mode == 0 ? bias : nullptr, mode == 1 ? bias : nullptr,
                           out, n, p.z, Ncols, mode == 0 ? act : 0, sc.amax_out);
    }
    return SQD_OK;
}

// Register a measured tile / split-K plan for one geometry (mode 0 = fwd, 1 = dgrad): bm x bn in {128x128, 128x64, 64x128,
// 64x64, 128x32}, z >= 1 split-K factor, bk = 16 | 32 channels per reduction slice (+ 256: 8-wave workgroups on the 128x128 /
// 128x64 / 64x128 tiles — twice the resident waves for the same tiles); bm = 0 removes the entry.  Returns SQD_EINVAL when the plan cannot run on this
// geometry (tile wider than twice the channel count, z larger than a quarter of the reduction slices, workspace > 64 MB).
// The next sqd_conv_plan / sqd_conv_fwd / sqd_conv_dgrad calls of that geometry use it.
extern "C" int sqd_conv_set_plan(int mode, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                                 int bm, int bn, int z, int bk) {
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    std::lock_guard<std::mutex> lk(plan_mutex());
    if (bm == 0) {
        plan_table().erase(plan_key(mode, g));
        return SQD_OK;
    }
    const int Ncols = mode == 0 ? K : C;
    const int taps = mode == 0 ? R * S : ((R + stride - 1) / stride) * ((S + stride - 1) / stride);
    const int waves = (bk & 256) ? 8 : 4;                        // bk + 256: 8-wave workgroups (one 32x32 tile per wave)
    const int single = (bk & 512) ? 1 : 0;                       // bk + 512: single-buffered LDS (twice the resident workgroups)
    const int split3 = (bk & 1024) ? 1 : 0;                      // bk + 1024: three-term bf16 operands (fp32-level accuracy on the bf16 matrix cores)
    const int halo = (bk & 2048) ? 1 : 0;                        // bk + 2048: the input-patch kernel for 3x3 / stride 1 / pad 1 (three-term bf16 operands)
    const int h2 = (bk & 4096) ? 4096 : 0;                       // bk + 4096 (with + 1024): two fp16 terms of the scaled operands instead of three bf16 terms
    bk &= 255;
    SQD_CHECK_ARG(!h2 || split3, "sqd_conv_set_plan: + 4096 (two-term fp16 operands) modifies the + 1024 plans");
    SQD_CHECK_ARG(conv_precision() == 0 || !(halo || split3 || single || waves == 8),
                  "sqd_conv_set_plan: the single-buffered / 8-wave / three-term / input-patch variants exist for the fp32 arithmetic only "
                  "(sqd_conv_set_precision is %d)", conv_precision());
    if (halo) {
        const int Cr = mode == 0 ? C : K;
        SQD_CHECK_ARG(split3 && !single && bk == 32 && (waves == 4 || (R == 3 && bn >= 64)),
                      "sqd_conv_set_plan: the input-patch plans are bk = 32 + 1024 + 2048 (+ 256: 8-wave workgroups, 3x3 with >= 64-channel tiles)");
        SQD_CHECK_ARG(((R == 3 && S == 3 && pad == 1) || (R == 4 && S == 4 && pad == 2 && mode == 0)) && stride == 1 && Ho == H && Wo == W,
                      "sqd_conv_set_plan: the input-patch kernel is 3x3 / stride 1 / pad 1, or the forward 4x4 / stride 1 / pad 2 of the space-to-depth stems");
        SQD_CHECK_ARG((bm == 128 || bm == 64) && (bn == 128 || bn == 64 || bn == 32) && !(bn > 32 && bn >= 2 * Ncols) && !(bn == 32 && Ncols > 32) &&
                          !(R == 4 && bn == 128),
                      "sqd_conv_set_plan: input-patch tiles are 128|64 pixels x 128|64|32 channels (4x4: 64|32 channels)");
        const int64_t oe = (int64_t)N * H * W * Ncols;
        SQD_CHECK_ARG(z >= 1 && z <= 64 && (z == 1 || (z <= (Cr + 31) / 32 && z * oe * 4 <= (64ll << 20) && oe % 4 == 0)), "sqd_conv_set_plan: split %d not possible here", z);
        plan_table()[plan_key(mode, g)] = std::make_tuple(bm, bn, z, 32 | 1024 | 2048 | h2 | (waves == 8 ? 256 : 0));
        return SQD_OK;
    }
    if (split3 && waves == 8) {
        SQD_CHECK_ARG(!single && bk == 32 && ((bm == 128 && (bn == 128 || (bn == 64 && mode == 0))) || (bm == 64 && bn == 128)),
                      "sqd_conv_set_plan: the 8-wave three-term variants are 128x128, 64x128 and (forward) 128x64 tiles with 32-channel slices");
    } else if (split3) {
        SQD_CHECK_ARG(waves == 4 && !single && bk == 32 && (((bm == 128 || bm == 64) && (bn == 128 || bn == 64)) || (bm == 128 && bn == 32)),
                      "sqd_conv_set_plan: the three-term bf16 variants are 4-wave 128/64 x 128/64 and 128x32 tiles with 32-channel slices");
    }
    SQD_CHECK_ARG(!single || (waves == 4 && ((bm == 64 && bn == 64) || (bm + bn == 192 && bk == 32) || (bm == 128 && bn == 128 && bk == 16) ||
                                             (bm == 128 && bn == 32))),
                  "sqd_conv_set_plan: single-buffered variants exist for 64x64, 128x32, 128x64 / 64x128 (bk 32) and 128x128 (bk 16)");
    SQD_CHECK_ARG(waves == 4 || split3 || (bm == 128 && bn == 128 && bk == 16) || (bm == 128 && bn == 64) || (bm == 64 && bn == 128 && bk == 32),
                  "sqd_conv_set_plan: 8-wave workgroups exist for 128x128 (bk 16), 128x64 and 64x128 (bk 32) tiles");
    SQD_CHECK_ARG(bk == 16 || split3 || (bk == 32 && (mode == 0 ? C : K) % 32 == 0 && bm + bn <= 192) ||
                      (bk == 64 && single && bm == 64 && bn == 64 && (mode == 0 ? C : K) % 64 == 0),
                  "sqd_conv_set_plan: slice width %d not possible here (32 needs 32 | reduced channels and bm + bn <= 192; 64 exists "
                  "for the single-buffered 64x64 tile with 64 | reduced channels)", bk);
    const int T = taps * (((mode == 0 ? C : K) + bk - 1) / bk);
    const int64_t out_elems = mode == 0 ? (int64_t)N * Ho * Wo * K : (int64_t)N * H * W * C;
    const bool tile_ok = (bm == 128 && (bn == 128 || bn == 64 || bn == 32)) || (bm == 64 && (bn == 128 || bn == 64));
    SQD_CHECK_ARG(tile_ok && z >= 1 && z <= 64, "sqd_conv_set_plan: unsupported plan %dx%d z=%d", bm, bn, z);
    SQD_CHECK_ARG(!(bn > 32 && bn >= 2 * Ncols) && !(bn == 32 && Ncols > 32), "sqd_conv_set_plan: tile width %d does not fit %d channels", bn, Ncols);
    SQD_CHECK_ARG(z == 1 || (z <= T / 2 && z * out_elems * 4 <= (64ll << 20) && out_elems % 4 == 0),
                  "sqd_conv_set_plan: split-K %d not possible here", z);
    plan_table()[plan_key(mode, g)] = std::make_tuple(bm, bn, z, bk | (waves == 8 ? 256 : 0) | (single ? 512 : 0) | (split3 ? 1024 : 0) | h2);
    return SQD_OK;
}

// arithmetic of sqd_conv_fwd / sqd_conv_dgrad: 0 = fp32 MFMA (default), 1 = split-precision bf16 MFMA (3 bf16 terms per fp32
// operand, 6 products, fp32 accumulation: fp32-level accuracy, experimental)
extern "C" int sqd_conv_set_precision(int prec) {
    SQD_CHECK_ARG(prec >= 0 && prec <= 2, "sqd_conv_set_precision: %d (0 fp32, 1 three-term bf16 split, 2 bf16 operands)", prec);
    if (prec != conv_precision()) {
        // plans are measured per arithmetic: a table filled under another mode would name kernels this mode does not have
        std::lock_guard<std::mutex> lk(plan_mutex());
        plan_table().clear();
    }
    conv_precision() = prec;
    return SQD_OK;
}
extern "C" int sqd_conv_precision(void) { return conv_precision(); }

// workspace (floats) sqd_conv_fwd (mode 0) / sqd_conv_dgrad (mode 1) need for this geometry (0 = none)
extern "C" int sqd_conv_plan(int mode, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                             int64_t *ws_floats) {
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (ws_floats) *ws_floats = plan_gemm(mode, g).ws_floats;
    return SQD_OK;
}

// Rows of BatchNorm partials sqd_conv_fwd writes into `stats` for this geometry under the current plan ([rows][K][2] floats:
// per-channel sum and sum of squares of a tile of output rows, the layout sqd_bn_train_fwd's finalize reads): ceil(M / tile
// rows) — ceil(M / 64) when the plan splits the reduction (the sum over the splits takes the statistics); the input-patch plans
// write one row per patch.  Never more than max(ceil(M/64), N * ceil(Ho/4) * ceil(Wo/16)) rows: size `stats` for that when the plan may still change.
extern "C" int sqd_conv_fwd_stats_rows(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo) {
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    const GemmPlan p = plan_gemm(0, g);
    if (p.z > 1) return (N * Ho * Wo + RED_ROWS - 1) / RED_ROWS;                      // taken by the split-K sum (gemm_reduce_stats_kernel)
    if (p.halo) return N * ((H + p.bm / 16 - 1) / (p.bm / 16)) * ((W + 15) / 16);     // one row of partials per patch
    return (N * Ho * Wo + p.bm - 1) / p.bm;
}

// x [N,H,W,C], w [K,R,S,C], bias [K] or NULL -> y [N,Ho,Wo,K]; act: 0 none, 1 ReLU, 2 LeakyReLU(0.01).  C, K multiples of 16.
// stats (may be NULL): see sqd_conv_fwd_stats_rows — per-channel partial sums of y for the BatchNorm that follows.
extern "C" int sqd_conv_fwd(const float *x, const float *w, const float *bias, float *y, float *ws, float *stats, int N, int H, int W,
                            int C, int K, int R, int S, int stride, int pad, int Ho, int Wo, int act, void *stream) {
    SQD_CHECK_ARG(x && w && y, "sqd_conv_fwd: null pointer");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_fwd", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(C % 4 == 0 && K % 4 == 0, "sqd_conv_fwd: C=%d and K=%d must be multiples of 4", C, K);
    if (launch_gemm(0, x, w, bias, y, ws, g, act, stream, stats)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_conv_fwd");
    return SQD_OK;
}

// sqd_conv_fwd for plans on two-term fp16 operands (sqd_conv_set_plan bk + 4096): amax_x / amax_w = device scalars holding max |x| / max |w|
// (bit pattern of a non-negative float; any upper bound is safe, a tighter one is more accurate) — written by the producer of the tensor
// (the `amax` outputs of the BatchNorm / element-wise kernels, sqd_amax, sqd_amax_multi).  Other plans ignore them (NULL allowed).
// amax_y (any plan; may be NULL; cleared by the caller): the epilogue records max |y| for the next convolution.
extern "C" int sqd_conv_fwd_scaled(const float *x, const float *w, const float *bias, float *y, float *ws, float *stats, const float *amax_x,
                                   const float *amax_w, float *amax_y, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho,
                                   int Wo, int act, void *stream) {
    SQD_CHECK_ARG(x && w && y, "sqd_conv_fwd_scaled: null pointer");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_fwd_scaled", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(C % 4 == 0 && K % 4 == 0, "sqd_conv_fwd_scaled: C=%d and K=%d must be multiples of 4", C, K);
    if (launch_gemm(0, x, w, bias, y, ws, g, act, stream, stats, BnBwdSrc{nullptr, nullptr, nullptr, nullptr, 0}, OpScale{amax_x, amax_w, (unsigned *)amax_y})) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_conv_fwd_scaled");
    return SQD_OK;
}

// g, y, out: n floats (same memory order; out may alias g) -> out = g * act'(y); act 1 ReLU, 2 LeakyReLU(0.01)
extern "C" int sqd_act_bwd(const float *g, const float *y, float *out, int64_t n, int act, void *stream) {
    return sqd_act_bwd_amax(g, y, out, n, act, nullptr, stream);
}
// ... and amax_out (may be NULL; cleared by the caller): the bit pattern of max |out|
extern "C" int sqd_act_bwd_amax(const float *g, const float *y, float *out, int64_t n, int act, float *amax_out, void *stream) {
    SQD_CHECK_ARG(g && y && out && n > 0 && (act == 1 || act == 2), "sqd_act_bwd: bad arguments");
    SQD_CHECK_ARG(((uintptr_t)g & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)out & 15) == 0, "sqd_act_bwd: 16-byte aligned pointers");
    (void)hipGetLastError();
    const size_t n4 = (size_t)n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 1 ? 1 : ((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256));
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, y, out, n4, (size_t)n, act, (unsigned *)amax_out);
    SQD_CHECK_LAUNCH("sqd_act_bwd");
    return SQD_OK;
}

// dy [N,Ho,Wo,K], w [K,R,S,C] -> dx [N,H,W,C] (+ addend [N,H,W,C] when not NULL: the gradient arriving over a second path,
// e.g. the residual branch, is added in the epilogue instead of by a separate pass)
extern "C" int sqd_conv_dgrad(const float *dy, const float *w, const float *addend, float *dx, float *ws, int N, int H, int W, int C,
                              int K, int R, int S, int stride, int pad, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(dy && w && dx, "sqd_conv_dgrad: null pointer");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_dgrad", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(K % 4 == 0 && C % 4 == 0, "sqd_conv_dgrad: K=%d and C=%d must be multiples of 4", K, C);
    SQD_CHECK_ARG(addend != dx, "sqd_conv_dgrad: addend must not alias dx");
    if (launch_gemm(1, dy, w, addend, dx, ws, g, 0, stream)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_conv_dgrad");
    return SQD_OK;
}

// Rows of BatchNorm-backward partials sqd_conv_dgrad_bn writes for this geometry under the current plan ([rows][C][2] floats: per-channel
// sum dz and sum dz * xhat of a tile of input pixels): M-tiles x stride classes, one row per patch for the input-patch plans, ceil(N*H*W / 64) when the
// plan splits the reduction (the sum over the splits takes them: gemm_reduce_stats_kernel).
extern "C" int sqd_conv_dgrad_stats_rows(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo) {
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    const GemmPlan p = plan_gemm(1, g);
    if (p.z > 1) return (N * H * W + RED_ROWS - 1) / RED_ROWS;
    if (p.halo) return N * ((H + p.bm / 16 - 1) / (p.bm / 16)) * ((W + 15) / 16);
    const int Mcls = N * ((H + stride - 1) / stride) * ((W + stride - 1) / stride);
    return stride * stride * ((Mcls + p.bm - 1) / p.bm);
}

// sqd_conv_dgrad + the partial sums of the BatchNorm backward whose output act(BN(bn_x)) this convolution differentiates (dx, i.e.
// dgrad + addend, must be the COMPLETE gradient of that output): bn_x [N,H,W,C] the BatchNorm's input, bn_mask its sign bytes
// ([N*H*W*C/4], NULL without activation), bn_mean / bn_rstd [C], bn_act 0 | 1 (ReLU) | 2 (LeakyReLU 0.01)
// -> stats [sqd_conv_dgrad_stats_rows][C][2], consumed by sqd_bn_train_bwd_pre.  stats == NULL: plain sqd_conv_dgrad.
extern "C" int sqd_conv_dgrad_bn(const float *dy, const float *w, const float *addend, float *dx, float *ws, const float *bn_x,
                                 const unsigned char *bn_mask, const float *bn_mean, const float *bn_rstd, int bn_act, float *stats, int N,
                                 int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(dy && w && dx, "sqd_conv_dgrad_bn: null pointer");
    SQD_CHECK_ARG(!stats || (bn_x && bn_mean && bn_rstd && bn_act >= 0 && bn_act <= 2 && (bn_mask || bn_act == 0)),
                  "sqd_conv_dgrad_bn: the statistics need bn_x, bn_mean, bn_rstd and (for ReLU / LeakyReLU) bn_mask");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_dgrad_bn", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(K % 4 == 0 && C % 4 == 0, "sqd_conv_dgrad_bn: K=%d and C=%d must be multiples of 4", K, C);
    SQD_CHECK_ARG(addend != dx, "sqd_conv_dgrad_bn: addend must not alias dx");
    const BnBwdSrc bnb = {stats ? bn_x : nullptr, bn_mask, bn_mean, bn_rstd, bn_act};
    if (launch_gemm(1, dy, w, addend, dx, ws, g, 0, stream, stats, bnb)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_conv_dgrad_bn");
    return SQD_OK;
}

// sqd_conv_dgrad_bn for plans on two-term fp16 operands: amax_dy / amax_w as in sqd_conv_fwd_scaled.
extern "C" int sqd_conv_dgrad_scaled(const float *dy, const float *w, const float *addend, float *dx, float *ws, const float *bn_x,
                                     const unsigned char *bn_mask, const float *bn_mean, const float *bn_rstd, int bn_act, float *stats,
                                     const float *amax_dy, const float *amax_w, float *amax_dx, int N, int H, int W, int C, int K, int R, int S,
                                     int stride, int pad, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(dy && w && dx, "sqd_conv_dgrad_scaled: null pointer");
    SQD_CHECK_ARG(!stats || (bn_x && bn_mean && bn_rstd && bn_act >= 0 && bn_act <= 2 && (bn_mask || bn_act == 0)),
                  "sqd_conv_dgrad_scaled: the statistics need bn_x, bn_mean, bn_rstd and (for ReLU / LeakyReLU) bn_mask");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_dgrad_scaled", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(K % 4 == 0 && C % 4 == 0, "sqd_conv_dgrad_scaled: K=%d and C=%d must be multiples of 4", K, C);
    SQD_CHECK_ARG(addend != dx, "sqd_conv_dgrad_scaled: addend must not alias dx");
    const BnBwdSrc bnb = {stats ? bn_x : nullptr, bn_mask, bn_mean, bn_rstd, bn_act};
    if (launch_gemm(1, dy, w, addend, dx, ws, g, 0, stream, stats, bnb, OpScale{amax_dy, amax_w, (unsigned *)amax_dx})) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_conv_dgrad_scaled");
    return SQD_OK;
}

struct WgradPlan {
    bool direct;
    int kt, ct, tp, splits, px_per_wave;
    int shared = 0;                 // 1..4: shared-operand kernel, block 128x128 / 64x128 / 128x64 / 64x64 (filters x channels); 0: not
    int split3 = 0;                 // with shared != 0: the three-term bf16 kernel on the same blocks (impl 3) instead of fp32 (impl 2)
    int px_per_split = 0;
    int direct3 = 0;                // 1..: three-term bf16 operands straight from memory (impl 6), register tile variant
    int h2 = 0;                     // with direct3: two fp16 terms of the scaled operands (impl 7) instead of three bf16 terms
    int rows = 0;                   // row-window kernel (impl 4): `splits` workgroups of `steps_per_wg` (image, column chunk, row) steps
    int steps_per_wg = 0, nchunks = 0;
};
// (filters, channels, taps) the row-window kernel is instantiated for
static bool rows_shape(int C, int K, int R, int S) {
    if (R != S || C % 16 || K % 16) return false;
    const int kt = K / 16, ct = C / 16;
    if (R == 3) return (kt == 1 && (ct == 6 || ct == 2 || ct == 1)) || (kt == 2 && (ct == 2 || ct == 1)) || (kt == 4 && ct == 1);
    if (R == 4) return (kt == 1 && ct == 2) || (kt == 4 && ct == 1);
    return false;
}
static int rows_pxw(int) { return 64; }                       // pixels per step of the row-window kernel's instantiations
static void shared_block(int variant, int &tk, int &tc) {
    tk = (variant == 1 || variant == 3) ? 128 : 64;
    tc = (variant == 1 || variant == 2) ? 128 : 64;
}
// measured weight-gradient plans (sqd_conv_wgrad_set_plan): geometry -> (impl 0 = LDS-tiled / 1 = direct-operand, splits)
typedef std::tuple<int, int, int, int, int, int, int> WPlanKey;
static std::map<WPlanKey, std::pair<int, int>> &wplan_table() {
    static std::map<WPlanKey, std::pair<int, int>> t;
    return t;
}
static bool wplan_lookup(int N, int Ho, int Wo, int C, int K, int R, int S, int &impl, int &splits) {
    std::lock_guard<std::mutex> lk(plan_mutex());
    auto it = wplan_table().find(WPlanKey(N, Ho, Wo, C, K, R, S));
    if (it == wplan_table().end()) return false;
    impl = it->second.first;
    splits = it->second.second;
    return true;
}

// register tile (x 32 filters, x 32 channels per lane-row) of the impl-6 kernel's variants
static void direct3_tile(int variant, int &kt, int &ct) {
    static const int T[8][2] = {{2, 2}, {2, 1}, {1, 2}, {4, 2}, {2, 4}, {4, 4}, {4, 2}, {2, 2}};     // variants 7, 8: 8-wave workgroups
    kt = T[variant - 1][0];
    ct = T[variant - 1][1];
}
constexpr int DIRECT3_VARIANTS = 8;
static int direct3_waves(int variant) { return variant >= 7 ? 8 : 4; }
static WgradPlan plan_wgrad_direct(int N, int Ho, int Wo, int C, int K, int R, int S, int stride = 1, bool pairs_ok = true) {
    WgradPlan p;
    const int M = N * Ho * Wo;
    int m_impl = -1, m_splits = 0;
    const bool measured = wplan_lookup(N, Ho, Wo, C, K, R, S, m_impl, m_splits);
    if (measured && (m_impl & 15) == 4) {
        if (stride == 1) {                                       // row-window kernel: `m_splits` workgroups
            p.direct = false;
            p.kt = p.ct = p.tp = 1;
            p.px_per_wave = 0;
            p.rows = 1;
            p.nchunks = (Wo + rows_pxw(C) - 1) / rows_pxw(C);
            const int total = N * p.nchunks * Ho;
            int sp = std::min(std::max(m_splits, 1), total);
            const int64_t wsz = (int64_t)K * R * S * C;
            while (sp > 1 && (int64_t)sp * wsz * 4 > (64ll << 20)) --sp;
            p.steps_per_wg = (total + sp - 1) / sp;
            p.splits = (total + p.steps_per_wg - 1) / p.steps_per_wg;
            return p;
        }
        m_impl = 1;             // the plan table does not key on the stride: a strided convolution of the same output geometry runs the
    }                           // direct kernel on (at most) the same number of splits, inside the workspace the plan query reported
    if (measured && ((m_impl & 15) == 6 || (m_impl & 15) == 7)) {
        int kt3, ct3;
        p.h2 = (m_impl & 15) == 7;
        direct3_tile(((m_impl >> 4) & 15) + 1, kt3, ct3);
        if (pairs_ok && K % (32 * kt3) == 0 && C % (32 * ct3) == 0) {
            p.direct = false;
            p.kt = p.ct = p.tp = 1;
            p.direct3 = ((m_impl >> 4) & 15) + 1;
            int sp = m_splits;
            const int max_by_px = (M + 255) / 256;                   // >= 64 pixels per wave
            if (sp > max_by_px) sp = max_by_px;
            const int64_t wsz = (int64_t)K * R * S * C;
            while (sp > 1 && (int64_t)sp * wsz * 4 > (64ll << 20)) --sp;
            if (sp < 1) sp = 1;
            p.splits = sp;
            const int nw = direct3_waves(p.direct3);
            p.px_per_wave = ((M + sp * nw - 1) / (sp * nw) + 15) / 16 * 16;
            return p;
        }
        m_impl = 1;             // an odd Wo under a strided / padded convolution of the same output geometry: the fp32 direct kernel, same splits
    }
    if (measured && ((m_impl & 15) == 2 || (m_impl & 15) == 3)) {      // shared-operand kernels (measured plans only)
        p.split3 = (m_impl & 15) == 3;
        p.direct = false;
        p.kt = p.ct = p.tp = 1;
        p.px_per_wave = 0;
        p.shared = ((m_impl >> 4) & 15) + 1;
        int sp = m_splits;
        const int max_by_px = (M + 63) / 64;                     // >= 64 pixels per split
        if (sp > max_by_px) sp = max_by_px;
        const int64_t wsz = (int64_t)K * R * S * C;
        while (sp > 1 && (int64_t)sp * wsz * 4 > (64ll << 20)) --sp;
        if (sp < 1) sp = 1;
        p.px_per_split = ((M + sp - 1) / sp + 31) / 32 * 32;
        p.splits = (M + p.px_per_split - 1) / p.px_per_split;
        return p;
    }
    const int fkt = measured ? (m_impl >> 4) & 15 : 0, fct = measured ? (m_impl >> 8) & 15 : 0;   // measured register-tile shape (0: default)
    if (measured) m_impl &= 1;
    // the direct kernel wins where pixels are many and channels few (operand re-reads stay in L2); the LDS-tiled
    // kernel where K*C is large and the pixel count small
    p.direct = C % 16 == 0 && K % 16 == 0 && M >= 2048 && C <= 1024;
    if (measured) p.direct = m_impl == 1 && C % 16 == 0 && K % 16 == 0;
    if (!p.direct) {                                             // (channel counts below / not divisible by 16: the LDS-tiled kernel)
        p.kt = p.ct = p.tp = p.splits = 1;
        p.px_per_wave = 0;
        return p;
    }
    p.kt = K % 64 == 0 ? 4 : K % 32 == 0 ? 2 : 1;
    p.ct = C % 64 == 0 ? 4 : C % 32 == 0 ? 2 : 1;
    if (fkt) p.kt = fkt;                                         // a smaller register tile = more resident waves (the 4x4 tile's 184
    if (fct) p.ct = fct;                                         // VGPRs and 66 KB of LDS leave two workgroups per CU)
    p.tp = 1;     // (a filter row of 3 taps per wave was measured slower on every config-B layer — 140 us vs 102 us: DESIGN.md §3.4)
    const int groups = (K / (16 * p.kt)) * (C / (16 * p.ct)) * (R * S / p.tp);
    int sp = (512 + groups - 1) / groups;                        // ~512 workgroups of 4 waves (measured plans mostly land at
                                                                 // a quarter to a half of the ~1k first assumed)
    if (measured) sp = m_splits;
    const int max_by_px = (M + 255) / 256;                       // >= 64 pixels per wave
    if (sp > max_by_px) sp = max_by_px;
    const int64_t wsz = (int64_t)K * R * S * C;
    while (sp > 1 && (int64_t)sp * wsz * 4 > (64ll << 20)) --sp;
    if (sp < 1) sp = 1;
    int ppw = (M + sp * 4 - 1) / (sp * 4);
    ppw = (ppw + 3) / 4 * 4;
    p.splits = sp;
    p.px_per_wave = ppw;
    return p;
}

extern "C" int sqd_conv_wgrad_plan(int N, int Ho, int Wo, int C, int K, int R, int S, int *splits, int64_t *part_floats) {
    const int M = N * Ho * Wo;
    const WgradPlan dp = plan_wgrad_direct(N, Ho, Wo, C, K, R, S);
    if (dp.direct || dp.shared || dp.rows || dp.direct3) {
        if (splits) *splits = dp.splits;
        if (part_floats) *part_floats = (int64_t)dp.splits * K * R * S * C;
        return SQD_OK;
    }
    const int bm = K >= 128 ? 128 : 64, bn = C >= 128 ? 128 : 64;
    const int tiles = ((K + bm - 1) / bm) * ((C + bn - 1) / bn) * R * S;
    int sp = (1536 + tiles - 1) / tiles;                         // aim at ~1.5k workgroups
    {
        int m_impl, m_splits;
        if (wplan_lookup(N, Ho, Wo, C, K, R, S, m_impl, m_splits)) sp = m_splits;
    }
    const int max_by_px = (M + 255) / 256;                      // at least 256 pixels per split
    if (sp > max_by_px) sp = max_by_px;
    const int64_t wsz = (int64_t)K * R * S * C;
    while (sp > 1 && (int64_t)sp * wsz * 4 > (64ll << 20)) --sp; // partial buffer <= 64 MB
    if (sp < 1) sp = 1;
    if (splits) *splits = sp;
    if (part_floats) *part_floats = (int64_t)sp * wsz;
    return SQD_OK;
}

// Register a measured weight-gradient plan: impl 1 = direct-operand kernel (+ 16*kt + 256*ct: its register tile covers 16*kt
// filters x 16*ct channels, kt, ct in {1,2,4}; 0 = the widest that divides), 0 = LDS-tiled kernel, -1 = clear; `splits` pixel
// ranges (clamped by the library to >= 256 pixels per range and a 64 MB partial buffer).  sqd_conv_wgrad_plan reports the
// resulting workspace.
extern "C" int sqd_conv_wgrad_set_plan(int N, int Ho, int Wo, int C, int K, int R, int S, int impl, int splits) {
    std::lock_guard<std::mutex> lk(plan_mutex());
    if (impl < 0) {
        wplan_table().erase(WPlanKey(N, Ho, Wo, C, K, R, S));
        return SQD_OK;
    }
    const int kt = (impl >> 4) & 15, ct = (impl >> 8) & 15;      // optional register-tile shape of the direct kernel: 16*kt x 16*ct
    if ((impl & 15) == 4) {                                      // row-window kernel (stride 1; few channels): `splits` workgroups
        SQD_CHECK_ARG(impl == 4 && rows_shape(C, K, R, S) && splits >= 1 && splits <= 65535,
                      "sqd_conv_wgrad_set_plan: the row-window kernel is not built for C=%d K=%d %dx%d", C, K, R, S);
        wplan_table()[WPlanKey(N, Ho, Wo, C, K, R, S)] = std::make_pair(impl, splits);
        return SQD_OK;
    }
    if ((impl & 15) == 6 || (impl & 15) == 7) {                  // three-term bf16 (6) / two-term fp16 (7) operands straight from memory: + 16 * variant
        const int variant = (impl >> 4) + 1;
        int kt3 = 0, ct3 = 0;
        if (variant >= 1 && variant <= DIRECT3_VARIANTS) direct3_tile(variant, kt3, ct3);
        SQD_CHECK_ARG(variant >= 1 && variant <= DIRECT3_VARIANTS && K % (32 * kt3) == 0 && C % (32 * ct3) == 0 && splits >= 1 && splits <= 65535 &&
                          (Wo % 2 == 0 || (R == 1 && S == 1)),
                      "sqd_conv_wgrad_set_plan: the three-term direct kernel (variant %d) does not fit K=%d, C=%d, Wo=%d", variant, K, C, Wo);
        wplan_table()[WPlanKey(N, Ho, Wo, C, K, R, S)] = std::make_pair(impl, splits);
        return SQD_OK;
    }
    if ((impl & 15) == 2 || (impl & 15) == 3) {                  // shared-operand kernels: impl 2 (fp32) | 3 (three-term bf16) + 16 * variant
        const int variant = (impl >> 4) + 1;
        int tk, tc;
        shared_block(variant, tk, tc);
        SQD_CHECK_ARG(variant >= 1 && variant <= 4 && K % tk == 0 && C % tc == 0 && splits >= 1 && splits <= 65535,
                      "sqd_conv_wgrad_set_plan: shared-operand block %d does not fit K=%d, C=%d", variant, K, C);
        wplan_table()[WPlanKey(N, Ho, Wo, C, K, R, S)] = std::make_pair(impl, splits);
        return SQD_OK;
    }
    SQD_CHECK_ARG((impl & ~0xff1) == 0 && splits >= 1 && splits <= 65535, "sqd_conv_wgrad_set_plan: bad plan impl=%d splits=%d", impl, splits);
    SQD_CHECK_ARG((impl & 1) == 0 || (C % 16 == 0 && K % 16 == 0), "sqd_conv_wgrad_set_plan: the direct kernel needs C, K multiples of 16");
    SQD_CHECK_ARG((kt == 0 && ct == 0) || ((impl & 1) && (kt == 1 || kt == 2 || kt == 4) && (ct == 1 || ct == 2 || ct == 4) &&
                                          K % (16 * kt) == 0 && C % (16 * ct) == 0),
                  "sqd_conv_wgrad_set_plan: register tile %dx%d does not fit K=%d, C=%d", kt, ct, K, C);
    wplan_table()[WPlanKey(N, Ho, Wo, C, K, R, S)] = std::make_pair(impl, splits);
    return SQD_OK;
}

// The kernel family sqd_conv_wgrad launches for this convolution under the registered plan: the plan table keys on the OUTPUT geometry, so a
// strided / padded layer that shares a key with a stride-1 layer may not be able to run the registered kernel (the row-window kernel is
// stride 1 only; the direct-operand split kernels need an even Wo unless the convolution is a plain 1x1) and falls back to the fp32
// direct kernel (1) inside the same workspace.  Returns impl & 15 of what runs: 0 LDS-tiled, 1 direct fp32, 2 / 3 shared-operand,
// 4 row-window, 6 three-term direct, 7 two-term direct.  The plan timing uses it to discard candidates that would not run as named.
extern "C" int sqd_conv_wgrad_effective_impl(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo) {
    (void)H; (void)W;
    const bool flat = R == 1 && S == 1 && stride == 1 && pad == 0;
    const WgradPlan dp = plan_wgrad_direct(N, Ho, Wo, C, K, R, S, stride, flat || Wo % 2 == 0);
    if (dp.direct3) return dp.h2 ? 7 : 6;
    if (dp.rows) return 4;
    if (dp.shared) return dp.split3 ? 3 : 2;
    return dp.direct ? 1 : 0;
}

// dy [N,Ho,Wo,K], x [N,H,W,C] -> dw [K,R,S,C]; dbias [K] (may be NULL); part: workspace of sqd_conv_wgrad_plan floats
// (+ bias scratch appended when dbias is requested: max(ceil(M/1024), splits) * K floats)
// ---------------------------------------------------------------------------------------------------
// Weight gradient of a 1x1 convolution over a HANDFUL of rows (M <= 32): the bins regressor's Linear layers run as 1x1 convolutions
// whose "pixels" are the batch's 12 rows (reference networks/depth_decoder_QTR.py:22-26,49), so dW [K][C] = sum_m dy[m][k] x[m][c] is
// M outer products and 4 K C bytes of output — 8.4 MB for the 2048 -> 1024 layer, which the pixel-split MFMA kernels above wrote in
// 30 us (1.7 TFLOP/s: a tile of 64 x 64 filters for 12 pixels).  Here a thread owns a float4 of input channels for KB filters: x is read
// once per block of KB filters (coalesced), dy[m][k] is wave-uniform (scalar loads), dW leaves as coalesced 16-byte stores; the bias
// gradient is the column sum of dy.  Fixed summation order over m.
// ---------------------------------------------------------------------------------------------------
template <int KB>
__global__ __launch_bounds__(256) void conv_wgrad_fewrows_kernel(const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ dw,
                                                                 float *__restrict__ dbias, int M, int C, int K) {
    const int c4 = blockIdx.x * 256 + threadIdx.x, k0 = blockIdx.y * KB;
    if (dbias && blockIdx.x == 0 && threadIdx.x < KB && k0 + (int)threadIdx.x < K) {
        float sb = 0.f;
        for (int m = 0; m < M; ++m) sb += dy[(size_t)m * K + k0 + threadIdx.x];
        dbias[k0 + threadIdx.x] = sb;
    }
    if (c4 * 4 >= C) return;
    float4 acc[KB];
#pragma unroll
    for (int q = 0; q < KB; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = 0; m < M; ++m) {
        const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)m * C)[c4];
        const float *__restrict__ dr = dy + (size_t)m * K + k0;          // wave-uniform: scalar loads
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            const float d = k0 + q < K ? dr[q] : 0.f;
            acc[q].x = fmaf(d, xv.x, acc[q].x); acc[q].y = fmaf(d, xv.y, acc[q].y);
            acc[q].z = fmaf(d, xv.z, acc[q].z); acc[q].w = fmaf(d, xv.w, acc[q].w);
        }
    }
#pragma unroll
    for (int q = 0; q < KB; ++q)
        if (k0 + q < K) reinterpret_cast<float4 *>(dw + (size_t)(k0 + q) * C)[c4] = acc[q];
}
constexpr int FEWROWS_MAX = 32;

static int conv_wgrad_impl(const float *dy, const float *x, float *dw, float *dbias, float *part, int N, int H, int W, int C,
                           int K, int R, int S, int stride, int pad, int Ho, int Wo, void *stream, bool reduce_dw, int *splits_out,
                           OpScale sc = OpScale{nullptr, nullptr, nullptr}) {
    SQD_CHECK_ARG(dy && x && dw && part, "sqd_conv_wgrad: null pointer");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_wgrad", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(C % 4 == 0 && K % 4 == 0, "sqd_conv_wgrad: C=%d and K=%d must be multiples of 4", C, K);
    if (R == 1 && S == 1 && stride == 1 && pad == 0 && N * Ho * Wo <= FEWROWS_MAX) {
        // a handful of rows (the bins regressor's Linear layers): M outer products, no pixel splits — whatever plan is registered
        float *out = reduce_dw ? dw : part;
        if (splits_out) *splits_out = 1;
        (void)hipGetLastError();
        hipLaunchKernelGGL(conv_wgrad_fewrows_kernel<8>, dim3((C / 4 + 255) / 256, (K + 7) / 8), dim3(256), 0, (hipStream_t)stream, dy, x, out, dbias,
                           N * Ho * Wo, C, K);
        SQD_CHECK_LAUNCH("sqd_conv_wgrad");
        return SQD_OK;
    }
    int splits;
    int64_t pf;
    sqd_conv_wgrad_plan(N, Ho, Wo, C, K, R, S, &splits, &pf);
    const int M = N * Ho * Wo;
    int pps = (M + splits - 1) / splits;
    pps = ((pps + BK - 1) / BK) * BK;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    const bool flat = R == 1 && S == 1 && stride == 1 && pad == 0;
    const WgradPlan dp = plan_wgrad_direct(N, Ho, Wo, C, K, R, S, stride, flat || Wo % 2 == 0);
    if (dp.direct3) {
        int kt3, ct3;
        direct3_tile(dp.direct3, kt3, ct3);
        const dim3 grid(((K / (32 * kt3)) * (C / (32 * ct3)) * R * S * dp.splits + 7) / 8 * 8);
        // (an fp32 fallback of a two-term plan — odd Wo under a strided layer of the same output geometry — never reaches this branch)
        SQD_CHECK_ARG(!dp.h2 || (sc.amax_a && sc.amax_b), "sqd_conv_wgrad: the plan registered for this geometry runs on two-term fp16 operands and "
                      "needs the max |.| of dy and x (sqd_conv_wgrad_scaled)");
#define LAUNCH_W6(KT, CT, NW)                                                                                                          \
    do {                                                                                                                               \
        if (dp.h2 && flat) hipLaunchKernelGGL((conv_wgrad_direct3_kernel<KT, CT, true, NW, true>), grid, dim3(NW * 64), 0, st, dy, x, part, g, dp.px_per_wave, dp.splits, sc); \
        else if (dp.h2) hipLaunchKernelGGL((conv_wgrad_direct3_kernel<KT, CT, false, NW, true>), grid, dim3(NW * 64), 0, st, dy, x, part, g, dp.px_per_wave, dp.splits, sc);  \
        else if (flat) hipLaunchKernelGGL((conv_wgrad_direct3_kernel<KT, CT, true, NW>), grid, dim3(NW * 64), 0, st, dy, x, part, g, dp.px_per_wave, dp.splits, sc); \
        else hipLaunchKernelGGL((conv_wgrad_direct3_kernel<KT, CT, false, NW>), grid, dim3(NW * 64), 0, st, dy, x, part, g, dp.px_per_wave, dp.splits, sc);    \
    } while (0)
        switch (dp.direct3) {
            case 1: LAUNCH_W6(2, 2, 4); break;
            case 2: LAUNCH_W6(2, 1, 4); break;
            case 3: LAUNCH_W6(1, 2, 4); break;
            case 4: LAUNCH_W6(4, 2, 4); break;
            case 5: LAUNCH_W6(2, 4, 4); break;
            case 6: LAUNCH_W6(4, 4, 4); break;
            case 7: LAUNCH_W6(4, 2, 8); break;
            default: LAUNCH_W6(2, 2, 8); break;
        }
    } else if (dp.rows) {
        const dim3 grid((dp.splits + 7) / 8 * 8);
        float *bias_part = dbias ? part + (size_t)dp.splits * K * R * S * C : nullptr;   // [splits][K]
#define LAUNCH_WR(KT, CT, RR, IG, PG, PXW)                                                                                     \
    hipLaunchKernelGGL((conv_wgrad_rows_kernel<KT, CT, RR, IG, PG, PXW>), grid, dim3(IG * PG * 64), 0, st, dy, x, part, g,         \
                       dp.steps_per_wg, dp.nchunks, dp.splits, bias_part)
        const int kt = K / 16, ct = C / 16;
        if (R == 3 && kt == 1 && ct == 6) LAUNCH_WR(1, 6, 3, 4, 2, 64);
        else if (R == 3 && kt == 1 && ct == 2) LAUNCH_WR(1, 2, 3, 2, 4, 64);
        else if (R == 3 && kt == 1 && ct == 1) LAUNCH_WR(1, 1, 3, 1, 8, 64);
        else if (R == 3 && kt == 2 && ct == 2) LAUNCH_WR(2, 2, 3, 2, 4, 64);
        else if (R == 3 && kt == 2 && ct == 1) LAUNCH_WR(2, 1, 3, 1, 8, 64);
        else if (R == 3 && kt == 4 && ct == 1) LAUNCH_WR(4, 1, 3, 1, 8, 64);
        else if (R == 4 && kt == 1 && ct == 2) LAUNCH_WR(1, 2, 4, 4, 2, 64);
        else LAUNCH_WR(4, 1, 4, 4, 2, 64);
    } else if (dp.shared) {
        int tk, tc;
        shared_block(dp.shared, tk, tc);
        const dim3 grid(((K / tk) * (C / tc) * R * S * dp.splits + 7) / 8 * 8);
#define LAUNCH_WS(WK, WC, KT, CT) \
    hipLaunchKernelGGL((conv_wgrad_shared_kernel<WK, WC, KT, CT>), grid, dim3(256), 0, st, dy, x, part, g, dp.px_per_split, dp.splits)
#define LAUNCH_W3(WK, WC, WTK, WTC) \
    hipLaunchKernelGGL((conv_wgrad_split3_kernel<WK, WC, WTK, WTC>), grid, dim3(256), 0, st, dy, x, part, g, dp.px_per_split, dp.splits)
        if (dp.split3) {
            if (dp.shared == 1) LAUNCH_W3(2, 2, 2, 2);
            else if (dp.shared == 2) LAUNCH_W3(1, 4, 2, 1);
            else if (dp.shared == 3) LAUNCH_W3(4, 1, 1, 2);
            else LAUNCH_W3(2, 2, 1, 1);
        } else if (dp.shared == 1) LAUNCH_WS(2, 2, 4, 4);
        else if (dp.shared == 2) LAUNCH_WS(1, 4, 4, 2);
        else if (dp.shared == 3) LAUNCH_WS(4, 1, 2, 4);
        else LAUNCH_WS(2, 2, 2, 2);
    } else if (dp.direct) {
        const int kgroups = K / (16 * dp.kt);
        const dim3 grid((kgroups * (C / (16 * dp.ct)) * dp.splits * (R * S / dp.tp) + 7) / 8 * 8);
        float *bias_part = dbias ? part + (size_t)dp.splits * K * R * S * C : nullptr;   // [splits][K]
#define LAUNCH_WD(KT, CT)                                                                                                       \
    hipLaunchKernelGGL((conv_wgrad_direct_kernel<KT, CT, 1>), grid, dim3(256), 0, st, dy, x, part, g, dp.px_per_wave, kgroups, \
                       dp.splits, bias_part)
        switch (dp.kt * 8 + dp.ct) {
            case 4 * 8 + 4: LAUNCH_WD(4, 4); break;
            case 4 * 8 + 2: LAUNCH_WD(4, 2); break;
            case 4 * 8 + 1: LAUNCH_WD(4, 1); break;
            case 2 * 8 + 4: LAUNCH_WD(2, 4); break;
            case 2 * 8 + 2: LAUNCH_WD(2, 2); break;
            case 2 * 8 + 1: LAUNCH_WD(2, 1); break;
            case 1 * 8 + 4: LAUNCH_WD(1, 4); break;
            case 1 * 8 + 2: LAUNCH_WD(1, 2); break;
            default: LAUNCH_WD(1, 1); break;
        }
    } else {
        const bool bigk = K >= 128, bigc = C >= 128;
        const int bm = bigk ? 128 : 64, bn = bigc ? 128 : 64;
        const dim3 grid(((K + bm - 1) / bm) * ((C + bn - 1) / bn), splits, R * S);
        if (bigk && bigc) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, dim3(256), 0, st, dy, x, part, g, pps);
        else if (bigk) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, dim3(256), 0, st, dy, x, part, g, pps);
        else if (bigc) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, dim3(256), 0, st, dy, x, part, g, pps);
        else hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, dim3(256), 0, st, dy, x, part, g, pps);
    }
    const size_t wsz = (size_t)K * R * S * C;
    // (leaving the ~100 reductions of a backward pass to two or three multi-task launches at its end was measured: bit-identical and
    // 0.7 ms SLOWER per step — reduced on the spot the partials of most layers are still in the 256 MB Infinity Cache)
    if (dp.direct || dp.shared || dp.rows || dp.direct3) splits = dp.splits;    // (a strided convolution under a row-window plan: see plan_wgrad_direct)
    if (splits_out) *splits_out = splits;
    const bool bias_splits = dbias && ((dp.direct && !dp.shared) || dp.rows);       // the kernel left [splits][K] bias partials behind the filter partials
    if (reduce_dw && bias_splits) {
        const unsigned nb1 = (unsigned)((wsz / 4 + 15) / 16);
        hipLaunchKernelGGL(split_reduce2_kernel, dim3(nb1 + (unsigned)((K / 4 + 15) / 16)), dim3(256), 0, st, part, dw, wsz,
                           part + (size_t)splits * wsz, dbias, (size_t)K, splits, (int)nb1);
    } else if (reduce_dw) {
        hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)((wsz / 4 + 15) / 16)), dim3(256), 0, st, part, dw, wsz, splits);
    }
    if (bias_splits && !reduce_dw) {
        hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)((K / 4 + 15) / 16)), dim3(256), 0, st, part + (size_t)splits * wsz, dbias,
                           (size_t)K, splits);
    } else if (dbias && !bias_splits) {
        float *cpart = part + (size_t)splits * wsz;
        const int rpb = 1024, nblk = (M + rpb - 1) / rpb;
        int cpb = 1;
        while (cpb < 64 && cpb * 4 < K) cpb <<= 1;              // float4 columns per block (power of two <= 64)
        const int bands = (K / 4 + 63) / 64;
        hipLaunchKernelGGL(colsum_kernel, dim3(nblk, bands), dim3(256), 0, st, dy, cpart, M, K, rpb, cpb);
        hipLaunchKernelGGL(colsum_kernel, dim3(1, bands), dim3(256), 0, st, cpart, dbias, nblk, K, nblk, cpb);
    }
    SQD_CHECK_LAUNCH("sqd_conv_wgrad");
    return SQD_OK;
}

extern "C" int sqd_conv_wgrad(const float *dy, const float *x, float *dw, float *dbias, float *part, int N, int H, int W, int C,
                              int K, int R, int S, int stride, int pad, int Ho, int Wo, void *stream) {
    return conv_wgrad_impl(dy, x, dw, dbias, part, N, H, W, C, K, R, S, stride, pad, Ho, Wo, stream, true, nullptr);
}

// sqd_conv_wgrad / sqd_conv_wgrad_partials (splits != NULL: no final sum, see below) for plans on two-term fp16 operands (impl 7):
// amax_dy / amax_x as in sqd_conv_fwd_scaled.  Other plans ignore them.
extern "C" int sqd_conv_wgrad_scaled(const float *dy, const float *x, float *dw, float *dbias, float *part, const float *amax_dy,
                                     const float *amax_x, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                                     int *splits, void *stream) {
    return conv_wgrad_impl(dy, x, dw, dbias, part, N, H, W, C, K, R, S, stride, pad, Ho, Wo, stream, splits == nullptr, splits, OpScale{amax_dy, amax_x, nullptr});
}

// The same without the final sum over the pixel splits: part[0 .. *splits)[K*R*S*C] holds the partial filter gradients, dw is NOT
// written (dbias is).  The caller adds them later on the same stream — sqd_split_reduce, or as extra workgroups of the next
// BatchNorm-backward finalize launch (sqd_bn_train_bwd_pre_red): one launch less per layer, the same bits.
extern "C" int sqd_conv_wgrad_partials(const float *dy, const float *x, float *dw, float *dbias, float *part, int N, int H, int W, int C,
                                       int K, int R, int S, int stride, int pad, int Ho, int Wo, int *splits, void *stream) {
    SQD_CHECK_ARG(splits, "sqd_conv_wgrad_partials: null pointer");
    return conv_wgrad_impl(dy, x, dw, dbias, part, N, H, W, C, K, R, S, stride, pad, Ho, Wo, stream, false, splits);
}

// out[i] = sum_{s < splits} part[s * n + i], i < n (n a multiple of 4; fixed summation order)
extern "C" int sqd_split_reduce(const float *part, float *out, int64_t n, int splits, void *stream) {
    SQD_CHECK_ARG(part && out && n > 0 && n % 4 == 0 && splits >= 1, "sqd_split_reduce: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, (hipStream_t)stream, part, out, (size_t)n, splits);
    SQD_CHECK_LAUNCH("sqd_split_reduce");
    return SQD_OK;
}

// ---------------------------------------------------------------------------------------------------
// [rows][cols] -> [cols][rows] (fp32, 64 x 64 tiles through LDS, both sides coalesced), optionally with the column sums of each row
// tile on the way (colsum [ceil(rows / 64)][cols]: the bias gradient's partials).  Feeds the weight gradient of the wide 1x1 layers
// as a forward GEMM on transposed operands (nnkernels._wgrad_transposed): dW [K][C] = sum_m dY[m][k] X[m][c] reduces over the slow
// axis of both tensors; transposed, it is the forward problem "K pixels, M channels, C filters" and runs the three-term kernels.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void transpose2d_kernel(const float *__restrict__ src, float *__restrict__ dst, int rows, int cols,
                                                          float *__restrict__ colsum) {
    __shared__ float tile[64][65];
    __shared__ float cs[4][64];
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    const int R0 = blockIdx.y * 64, C0 = blockIdx.x * 64;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + 4 * i;
        const float v = (R0 + r < rows && C0 + c < cols) ? src[(size_t)(R0 + r) * cols + C0 + c] : 0.f;
        tile[r][c] = v;
        acc += v;
    }
    if (colsum) cs[r0][c] = acc;
    __syncthreads();
    if (colsum && r0 == 0 && C0 + c < cols) colsum[(size_t)blockIdx.y * cols + C0 + c] = ((cs[0][c] + cs[1][c]) + cs[2][c]) + cs[3][c];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + 4 * i;                       // column C0 + r of src = row of dst; c runs along the source rows
        if (C0 + r < cols && R0 + c < rows) dst[(size_t)(C0 + r) * rows + R0 + c] = tile[c][r];
    }
}
}  // namespace

// src [rows][cols] -> dst [cols][rows]; colsum (may be NULL): [ceil(rows / 64)][cols] partial column sums (sum them with sqd_colsum_multi)
extern "C" int sqd_transpose2d(const float *src, float *dst, int rows, int cols, float *colsum, void *stream) {
    SQD_CHECK_ARG(src && dst && src != dst && rows > 0 && cols > 0, "sqd_transpose2d: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(transpose2d_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, colsum);
    SQD_CHECK_LAUNCH("sqd_transpose2d");
    return SQD_OK;
}
