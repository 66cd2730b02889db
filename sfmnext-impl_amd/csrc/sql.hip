// sql.hip — the Self Query Layer (reference networks/layers.py:7-21, FullQueryLayer.forward) as fp32-MFMA
// kernels for gfx950:
//     y[b,q,n]       = sum_e K[b,q,e] * x[b,e,n]                    (energy maps, returned raw)
//     s[b,q,:]       = softmax over the N = h*w pixels of y[b,q,:]
//     summary[b,q,e] = sum_n s[b,q,n] * x[b,e,n]
//
// Forward: one pass over x.  A wavefront owns a run of pixel tiles; per tile it forms y^T (pixels x
// queries) with v_mfma_f32_16x16x4_f32, stores y, updates the running (max, sum) of an online softmax
// per query and feeds p = exp(y - max) — still sitting in the MFMA accumulator registers — straight
// back in as the B operand of the second MFMA (x * p, contraction over the pixels the accumulator rows
// index), so the N x Q probability matrix never exists in memory.  Per-workgroup partials
// (max, sum, unnormalised summary) are merged by a small second kernel, which also emits the
// log-sum-exp the backward needs.
//
// Backward: one pass over (x, y, g_y) in the transposed orientation (queries x pixels) so that the
// contraction over queries can again consume accumulator registers directly; the only layout change
// (for dK, a contraction over pixels) goes through a wave-private LDS tile.
//
// Roofline: HBM for the y traffic (4*Q B/px written forward, 8*Q B/px read backward), fp32 MFMA
// (157 TF) for the 4*Q*E flop/px forward — both are ~10-20 us at config B; the kernel exists to
// replace 5 ATen launches that move y four times.
#include "sqd_common.h"

namespace {
using namespace sqd;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// reduce over the four 16-lane groups (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// raw buffer accesses: the descriptor carries the extent of one image's planes, so a plane index beyond E (or Q) and —
// through a base offset of 2 GiB for lanes beyond the last pixel — a pixel beyond N read zeros / drop the store without a
// branch around every access
typedef int sql_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sql_rsrc(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ldb32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void stb32(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), r, byte_off, 0, 0);
}
constexpr unsigned SQL_OOB = 0x80000000u;

constexpr int PART_STRIDE_EXTRA = 2;   // per query: max, sum, then E summary values

// ---------------------------------------------------------------------------------------------------
// forward.  QT = ceil(Q/16) query tiles, ET = E/16 feature tiles (E multiple of 16), NT pixel tiles
// (of 16) per step; grid (chunks, B), 256 threads; each wave walks `tiles_per_wave` steps.
// ---------------------------------------------------------------------------------------------------
template <int QT, int ET, int NT>
__global__ __launch_bounds__(256) void sql_fwd_kernel(const float *__restrict__ x, const float *__restrict__ K,
                                                      float *__restrict__ y, float *__restrict__ part, int Q, int N,
                                                      int steps_per_wave, int nchunks, int xse, int xsn) {
    // x[b] is addressed as x[e * xse + n * xsn]: planar [E][N] (xse = N, xsn = 1) or pixel-major / channels-last [N][E]
    // (xse = 1, xsn = E) — the layout the producing convolution writes, so no layout copy precedes this kernel
    constexpr int E = ET * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float *xb = x + (size_t)b * E * N;
    const float *Kb = K + (size_t)b * Q * E;
    float *yb = y + (size_t)b * Q * N;
    const __amdgpu_buffer_rsrc_t x_r = sql_rsrc(xb, (unsigned)(E * N) * 4u);
    const int PX = NT * 16;                                   // pixels per step
    const int n_wave0 = (chunk * 4 + wave) * steps_per_wave * PX;

    float m_run[QT], l_run[QT];
    f32x4 acc[ET][QT];
#pragma unroll
    for (int j = 0; j < QT; ++j) {
        m_run[j] = -INFINITY;
        l_run[j] = 0.f;
#pragma unroll
        for (int t = 0; t < ET; ++t) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int st = 0; st < steps_per_wave; ++st) {
        const int n0 = n_wave0 + st * PX;
        if (n0 >= N) break;
        // ---- y^T tile: D[n][q] = sum_e x[e][n] * K[q][e]   (M = pixels, N = queries, K = features)
        f32x4 d[NT][QT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < QT; ++j) d[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += 4) {
            float bq[QT];
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                const int q = j * 16 + c;
                bq[j] = q < Q ? Kb[q * E + e0 + g] : 0.f;      // B[k=e][col=q]
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int n = n0 + i * 16 + c;
                const float a = ldb32(x_r, n < N ? ((unsigned)(e0 + g) * xse + (unsigned)n * xsn) * 4u : SQL_OOB);   // A[row=n][k=e]
#pragma unroll
                for (int j = 0; j < QT; ++j) d[i][j] = mfma16(a, bq[j], d[i][j]);
            }
        }
        // ---- store y (lane holds 4 consecutive pixels n0+i*16+4g.. of query j*16+c) and tile max
        float tmax[QT];
#pragma unroll
        for (int j = 0; j < QT; ++j) tmax[j] = -INFINITY;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n0 + i * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < QT; ++j) {
                const int q = j * 16 + c;
                f32x4 v = d[i][j];
                if (q < Q) {
                    if (n + 3 < N && (N & 3) == 0) {
                        *reinterpret_cast<f32x4 *>(yb + (size_t)q * N + n) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < N) yb[(size_t)q * N + n + r] = v[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= N) v[r] = -INFINITY;          // tail pixels take no softmax mass
                    tmax[j] = fmaxf(tmax[j], v[r]);
                }
                d[i][j] = v;
            }
        }
        // ---- online softmax update, p = exp(y - m) left in the accumulator registers
#pragma unroll
        for (int j = 0; j < QT; ++j) {
            const float mt = group_max(tmax[j]);
            const float mn = fmaxf(m_run[j], mt);
            const float sc = __expf(m_run[j] - mn);            // exp(-inf) = 0 on the first tile
            float ls = 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(d[i][j][r] - mn);
                    d[i][j][r] = p;
                    ls += p;
                }
            }
            l_run[j] = l_run[j] * sc + group_sum(ls);
            m_run[j] = mn;
#pragma unroll
            for (int t = 0; t < ET; ++t) acc[t][j] *= sc;
        }
        // ---- summary^T[e][q] += sum_n x[e][n] * p[n][q]: the k-slot (g, r) of pixel tile i is pixel n0+i*16+4g+r
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n0 + i * 16 + 4 * g;
#pragma unroll
            for (int t = 0; t < ET; ++t) {
                const unsigned xo = ((unsigned)(t * 16 + c) * xse + (unsigned)n * xsn) * 4u;   // A[row=e][k]: 4 consecutive pixels
                float a4[4];
                if ((N & 3) == 0 && xsn == 1) {                             // planar: a float4 lies inside a plane or beyond the last pixel
                    const sql_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(x_r, n < N ? xo : SQL_OOB, 0, 0);
                    a4[0] = __int_as_float(v.x); a4[1] = __int_as_float(v.y); a4[2] = __int_as_float(v.z); a4[3] = __int_as_float(v.w);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a4[r] = ldb32(x_r, n + r < N ? xo + 4u * (unsigned)(r * xsn) : SQL_OOB);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < QT; ++j) acc[t][j] = mfma16(a4[r], d[i][j][r], acc[t][j]);
            }
        }
    }

    // ---- merge the four waves of the workgroup through LDS, one partial record per workgroup
    constexpr int QP = QT * 16;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sm = lds;                                   // [4][QP][2]
    float *sacc = lds + 4 * QP * 2;                    // [4][E][QP+1]
    if (g == 0) {
#pragma unroll
        for (int j = 0; j < QT; ++j) {
            sm[(wave * QP + j * 16 + c) * 2] = m_run[j];
            sm[(wave * QP + j * 16 + c) * 2 + 1] = l_run[j];
        }
    }
#pragma unroll
    for (int t = 0; t < ET; ++t)
#pragma unroll
        for (int j = 0; j < QT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)                 // D: row e = 4g+r, col q = c
                sacc[((size_t)wave * E + t * 16 + 4 * g + r) * (QP + 1) + j * 16 + c] = acc[t][j][r];
    __syncthreads();
    float *po = part + ((size_t)b * nchunks + chunk) * Q * (E + PART_STRIDE_EXTRA);
    for (int q = threadIdx.x; q < Q; q += 256) {
        float mk[4], M = -INFINITY, L = 0.f, w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mk[k] = sm[(k * QP + q) * 2];
            M = fmaxf(M, mk[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[k] = mk[k] == -INFINITY ? 0.f : __expf(mk[k] - M);
            L += sm[(k * QP + q) * 2 + 1] * w[k];
        }
        float *o = po + (size_t)q * (E + PART_STRIDE_EXTRA);
        o[0] = M;
        o[1] = L;
        for (int e = 0; e < E; ++e) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += sacc[((size_t)k * E + e) * (QP + 1) + q] * w[k];
            o[2 + e] = s;
        }
    }
}

// merge the per-workgroup partials: summary[b,q,e], lse[b,q,2] = (max, 1/sum).  One wavefront per (b, q): the lanes take the
// chunks in parallel for the maximum and the sum (fixed-order butterflies), leave the chunk weights in LDS, and then take the features
// — every load of a pass is independent of the others (the first version chained three serial loops over the chunks: 32-44 us).
__global__ __launch_bounds__(64) void sql_merge_kernel(const float *__restrict__ part, float *__restrict__ summary,
                                                       float *__restrict__ lse, int Q, int E, int nchunks) {
    extern __shared__ float wl[];                        // [nchunks] weight exp(max_k - max) of every chunk
    const int b = blockIdx.y, q = blockIdx.x, lane = threadIdx.x;
    const size_t rec = (size_t)(E + PART_STRIDE_EXTRA), cs = (size_t)Q * rec;
    const float *p0 = part + ((size_t)b * nchunks) * cs + (size_t)q * rec;
    float M = -INFINITY;
    for (int k = lane; k < nchunks; k += 64) M = fmaxf(M, p0[(size_t)k * cs]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
    float L = 0.f;
    for (int k = lane; k < nchunks; k += 64) {
        const float mk = p0[(size_t)k * cs];
        const float w = mk == -INFINITY ? 0.f : __expf(mk - M);
        wl[k] = w;
        L += p0[(size_t)k * cs + 1] * w;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) L += __shfl_xor(L, o, 64);
    __syncthreads();
    const float iL = 1.f / L;
    for (int e = lane; e < E; e += 64) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < nchunks; k += 4) {
            s0 += p0[(size_t)k * cs + 2 + e] * wl[k];
            s1 += p0[(size_t)(k + 1) * cs + 2 + e] * wl[k + 1];
            s2 += p0[(size_t)(k + 2) * cs + 2 + e] * wl[k + 2];
            s3 += p0[(size_t)(k + 3) * cs + 2 + e] * wl[k + 3];
        }
        for (; k < nchunks; ++k) s0 += p0[(size_t)k * cs + 2 + e] * wl[k];
        summary[((size_t)b * Q + q) * E + e] = ((s0 + s1) + (s2 + s3)) * iL;
    }
    if (lane == 0) {
        lse[((size_t)b * Q + q) * 2] = M;
        lse[((size_t)b * Q + q) * 2 + 1] = iL;
    }
}


// ---------------------------------------------------------------------------------------------------
// backward, second formulation (the one dispatched): v_mfma_f32_32x32x2_f32 with lane = pixel, so every read of x, y,
// g_y and every write of g_x is a 128-byte row per half-wave (the 16x16 kernel above moves 64-byte pieces and needs
// 236-256 VGPRs, i.e. one wave per SIMD: 341 us at config B).  Same mathematics, same g_K partial layout.
//   t    = gS . x            A = gS [q][e] (LDS), B = x read as planes (lane = pixel)
//   gyt, s element-wise in the accumulator layout (row q = 32 qt + acc_row(r, h), column = pixel)
//   g_x  = K^T . gyt + gS^T . s        B operands are the accumulator registers themselves
//   g_K += gyt . x^T         gyt through a wave-private LDS tile [q][pixel], x re-read as float4 along the pixels
// ---------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
constexpr int TP = 36;                                   // floats per row of the wave-private [q][32 pixels] tile
// uniform base + 32-bit byte offset (saddr + voffset addressing: no 64-bit address registers per access)
__device__ __forceinline__ float ldg32(const float *__restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void stg32(float *__restrict__ base, unsigned byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// EH = ceil(E / 32) blocks of 32 features (E <= 64): t sums over both blocks, g_x and g_K are produced one block at a time
template <int QT, int EH>
__global__ __launch_bounds__(256) void sql_bwd32_kernel(const float *__restrict__ x, const float *__restrict__ K,
                                                        const float *__restrict__ y, const float *__restrict__ g_y,
                                                        const float *__restrict__ gS, const float *__restrict__ summary,
                                                        const float *__restrict__ lse, float *__restrict__ g_x,
                                                        float *__restrict__ gK_part, int Q, int E, int N, int nchunks, int xse,
                                                        int xsn, unsigned *__restrict__ amax_gx) {
    // amax_gx (may be NULL; cleared by the caller): the bit pattern of max |g_x| — g_x is the output gradient of the convolution that produced the
    // features, which reads it on two-term fp16 operands (sqd.h section 10b)
    unsigned am = 0u;
    constexpr int QP = QT * 32, EP = EH * 32 + 1;                 // x, g_x: element (e, n) at e * xse + n * xsn (see sql_fwd_kernel)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Kl = lds, *Sl = Kl + QP * EP, *qc = Sl + QP * EP;      // K[q][e], gS[q][e], per-query (max, 1/sum, dot, -)
    float *tiles = qc + QP * 4;                                   // [4 waves][QP][TP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float *xb = x + (size_t)b * E * N;
    const float *yb = y + (size_t)b * Q * N;
    const float *gyb = g_y ? g_y + (size_t)b * Q * N : nullptr;
    float *gxb = g_x + (size_t)b * E * N;
    const __amdgpu_buffer_rsrc_t x_r = sql_rsrc(xb, (unsigned)(E * N) * 4u), y_r = sql_rsrc(yb, (unsigned)(Q * N) * 4u);
    const __amdgpu_buffer_rsrc_t gy_r = sql_rsrc(gyb ? gyb : yb, gyb ? (unsigned)(Q * N) * 4u : 0u);      // no g_y: every read is 0
    const __amdgpu_buffer_rsrc_t gx_r = sql_rsrc(gxb, (unsigned)(E * N) * 4u);
    for (int idx = threadIdx.x; idx < QP * EP; idx += 256) {
        const int q = idx / EP, e = idx - q * EP;
        const bool ok = q < Q && e < E;
        Kl[idx] = ok ? K[((size_t)b * Q + q) * E + e] : 0.f;
        Sl[idx] = ok ? gS[((size_t)b * Q + q) * E + e] : 0.f;
    }
    for (int q = threadIdx.x; q < QP; q += 256) {
        float dsum = 0.f, mx = 0.f, il = 0.f;
        if (q < Q) {
            mx = lse[((size_t)b * Q + q) * 2];
            il = lse[((size_t)b * Q + q) * 2 + 1];
            for (int e = 0; e < E; ++e) dsum += gS[((size_t)b * Q + q) * E + e] * summary[((size_t)b * Q + q) * E + e];
        }
        qc[q * 4] = mx; qc[q * 4 + 1] = il; qc[q * 4 + 2] = dsum;
    }
    __syncthreads();
    float *tl = tiles + wave * QP * TP;
    f32x16 accK[EH][QT];                                          // g_K tile: row q, column e = 32 eh + (lane & 31)
#pragma unroll
    for (int eh = 0; eh < EH; ++eh)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accK[eh][qt][r] = 0.f;
    const int ntiles = (N + 31) / 32;
    const bool vec_ok = (N & 3) == 0;

    for (int tile = chunk * 4 + wave; tile < ntiles; tile += nchunks * 4) {
        const int p0 = tile * 32, p = p0 + i;
        const bool pv = p < N;
        // ---- t[q][p] = sum_e gS[q][e] x[e][p]
        const unsigned N4 = (unsigned)N * 4u;
        const unsigned lane_x = pv ? ((unsigned)h * xse + (unsigned)p * xsn) * 4u : SQL_OOB;   // feature h, pixel p
        const unsigned lane_q = pv ? ((unsigned)(4 * h) * N + p) * 4u : SQL_OOB;     // row 4h of a 32-row group of y / g_y, pixel p
        const unsigned lane_gx = pv ? ((unsigned)(4 * h) * xse + (unsigned)p * xsn) * 4u : SQL_OOB;
        const unsigned xse4 = (unsigned)xse * 4u;
        float xe[EH][16];
#pragma unroll
        for (int eh = 0; eh < EH; ++eh) {
            if (xse == 1) {                                  // pixel-major: 32 features of the lane are one 128-byte run — 8 x 16-byte loads
                const unsigned px_off = pv ? (unsigned)p * (unsigned)xsn * 4u : SQL_OOB;
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                    const sql_i32x4 f = __builtin_amdgcn_raw_buffer_load_b128(x_r, eh * 32 + 4 * q8 < E ? px_off + 128u * eh + 16u * q8 : SQL_OOB, 0, 0);
                    xe[eh][2 * q8] = __int_as_float(h ? f.y : f.x);          // feature 32 eh + 4 q8 + h
                    xe[eh][2 * q8 + 1] = __int_as_float(h ? f.w : f.z);      // feature 32 eh + 4 q8 + 2 + h
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    xe[eh][s] = ldb32(x_r, eh * 32 + 2 * s + h < E ? lane_x + (unsigned)(eh * 32 + 2 * s) * xse4 : SQL_OOB);
            }
        }
        // every global read of the tile is issued up front (y, g_y, the float4 pieces of x for the g_K product): the first
        // product then runs under their latency instead of each phase waiting for its own loads
        f32x16 yv[QT], gv[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned o = lane_q + (unsigned)(qt * 32 + (r & 3) + 8 * (r >> 2)) * N4;
                yv[qt][r] = ldb32(y_r, o);
                gv[qt][r] = ldb32(gy_r, o);
            }
        float xv[EH][4][4];
#pragma unroll
        for (int eh = 0; eh < EH; ++eh)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int px = 8 * gq + 4 * h, ei = eh * 32 + i;
                if (vec_ok && xsn == 1) {                            // planar, N % 4 == 0: a float4 is inside a plane or beyond it
                    const sql_i32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(
                        x_r, (ei < E && p0 + px < N) ? ((unsigned)ei * xse + p0 + px) * 4u : SQL_OOB, 0, 0);
                    xv[eh][gq][0] = __int_as_float(t4.x); xv[eh][gq][1] = __int_as_float(t4.y);
                    xv[eh][gq][2] = __int_as_float(t4.z); xv[eh][gq][3] = __int_as_float(t4.w);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xv[eh][gq][j] = ldb32(x_r, (ei < E && p0 + px + j < N) ? ((unsigned)ei * xse + (unsigned)(p0 + px + j) * xsn) * 4u : SQL_OOB);
                }
            }
        f32x16 acc[QT], sreg[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qt][r] = 0.f;
#pragma unroll
        for (int eh = 0; eh < EH; ++eh)
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc[qt] = mfma32(Sl[(qt * 32 + i) * EP + eh * 32 + 2 * s + h], xe[eh][s], acc[qt]);
        // ---- s and gyt, element-wise; gyt also to the LDS tile (operand of the g_K product)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + acc_row(r, h);
                float sv = 0.f, gyt = 0.f;
                if (q < Q && pv) {
                    sv = __expf(yv[qt][r] - qc[q * 4]) * qc[q * 4 + 1];
                    gyt = gv[qt][r] + sv * (acc[qt][r] - qc[q * 4 + 2]);
                }
                sreg[qt][r] = sv;
                acc[qt][r] = gyt;
                tl[q * TP + i] = gyt;
            }
        // ---- g_x[e][p] = sum_q K[q][e] gyt[q][p] + gS[q][e] s[q][p]
#pragma unroll
        for (int eh = 0; eh < EH; ++eh) {
            f32x16 gx;
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[r] = 0.f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 32 + acc_row(r, h);
                    gx = mfma32(Kl[q * EP + eh * 32 + i], acc[qt][r], gx);
                    gx = mfma32(Sl[q * EP + eh * 32 + i], sreg[qt][r], gx);
                }
            if (xse == 1) {                                  // pixel-major: registers 4g..4g+3 are 4 consecutive features -> one 16-byte store
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int e0 = eh * 32 + 8 * g4 + 4 * h;
                    sql_i32x4 v;
                    v.x = __float_as_int(gx[4 * g4]); v.y = __float_as_int(gx[4 * g4 + 1]);
                    v.z = __float_as_int(gx[4 * g4 + 2]); v.w = __float_as_int(gx[4 * g4 + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, gx_r, (pv && e0 < E) ? ((unsigned)p * (unsigned)xsn + e0) * 4u : SQL_OOB, 0, 0);
                    if (pv && e0 < E)
                        am = max(max(am, abs_bits(gx[4 * g4])), max(abs_bits(gx[4 * g4 + 1]), max(abs_bits(gx[4 * g4 + 2]), abs_bits(gx[4 * g4 + 3]))));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    stb32(gx_r, eh * 32 + acc_row(r, h) < E ? lane_gx + (unsigned)(eh * 32 + (r & 3) + 8 * (r >> 2)) * xse4 : SQL_OOB,
                          gx[r]);                                                                           // pixel >= N: dropped
                    if (pv && eh * 32 + acc_row(r, h) < E) am = max(am, abs_bits(gx[r]));
                }
            }
        }
        // ---- g_K[q][e] += sum_p gyt[q][p] x[e][p]; k-step (gq, j): half-wave 0 takes pixel 8gq+j, half-wave 1 pixel 8gq+4+j
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int px = 8 * gq + 4 * h;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float4 a4 = *reinterpret_cast<const float4 *>(tl + (qt * 32 + i) * TP + px);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int eh = 0; eh < EH; ++eh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accK[eh][qt] = mfma32(av[j], xv[eh][gq][j], accK[eh][qt]);
            }
        }
        __builtin_amdgcn_wave_barrier();                           // the tile is rewritten by the next iteration
    }
    // ---- workgroup merge of g_K (fixed order) and the partial of this chunk
    float *red = tiles;                                            // [4][QP][32]
    float *po = gK_part + ((size_t)b * nchunks + chunk) * Q * E;
#pragma unroll
    for (int eh = 0; eh < EH; ++eh) {
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((size_t)wave * QP + qt * 32 + acc_row(r, h)) * 32 + i] = accK[eh][qt][r];
        __syncthreads();
        const int ew = min(32, E - eh * 32);                       // features of this block
        for (int idx = threadIdx.x; idx < Q * ew; idx += 256) {
            const int q = idx / ew, e = idx - q * ew;
            po[(size_t)q * E + eh * 32 + e] = ((red[((size_t)0 * QP + q) * 32 + e] + red[((size_t)1 * QP + q) * 32 + e]) +
                                               red[((size_t)2 * QP + q) * 32 + e]) + red[((size_t)3 * QP + q) * 32 + e];
        }
    }
    amax_commit(am, amax_gx);
}

// ---------------------------------------------------------------------------------------------------
// backward, third formulation (dispatched for Q > 32): the SAME products, but a wave owns ONE group of 32 queries.  With all QT
// groups in one wave the kernel above needs 239 - 256 registers (+ 396 bytes of scratch at Q = 120 / E = 64), i.e. one wave per
// SIMD, and the load / product / exp / LDS phases of a tile run one after the other.  Here the four waves of a workgroup are QT
// query-group waves x 4 / QT pixel tiles; g_x — the only product that contracts over the queries — is formed per group and the QT
// partial tiles are added through LDS (fixed order), each wave storing its share of the feature rows; g_K needs no exchange until
// the final merge.  ~130 - 180 registers: two to three waves per SIMD.
// ---------------------------------------------------------------------------------------------------
template <int QT, int EH>
__global__ __launch_bounds__(256) void sql_bwd32q_kernel(const float *__restrict__ x, const float *__restrict__ K,
                                                         const float *__restrict__ y, const float *__restrict__ g_y,
                                                         const float *__restrict__ gS, const float *__restrict__ summary,
                                                         const float *__restrict__ lse, float *__restrict__ g_x,
                                                         float *__restrict__ gK_part, int Q, int E, int N, int nchunks, int xse,
                                                         int xsn, unsigned *__restrict__ amax_gx) {
    static_assert(QT == 2 || QT == 4, "query-group waves per pixel tile");
    unsigned am = 0u;                                             // (max |g_x| over this thread's stores: sql_bwd32_kernel)
    constexpr int QP = QT * 32, EP = EH * 32 + 1, PTW = 4 / QT;   // pixel tiles per workgroup iteration
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Kl = lds, *Sl = Kl + QP * EP, *qc = Sl + QP * EP;      // K[q][e], gS[q][e], per-query (max, 1/sum, dot, -)
    float *tiles = qc + QP * 4;                                   // [4 waves][32][TP]: gyt of the wave's query group
    // [4 waves][EH][16][64]: partial g_x tiles — with one feature block they reuse the wave's gyt tile (dead once the g_K product is
    // issued), which keeps the workgroup at 37 KB of LDS: three workgroups per CU (the register file's limit) instead of two
    constexpr int GXW = EH == 1 ? 32 * TP : EH * 16 * 64;         // floats per wave
    float *gxs = EH == 1 ? tiles : tiles + 4 * 32 * TP;
    // round 5: the x tile(s) of an iteration — 32 pixels x E features each, needed by all QT query-group waves of a pixel tile in two operand
    // layouts — are fetched ONCE per workgroup (one 16-byte load per thread and 1024 floats, fully coalesced) into [PTW][32][XP] and read from
    // there; before, every wave fetched them itself: 8 loads of 32 cache lines + 16 loads per wave and tile, 4x over (pixel-major x only)
    constexpr int XP = EH * 32 + 4;                               // row pitch: 16-byte reads of 16 consecutive rows are conflict-free
    float *xs = gxs + (EH == 1 ? 0 : 4 * GXW) + (EH == 1 ? 4 * 32 * TP : 0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wq = wave % QT, wp = wave / QT;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float *xb = x + (size_t)b * E * N;
    const float *yb = y + (size_t)b * Q * N;
    const float *gyb = g_y ? g_y + (size_t)b * Q * N : nullptr;
    float *gxb = g_x + (size_t)b * E * N;
    const __amdgpu_buffer_rsrc_t x_r = sql_rsrc(xb, (unsigned)(E * N) * 4u), y_r = sql_rsrc(yb, (unsigned)(Q * N) * 4u);
    const __amdgpu_buffer_rsrc_t gy_r = sql_rsrc(gyb ? gyb : yb, gyb ? (unsigned)(Q * N) * 4u : 0u);      // no g_y: every read is 0
    const __amdgpu_buffer_rsrc_t gx_r = sql_rsrc(gxb, (unsigned)(E * N) * 4u);
    for (int idx = threadIdx.x; idx < QP * EP; idx += 256) {
        const int q = idx / EP, e = idx - q * EP;
        const bool ok = q < Q && e < E;
        Kl[idx] = ok ? K[((size_t)b * Q + q) * E + e] : 0.f;
        Sl[idx] = ok ? gS[((size_t)b * Q + q) * E + e] : 0.f;
    }
    for (int q = threadIdx.x; q < QP; q += 256) {
        float dsum = 0.f, mx = 0.f, il = 0.f;
        if (q < Q) {
            mx = lse[((size_t)b * Q + q) * 2];
            il = lse[((size_t)b * Q + q) * 2 + 1];
            for (int e = 0; e < E; ++e) dsum += gS[((size_t)b * Q + q) * E + e] * summary[((size_t)b * Q + q) * E + e];
        }
        qc[q * 4] = mx; qc[q * 4 + 1] = il; qc[q * 4 + 2] = dsum;
    }
    __syncthreads();
    float *tl = tiles + wave * 32 * TP;
    const float *Kq = Kl + wq * 32 * EP, *Sq = Sl + wq * 32 * EP, *qcq = qc + wq * 32 * 4;
    f32x16 accK[EH];                                              // g_K tile of the wave's group: row q, column e = 32 eh + (lane & 31)
#pragma unroll
    for (int eh = 0; eh < EH; ++eh)
#pragma unroll
        for (int r = 0; r < 16; ++r) accK[eh][r] = 0.f;
    const int ntiles = (N + 31) / 32;
    const bool vec_ok = (N & 3) == 0;
    const int iters = (ntiles + nchunks * PTW - 1) / (nchunks * PTW);

    for (int it = 0; it < iters; ++it) {                          // (uniform trip count: the loop holds workgroup barriers)
        const int tile = (it * nchunks + chunk) * PTW + wp;
        const int p0 = tile * 32, p = p0 + i;
        const bool pv = p < N;                                    // (a tile beyond the image: every access masked)
        const unsigned N4 = (unsigned)N * 4u;
        const unsigned lane_x = pv ? ((unsigned)h * xse + (unsigned)p * xsn) * 4u : SQL_OOB;
        const unsigned lane_q = pv ? ((unsigned)(wq * 32 + 4 * h) * N + p) * 4u : SQL_OOB;   // row 4h of the group's rows of y / g_y, pixel p
        const unsigned lane_gx = pv ? ((unsigned)(4 * h) * xse + (unsigned)p * xsn) * 4u : SQL_OOB;
        const unsigned xse4 = (unsigned)xse * 4u;
        const bool shared_x = xse == 1;                          // (wave-uniform)
        if (shared_x) {
            constexpr int F4 = EH * 8;                           // 16-byte pieces per pixel row
#pragma unroll
            for (int v = 0; v < PTW * 32 * F4 / 256; ++v) {
                const int idx = threadIdx.x + 256 * v, pt = idx / (32 * F4), rem = idx - pt * (32 * F4), px = rem / F4, f4 = rem - px * F4;
                const int pg = ((it * nchunks + chunk) * PTW + pt) * 32 + px;
                const sql_i32x4 f = __builtin_amdgcn_raw_buffer_load_b128(x_r, (pg < N && 4 * f4 < E) ? ((unsigned)pg * (unsigned)xsn + 4u * f4) * 4u : SQL_OOB, 0, 0);
                *reinterpret_cast<sql_i32x4 *>(xs + (pt * 32 + px) * XP + 4 * f4) = f;
            }
            __syncthreads();
        }
        const float *xt = xs + wp * 32 * XP;                      // this wave's pixel tile
        float xe[EH][16];
#pragma unroll
        for (int eh = 0; eh < EH; ++eh) {
            if (shared_x) {                                  // the lane's pixel row: 8 x 16-byte LDS reads per 32 features
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                    const float4 f = *reinterpret_cast<const float4 *>(xt + i * XP + eh * 32 + 4 * q8);
                    xe[eh][2 * q8] = h ? f.y : f.x;          // feature 32 eh + 4 q8 + h
                    xe[eh][2 * q8 + 1] = h ? f.w : f.z;      // feature 32 eh + 4 q8 + 2 + h
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    xe[eh][s] = ldb32(x_r, eh * 32 + 2 * s + h < E ? lane_x + (unsigned)(eh * 32 + 2 * s) * xse4 : SQL_OOB);
            }
        }
        f32x16 yv, gv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool qv = wq * 32 + acc_row(r, h) < Q;
            const unsigned o = qv ? lane_q + (unsigned)((r & 3) + 8 * (r >> 2)) * N4 : SQL_OOB;
            yv[r] = ldb32(y_r, o);
            gv[r] = ldb32(gy_r, o);
        }
        float xv[EH][4][4];
#pragma unroll
        for (int eh = 0; eh < EH; ++eh)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int px = 8 * gq + 4 * h, ei = eh * 32 + i;
                if (shared_x) {                                      // lane = feature: four pixels of the shared tile
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[eh][gq][j] = xt[(px + j) * XP + ei];
                } else if (vec_ok && xsn == 1) {                     // planar, N % 4 == 0: a float4 is inside a plane or beyond it
                    const sql_i32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(
                        x_r, (ei < E && p0 + px < N) ? ((unsigned)ei * xse + p0 + px) * 4u : SQL_OOB, 0, 0);
                    xv[eh][gq][0] = __int_as_float(t4.x); xv[eh][gq][1] = __int_as_float(t4.y);
                    xv[eh][gq][2] = __int_as_float(t4.z); xv[eh][gq][3] = __int_as_float(t4.w);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xv[eh][gq][j] = ldb32(x_r, (ei < E && p0 + px + j < N) ? ((unsigned)ei * xse + (unsigned)(p0 + px + j) * xsn) * 4u : SQL_OOB);
                }
            }
        // ---- t[q][p] = sum_e gS[q][e] x[e][p]
        f32x16 acc, sreg;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // (round 5: every LDS operand of a product is fetched BEFORE its matrix instructions, behind a scheduling fence — left to itself the
        //  compiler emitted read - wait - multiply per instruction, and the two waves of a SIMD spent their time at those waits: the matrix
        //  pipe was busy 32 % of the launch)
        {
            float sa[EH][16];
#pragma unroll
            for (int eh = 0; eh < EH; ++eh)
#pragma unroll
                for (int s = 0; s < 16; ++s) sa[eh][s] = Sq[i * EP + eh * 32 + 2 * s + h];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int eh = 0; eh < EH; ++eh)
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = mfma32(sa[eh][s], xe[eh][s], acc);
        }
        // ---- s and gyt, element-wise; gyt also to the LDS tile (operand of the g_K product)
        {
            float4 qv4[16];                                           // (max, 1/sum, dot, -) of the lane's 16 query rows: 16-byte reads, issued together
#pragma unroll
            for (int r = 0; r < 16; ++r) qv4[r] = *reinterpret_cast<const float4 *>(qcq + acc_row(r, h) * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = acc_row(r, h);
                float sv = 0.f, gyt = 0.f;
                if (wq * 32 + ql < Q && pv) {
                    sv = __expf(yv[r] - qv4[r].x) * qv4[r].y;
                    gyt = gv[r] + sv * (acc[r] - qv4[r].z);
                }
                sreg[r] = sv;
                acc[r] = gyt;
                tl[ql * TP + i] = gyt;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- g_K[q][e] += sum_p gyt[q][p] x[e][p]; k-step (gq, j): half-wave 0 takes pixel 8gq+j, half-wave 1 pixel 8gq+4+j
        {
            float4 a4[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) a4[gq] = *reinterpret_cast<const float4 *>(tl + i * TP + 8 * gq + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float av[4] = {a4[gq].x, a4[gq].y, a4[gq].z, a4[gq].w};
#pragma unroll
                for (int eh = 0; eh < EH; ++eh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accK[eh] = mfma32(av[j], xv[eh][gq][j], accK[eh]);
            }
        }
        __builtin_amdgcn_wave_barrier();                           // (the tile's reads are issued before it is overwritten below)
        // ---- the group's part of g_x[e][p] = sum_q K[q][e] gyt[q][p] + gS[q][e] s[q][p]
#pragma unroll
        for (int eh = 0; eh < EH; ++eh) {
            f32x16 gx;
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[r] = 0.f;
            float ka[16], sb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ka[r] = Kq[acc_row(r, h) * EP + eh * 32 + i];
                sb[r] = Sq[acc_row(r, h) * EP + eh * 32 + i];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                gx = mfma32(ka[r], acc[r], gx);
                gx = mfma32(sb[r], sreg[r], gx);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) gxs[wave * GXW + (eh * 16 + r) * 64 + lane] = gx[r];
        }
        __syncthreads();
        // ---- add the QT groups (fixed order); wave wq stores register groups g4 = wq * 4 / QT .. of every feature block
#pragma unroll
        for (int eh = 0; eh < EH; ++eh)
#pragma unroll
            for (int gg = 0; gg < 4 / QT; ++gg) {
                const int g4 = wq * (4 / QT) + gg;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sum = gxs[(wp * QT) * GXW + (eh * 16 + 4 * g4 + j) * 64 + lane];
#pragma unroll
                    for (int k = 1; k < QT; ++k) sum += gxs[(wp * QT + k) * GXW + (eh * 16 + 4 * g4 + j) * 64 + lane];
                    v[j] = sum;
                }
                if (xse == 1) {                              // pixel-major: registers 4 g4 .. + 3 are 4 consecutive features -> one 16-byte store
                    const int e0 = eh * 32 + 8 * g4 + 4 * h;
                    sql_i32x4 o;
                    o.x = __float_as_int(v[0]); o.y = __float_as_int(v[1]); o.z = __float_as_int(v[2]); o.w = __float_as_int(v[3]);
                    __builtin_amdgcn_raw_buffer_store_b128(o, gx_r, (pv && e0 < E) ? ((unsigned)p * (unsigned)xsn + e0) * 4u : SQL_OOB, 0, 0);
                    if (pv && e0 < E) am = max(max(am, abs_bits(v[0])), max(abs_bits(v[1]), max(abs_bits(v[2]), abs_bits(v[3]))));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g4 + j;
                        stb32(gx_r, eh * 32 + acc_row(r, h) < E ? lane_gx + (unsigned)(eh * 32 + (r & 3) + 8 * (r >> 2)) * xse4 : SQL_OOB, v[j]);
                        if (pv && eh * 32 + acc_row(r, h) < E) am = max(am, abs_bits(v[j]));
                    }
                }
            }
        __syncthreads();                                           // gxs and the gyt tiles are rewritten by the next iteration
    }
    // ---- workgroup merge of g_K (fixed order over the pixel-tile waves of a group) and the partial of this chunk
    float *red = tiles;                                            // [4][32][32] (TP = 36 >= 32)
    float *po = gK_part + ((size_t)b * nchunks + chunk) * Q * E;
#pragma unroll
    for (int eh = 0; eh < EH; ++eh) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((size_t)wave * 32 + acc_row(r, h)) * 32 + i] = accK[eh][r];
        __syncthreads();
        const int ew = min(32, E - eh * 32);                       // features of this block
        for (int idx = threadIdx.x; idx < Q * ew; idx += 256) {
            const int q = idx / ew, e = idx - q * ew, g = q >> 5, ql = q & 31;
            float sum = red[((size_t)g * 32 + ql) * 32 + e];
#pragma unroll
            for (int k = 1; k < PTW; ++k) sum += red[((size_t)(k * QT + g) * 32 + ql) * 32 + e];
            po[(size_t)q * E + eh * 32 + e] = sum;
        }
    }
    amax_commit(am, amax_gx);
}

// g_K[b,q,e] = sum over chunks of gK_part: 64 outputs per workgroup, the chunks dealt to its four waves (fixed order), added through LDS
__global__ __launch_bounds__(256) void sql_gk_reduce_kernel(const float *__restrict__ part, float *__restrict__ gK, int QE,
                                                            int nchunks) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, grp = threadIdx.x >> 6, idx = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (idx < QE) {
        int k = grp;
        for (; k + 4 < nchunks; k += 8) {
            s0 += part[((size_t)b * nchunks + k) * QE + idx];
            s1 += part[((size_t)b * nchunks + k + 4) * QE + idx];
        }
        if (k < nchunks) s0 += part[((size_t)b * nchunks + k) * QE + idx];
    }
    red[grp][lane] = s0 + s1;
    __syncthreads();
    if (grp == 0 && idx < QE) gK[(size_t)b * QE + idx] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// ---------------------------------------------------------------------------------------------------
// forward, second formulation (the one dispatched for pixel-major x with 32 or 64 features): v_mfma_f32_32x32x2_f32 in the
// orientation of the backward — D[q][pixel], lane = pixel — so that every store of y is a 128-byte row per half-wave and x is read
// as 64-byte runs per lane / 128-byte rows per half-wave (the 16x16 kernel above fetches x and K as 16-byte pieces of 16 lines per
// instruction and re-reads K from memory every step: 111 us at config B against 28 us of y traffic).
//   y^T tile   D[q][n]  = sum_e K[q][e] x[n][e]        A = K from LDS, B = the lane's own 16 features (k-slot (s, h) = feature 16 h + s)
//   p          = exp(D - ref[q])                        ref[q]: the row maximum of the wave's FIRST tile; softmax does not care which
//                                                       reference is used as long as nothing overflows, so later tiles only test
//                                                       D - ref <= 60 (one ballot) and take the rescale path when the test fails
//   summary    D2[q][e] += sum_n p[q][n] x[n][e]        A = p through a wave-private LDS tile [q][pixel], B = x rows (lane = feature)
// Per-lane partial row sums of p (lane = pixel class mod 32) are added across lanes once, at the end; the four waves of a workgroup
// are merged through LDS into one (max, sum, unnormalised summary) record of the layout sql_merge_kernel reads.
// ---------------------------------------------------------------------------------------------------
constexpr int PT = 33;                                   // floats per row of the wave-private [q][32 pixels] probability tile
__device__ __forceinline__ float half_max(float v) {     // over the 32 lanes of a half-wave
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64));
    v = fmaxf(v, __shfl_xor(v, 4, 64));
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    return fmaxf(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ float half_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v + __shfl_xor(v, 16, 64);
}

template <int QT, int EH>
__global__ __launch_bounds__(256) void sql_fwd32_kernel(const float *__restrict__ x, const float *__restrict__ K, float *__restrict__ y,
                                                        float *__restrict__ part, int Qall, int N, int tiles_per_wave, int nchunks) {
    // blockIdx.z selects a group of 32 QT queries (Q of them valid, Qall in total): E = 64 with more than 64 queries runs as two groups
    constexpr int E = 32 * EH, QP = 32 * QT, EP = E + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Kl = lds;                                     // [QP][EP]
    float *refl = Kl + QP * EP;                          // [4 waves][QP]
    float *ptile = refl + 4 * QP;                        // [4 waves][32][PT]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, chunk = blockIdx.x, q0 = blockIdx.z * QP, Q = min(QP, Qall - q0);
    const float *xb = x + (size_t)b * E * N;
    float *yb = y + ((size_t)b * Qall + q0) * N;
    const __amdgpu_buffer_rsrc_t x_r = sql_rsrc(xb, (unsigned)(E * N) * 4u);
    const __amdgpu_buffer_rsrc_t y_r = sql_rsrc(yb, (unsigned)(Q * N) * 4u);
    for (int idx = threadIdx.x; idx < QP * EP; idx += 256) {
        const int q = idx / EP, e = idx - q * EP;
        Kl[idx] = (q < Q && e < E) ? K[((size_t)b * Qall + q0 + q) * E + e] : 0.f;
    }
    __syncthreads();
    float *rl = refl + wave * QP, *pt = ptile + wave * 32 * PT;

    f32x16 acc2[QT][EH];
    float lsum[QT][16];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
        for (int eh = 0; eh < EH; ++eh)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[qt][eh][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) lsum[qt][r] = 0.f;
    }
    const int tile0 = (chunk * 4 + wave) * tiles_per_wave, ntiles = (N + 31) / 32;
    bool first = true;
    for (int tl = tile0; tl < min(ntiles, tile0 + tiles_per_wave); ++tl) {
        const int n0 = tl * 32, n = n0 + i;
        const bool pv = n < N;
        // ---- x in the two operand layouts: the lane's pixel (16 features per half, 4 x 16-byte loads per 32 features) and, for the
        // summary product, rows of pixels (lane = feature, k-slot (s, h) = pixel 2 s + h)
        float xr[EH][16], xc[EH][16];
        const unsigned po = pv ? ((unsigned)n * E + 16u * h) * 4u : SQL_OOB;
#pragma unroll
        for (int eh = 0; eh < EH; ++eh)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const sql_i32x4 f = __builtin_amdgcn_raw_buffer_load_b128(x_r, pv ? po + 128u * eh + 16u * v : SQL_OOB, 0, 0);
                xr[eh][4 * v] = __int_as_float(f.x); xr[eh][4 * v + 1] = __int_as_float(f.y);
                xr[eh][4 * v + 2] = __int_as_float(f.z); xr[eh][4 * v + 3] = __int_as_float(f.w);
            }
#pragma unroll
        for (int eh = 0; eh < EH; ++eh)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int pn = n0 + 2 * s + h;
                xc[eh][s] = ldb32(x_r, pn < N ? ((unsigned)pn * E + 32u * eh + i) * 4u : SQL_OOB);
            }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            {                                                 // (operands first, then the products: see sql_bwd32q_kernel)
                float ka[EH][16];
#pragma unroll
                for (int eh = 0; eh < EH; ++eh)
#pragma unroll
                    for (int s = 0; s < 16; ++s) ka[eh][s] = Kl[(qt * 32 + i) * EP + eh * 32 + 16 * h + s];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int eh = 0; eh < EH; ++eh)
#pragma unroll
                    for (int s = 0; s < 16; ++s) d = mfma32(ka[eh][s], xr[eh][s], d);
            }
            // ---- y out: row q = qt * 32 + acc_row(r, h), 32 consecutive pixels per half-wave
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + acc_row(r, h);
                stb32(y_r, (pv && q < Q) ? ((unsigned)q * N + n) * 4u : SQL_OOB, d[r]);
                if (!pv) d[r] = -INFINITY;                   // tail pixels take no softmax mass
            }
            if (first) {                                      // the reference of this wave: row maxima of its first tile
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m = half_max(d[r]);
                    if (i == 0) rl[qt * 32 + acc_row(r, h)] = m;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            float ref[16];
            bool over = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ref[r] = rl[qt * 32 + acc_row(r, h)];
                over |= d[r] - ref[r] > 60.f;
            }
            if (__any(over)) {                                // rare: move the reference up and rescale what was accumulated under the old one
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m = fmaxf(ref[r], half_max(d[r]));
                    const float sc = __expf(ref[r] - m);
                    lsum[qt][r] *= sc;
#pragma unroll
                    for (int eh = 0; eh < EH; ++eh) acc2[qt][eh][r] *= sc;
                    ref[r] = m;
                    if (i == 0) rl[qt * 32 + acc_row(r, h)] = m;
                }
            }
            __builtin_amdgcn_wave_barrier();                  // the previous tile's reads of the probability tile are done
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __expf(d[r] - ref[r]);
                lsum[qt][r] += pr;
                pt[acc_row(r, h) * PT + i] = pr;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            {
                float pa[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) pa[s] = pt[i * PT + 2 * s + h];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int eh = 0; eh < EH; ++eh) acc2[qt][eh] = mfma32(pa[s], xc[eh][s], acc2[qt][eh]);
            }
        }
        first = false;
    }
    // ---- one record per workgroup: the four waves' (reference, sum, summary) through LDS
    __syncthreads();
    float *sm = lds;                                         // [4][QP][2]
    float *sacc = lds + 4 * QP * 2;                          // [4][QP][EP]
    float rv[QT][16];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[qt][r] = first ? -INFINITY : rl[qt * 32 + acc_row(r, h)];      // (a wave without tiles: no mass)
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = qt * 32 + acc_row(r, h);
            const float l = half_sum(lsum[qt][r]);
            if (i == 0) {
                sm[(wave * QP + q) * 2] = rv[qt][r];
                sm[(wave * QP + q) * 2 + 1] = l;
            }
#pragma unroll
            for (int eh = 0; eh < EH; ++eh) sacc[(wave * QP + q) * EP + eh * 32 + i] = acc2[qt][eh][r];
        }
    __syncthreads();
    float *po = part + (((size_t)b * nchunks + chunk) * Qall + q0) * (E + PART_STRIDE_EXTRA);
    for (int idx = threadIdx.x; idx < Q * 4; idx += 256) {          // 4 threads per query: a quarter of the features each
        const int q = idx >> 2, eq = idx & 3;
        float mk[4], M = -INFINITY, L = 0.f, w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mk[k] = sm[(k * QP + q) * 2];
            M = fmaxf(M, mk[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            w[k] = mk[k] == -INFINITY ? 0.f : __expf(mk[k] - M);
            L += sm[(k * QP + q) * 2 + 1] * w[k];
        }
        float *o = po + (size_t)q * (E + PART_STRIDE_EXTRA);
        if (eq == 0) {
            o[0] = M;
            o[1] = L;
        }
        for (int e = eq * (E / 4); e < (eq + 1) * (E / 4); ++e) {
            float sv = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sv += sacc[(k * QP + q) * EP + e] * w[k];
            o[2 + e] = sv;
        }
    }
}

struct Plan {
    int QT, ET, NT, steps, nchunks;
    int tiles32, nchunks32;          // sql_fwd32_kernel (E = 32 | 64, pixel-major x): 32-pixel tiles per wave, workgroups per image; 0: not built
    int nchunks_bwd;                 // workgroups per image of the backward: B * nchunks_bwd = one resident round (3 or 2 workgroups per CU)
};
int make_plan(int Q, int E, int N, Plan *p, int B = 12) {
    if (E % 16 != 0 || E > 64 || E < 16 || Q < 1 || Q > 128 || N < 1) return -1;
    int QT = (Q + 15) / 16;
    QT = QT <= 1 ? 1 : QT <= 2 ? 2 : QT <= 4 ? 4 : 8;
    p->QT = QT;
    p->ET = E / 16;
    p->NT = QT >= 8 ? 1 : QT >= 4 ? 2 : 4;
    const int px = p->NT * 16;
    // aim at >= ~2048 waves per launch-batch of 12 images: a wave walks `steps` pixel tiles
    int steps = 2;
    while (steps < 16 && N / (px * steps * 4) > 64) steps *= 2;
    p->steps = steps;
    p->nchunks = (N + px * steps * 4 - 1) / (px * steps * 4);
    p->tiles32 = p->nchunks32 = 0;
    {
        const int slots = E > 32 ? 512 : 768, ntiles = (N + 31) / 32;
        int nb = slots / (B < 1 ? 1 : B);
        nb = nb < 1 ? 1 : nb > ntiles ? ntiles : nb;
        p->nchunks_bwd = Q > 32 ? nb : p->nchunks;              // (Q <= 32: the one-group kernel keeps its plan)
    }
    if (E == 32 || E == 64) {                                   // 32 workgroups of 4 waves per image and query group
        const int ntiles = (N + 31) / 32;
        int tpw = (ntiles + 127) / 128;                          // (8 tiles per wave at config B: 60.5 -> 56 us against 4; the merge 8 -> 6 us)
        tpw = tpw < 1 ? 1 : tpw > 16 ? 16 : tpw;
        p->tiles32 = tpw;
        p->nchunks32 = (ntiles + 4 * tpw - 1) / (4 * tpw);
    }
    return 0;
}
}  // namespace

extern "C" int sqd_sql_workspace(int B, int Q, int E, int N, int64_t *part_floats, int64_t *gk_part_floats) {
    Plan p;
    SQD_CHECK_ARG(make_plan(Q, E, N, &p, B) == 0, "sqd_sql: unsupported Q=%d E=%d N=%d (E in {16, 32, 48, 64}, Q <= 128)", Q, E, N);
    if (part_floats) *part_floats = (int64_t)B * (p.nchunks > p.nchunks32 ? p.nchunks : p.nchunks32) * Q * (E + PART_STRIDE_EXTRA);
    if (gk_part_floats) *gk_part_floats = (int64_t)B * p.nchunks_bwd * Q * E;
    return SQD_OK;
}

#define SQL_DISPATCH(QT_, ET_, NT_, KERNEL, SHMEM, ...)                                                        \
    if (p.QT == QT_ && p.ET == ET_) {                                                                           \
        if ((SHMEM) > 48 * 1024)                                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<QT_, ET_, NT_>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SHMEM));                \
        hipLaunchKernelGGL((KERNEL<QT_, ET_, NT_>), dim3(p.nchunks, B), dim3(256), SHMEM, (hipStream_t)stream, __VA_ARGS__); \
        launched = true;                                                                                        \
    }

#define SQL_DISPATCH_ALL(KERNEL, SHMEM, ...)                 \
    SQL_DISPATCH(1, 1, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(2, 1, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(4, 1, 2, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(8, 1, 1, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(1, 2, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(2, 2, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(4, 2, 2, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(8, 2, 1, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(1, 3, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(2, 3, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(4, 3, 2, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(8, 3, 1, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(1, 4, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(2, 4, 4, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(4, 4, 2, KERNEL, SHMEM, __VA_ARGS__)        \
    SQL_DISPATCH(8, 4, 1, KERNEL, SHMEM, __VA_ARGS__)

extern "C" int sqd_sql_fwd(const float *x, const float *K, float *y, float *summary, float *lse, float *part, int B, int Q,
                           int E, int N, int x_nhwc, void *stream) {
    SQD_CHECK_ARG(x && K && y && summary && lse && part, "sqd_sql_fwd: null pointer");
    const int xse = x_nhwc ? 1 : N, xsn = x_nhwc ? E : 1;
    Plan p;
    SQD_CHECK_ARG(make_plan(Q, E, N, &p) == 0, "sqd_sql_fwd: unsupported Q=%d E=%d N=%d", Q, E, N);
    bool launched = false;
    (void)hipGetLastError();
    const bool fwd32 = x_nhwc && p.nchunks32 > 0;
    if (fwd32) {                     // pixel-major x, 32 or 64 features: the 32x32 formulation (lane = pixel)
        // one group of 32 queries per workgroup (grid.z): 132 registers = 3 waves per SIMD whose load / MFMA / exp / LDS phases overlap;
        // with 64 or 128 queries per wave (227 - 256 registers, one or two waves per SIMD) the phases of a tile ran one after the other
        const int eh = E / 32, groups = (Q + 31) / 32;
        const int qt = 1, QP = qt * 32, EP = E + 1;
        const size_t a = (size_t)QP * EP + 4 * QP + (size_t)4 * 32 * PT, m = (size_t)4 * QP * 2 + (size_t)4 * QP * EP;
        const size_t shmem = (a > m ? a : m) * sizeof(float);
#define SQL_FWD32(QT_, EH_)                                                                                                     \
    {                                                                                                                           \
        if (shmem > 48 * 1024)                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sql_fwd32_kernel<QT_, EH_>),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                                  \
        hipLaunchKernelGGL((sql_fwd32_kernel<QT_, EH_>), dim3(p.nchunks32, B, groups), dim3(256), shmem, (hipStream_t)stream,  \
                           x, K, y, part, Q, N, p.tiles32, p.nchunks32);                                                        \
    }
        if (eh == 1) {
            if (qt == 1) SQL_FWD32(1, 1) else if (qt == 2) SQL_FWD32(2, 1) else SQL_FWD32(4, 1)
        } else {
            if (qt == 1) SQL_FWD32(1, 2) else SQL_FWD32(2, 2)
        }
        launched = true;
    } else {
        const size_t fwd_lds = ((size_t)4 * p.QT * 16 * 2 + (size_t)4 * E * (p.QT * 16 + 1)) * sizeof(float);
        SQL_DISPATCH_ALL(sql_fwd_kernel, fwd_lds, x, K, y, part, Q, N, p.steps, p.nchunks, xse, xsn)
    }
    SQD_CHECK_ARG(launched, "sqd_sql_fwd: no kernel instance for QT=%d ET=%d", p.QT, p.ET);
    SQD_CHECK_LAUNCH("sqd_sql_fwd");
    const int nch = fwd32 ? p.nchunks32 : p.nchunks;
    hipLaunchKernelGGL(sql_merge_kernel, dim3(Q, B), dim3(64), (size_t)nch * sizeof(float), (hipStream_t)stream, part, summary, lse, Q, E, nch);
    SQD_CHECK_LAUNCH("sqd_sql_fwd(merge)");
    return SQD_OK;
}

extern "C" int sqd_sql_bwd(const float *x, const float *K, const float *y, const float *g_y, const float *g_summary,
                           const float *summary, const float *lse, float *g_x, float *g_K, float *gk_part, int B, int Q,
                           int E, int N, int x_nhwc, void *stream) {
    return sqd_sql_bwd_amax(x, K, y, g_y, g_summary, summary, lse, g_x, g_K, gk_part, B, Q, E, N, x_nhwc, nullptr, stream);
}
// ... and amax_gx (may be NULL; cleared by the caller): the bit pattern of max |g_x| (sqd.h section 10b)
extern "C" int sqd_sql_bwd_amax(const float *x, const float *K, const float *y, const float *g_y, const float *g_summary,
                                const float *summary, const float *lse, float *g_x, float *g_K, float *gk_part, int B, int Q,
                                int E, int N, int x_nhwc, float *amax_gx, void *stream) {
    SQD_CHECK_ARG(x && K && y && g_summary && summary && lse && g_x && g_K && gk_part, "sqd_sql_bwd: null pointer");
    const int xse = x_nhwc ? 1 : N, xsn = x_nhwc ? E : 1;
    Plan p;
    SQD_CHECK_ARG(make_plan(Q, E, N, &p, B) == 0, "sqd_sql_bwd: unsupported Q=%d E=%d N=%d", Q, E, N);
    SQD_CHECK_ARG((long long)N * 132 * 4 < (1ll << 32), "sqd_sql_bwd: N=%d too large for 32-bit plane offsets", N);
    (void)hipGetLastError();
    {
        const int qt = Q <= 32 ? 1 : Q <= 64 ? 2 : 4, QP = qt * 32, eh = E > 32 ? 2 : 1;
        const size_t shmem = ((size_t)2 * QP * (eh * 32 + 1) + QP * 4 + (size_t)4 * QP * TP) * sizeof(float);
#define SQL_BWD32(QT_, EH_)                                                                                                     \
    {                                                                                                                           \
        if (shmem > 48 * 1024)                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sql_bwd32_kernel<QT_, EH_>),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);                                  \
        hipLaunchKernelGGL((sql_bwd32_kernel<QT_, EH_>), dim3(p.nchunks_bwd, B), dim3(256), shmem, (hipStream_t)stream, x, K, y, g_y, \
                           g_summary, summary, lse, g_x, gk_part, Q, E, N, p.nchunks_bwd, xse, xsn, (unsigned *)amax_gx);               \
    }
        const size_t shmem_q = ((size_t)2 * QP * (eh * 32 + 1) + QP * 4 + (size_t)4 * 32 * TP + (eh == 1 ? 0 : (size_t)4 * eh * 16 * 64) +
                                (size_t)(4 / qt) * 32 * (eh * 32 + 4)) * sizeof(float);      // (+ the shared x tiles of an iteration)
#define SQL_BWD32Q(QT_, EH_)                                                                                                    \
    {                                                                                                                           \
        if (shmem_q > 48 * 1024)                                                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sql_bwd32q_kernel<QT_, EH_>),                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_q);                                \
        hipLaunchKernelGGL((sql_bwd32q_kernel<QT_, EH_>), dim3(p.nchunks_bwd, B), dim3(256), shmem_q, (hipStream_t)stream, x, K, y, g_y, \
                           g_summary, summary, lse, g_x, gk_part, Q, E, N, p.nchunks_bwd, xse, xsn, (unsigned *)amax_gx);               \
    }
        if (eh == 1) {
            if (qt == 1) SQL_BWD32(1, 1) else if (qt == 2) SQL_BWD32Q(2, 1) else SQL_BWD32Q(4, 1)
        } else {
            if (qt == 1) SQL_BWD32(1, 2) else if (qt == 2) SQL_BWD32Q(2, 2) else SQL_BWD32Q(4, 2)
        }
    }
    SQD_CHECK_LAUNCH("sqd_sql_bwd");
    hipLaunchKernelGGL(sql_gk_reduce_kernel, dim3((Q * E + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, gk_part, g_K,
                       Q * E, p.nchunks_bwd);
    SQD_CHECK_LAUNCH("sqd_sql_bwd(reduce)");
    return SQD_OK;
}
