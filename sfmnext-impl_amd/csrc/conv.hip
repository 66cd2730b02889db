// conv.hip — 2-D convolution as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32),
// channels-last activations ([N,H,W,C] memory) and KRSC weights (= torch channels_last weight memory).
// replaces: every nn.Conv2d of the networks (reference networks/resnet_encoder.py, pose_cnn.py,
//           depth_decoder_QTR.py) — forward, data gradient and weight gradient.
//
//   forward : y[m, k]  = sum_{r,s,c} x[pix(m; r,s), c] * w[k, r, s, c]        M = N*Ho*Wo, Kgemm = R*S*C
//   dgrad   : dx[p, c] = sum_{r,s,k} dy[opix(p; r,s), k] * w[k, r, s, c]      M = N*H*W,   Kgemm = R*S*K
//   wgrad   : dw[k,r,s,c] = sum_m dy[m, k] * x[pix(m; r,s), c]                split over pixel ranges
//
// Tiling: 256 threads = 4 wavefronts per workgroup, BM x BN output tile, BK = 16 reduction slice per
// step held in LDS as [row][16 (+4 pad)] for both operands — exactly the memory order of NHWC pixels
// and KRSC filters, so staging is plain 16-byte copies.  A lane feeds the MFMA with 8 consecutive
// reduction elements of its row (two ds_read_b128, conflict-free with the 20-float row pitch); the two
// half-waves own reduction elements 0-7 / 8-15 of the slice, so 8 MFMAs consume one slice.  Global
// loads of slice t+1 are issued before the MFMAs of slice t and written to the other LDS buffer after
// them (register-staged double buffering).  fp32 MFMA is an exact k-ordered fmaf chain, so results
// match a direct fp32 convolution to accumulation-order rounding.
// Roofline: fp32 MFMA, 157.3 TFLOP/s dense.
#include "sqd_common.h"

namespace {
using namespace sqd;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16, LDP = 20;     // reduction slice, LDS row pitch (floats)

struct ConvGeom {
    int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// MODE 0: forward (A rows = output pixels, gather x; B rows = out channels k, reduction over (r,s,c))
// MODE 1: dgrad   (A rows = input pixels, gather dy; B rows = in channels c, reduction over (r,s,k))
template <int MODE, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const float *__restrict__ a_src, const float *__restrict__ wgt,
                                                        const float *__restrict__ bias, float *__restrict__ out, ConvGeom g,
                                                        int act, int zsplits) {
    constexpr int WTM = BM / WGM / 32, WTN = BN / WGN / 32;     // 32x32 MFMA tiles per wave
    static_assert(WGM * WGN == 4 && WTM >= 1 && WTN >= 1, "4 waves per workgroup");
    constexpr int A_F4 = BM * 4 / 256, B_ROWS_PER_PASS = (MODE == 0) ? 64 : 64;
    static_assert(A_F4 >= 1, "BM >= 64");
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDP];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDP];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm0 = (wave / WGN) * (WTM * 32), wn0 = (wave % WGN) * (WTN * 32);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    // problem dims seen by the GEMM
    const int Mrows = MODE == 0 ? g.N * g.Ho * g.Wo : g.N * g.H * g.W;
    const int Ncols = MODE == 0 ? g.K : g.C;                    // output channels of this GEMM
    const int Cred = MODE == 0 ? g.C : g.K;                     // channels reduced per (r,s)
    const int cchunks = Cred / BK;
    const int Tall = g.R * g.S * cchunks;
    // split-K: workgroup z reduces slices [s_beg, s_end) and writes a partial tile (summed by gemm_reduce_kernel)
    const int s_beg = (int)((long long)Tall * blockIdx.z / zsplits), s_end = (int)((long long)Tall * (blockIdx.z + 1) / zsplits);
    const int T = s_end - s_beg;

    // ---- per-thread A staging rows: float4 column c4 of rows (t>>2) + 64*i
    const int c4 = t & 3;
    int an[A_F4], ah[A_F4], aw[A_F4];
    bool aval[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int m = m0 + (t >> 2) + 64 * i;
        aval[i] = m < Mrows;
        const int mm = aval[i] ? m : 0;
        if (MODE == 0) {
            const int wo = mm % g.Wo, t2 = mm / g.Wo;
            an[i] = t2 / g.Ho;
            ah[i] = (t2 % g.Ho) * g.stride - g.pad;
            aw[i] = wo * g.stride - g.pad;
        } else {
            const int wi = mm % g.W, t2 = mm / g.W;
            an[i] = t2 / g.H;
            ah[i] = (t2 % g.H) + g.pad;
            aw[i] = wi + g.pad;
        }
    }
    auto load_a = [&](int step, float4 *ra) {
        const int cc = step % cchunks, rs = step / cchunks;
        const int r = rs / g.S, s = rs - r * g.S;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 0) {
                const int hi = ah[i] + r, wi = aw[i] + s;
                if (aval[i] && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W)
                    v = *reinterpret_cast<const float4 *>(a_src + ((size_t)(an[i] * g.H + hi) * g.W + wi) * g.C + cc * BK + c4 * 4);
            } else {
                const int hn = ah[i] - r, wn = aw[i] - s;              // = ho*stride, wo*stride when divisible
                if (aval[i] && hn >= 0 && wn >= 0 && hn % g.stride == 0 && wn % g.stride == 0) {
                    const int ho = hn / g.stride, wo = wn / g.stride;
                    if (ho < g.Ho && wo < g.Wo)
                        v = *reinterpret_cast<const float4 *>(a_src + ((size_t)(an[i] * g.Ho + ho) * g.Wo + wo) * g.K + cc * BK + c4 * 4);
                }
            }
            ra[i] = v;
        }
    };
    // ---- B staging.  forward: rows = k, 16 consecutive c of filter tap (r,s): float4 copies.
    //      dgrad: rows = c, 16 k's strided by R*S*C: read float4 along c, transpose into LDS.
    constexpr int B_F4 = (BN * 4 + 255) / 256;
    auto load_b = [&](int step, float4 *rb) {
        const int cc = step % cchunks, rs = step / cchunks;
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int idx = t + 256 * i;
            if (MODE == 0) {
                const int row = idx >> 2, q4 = idx & 3;                  // row = out channel, q4 = float4 along c
                if (row < BN && n0 + row < g.K)
                    v = *reinterpret_cast<const float4 *>(wgt + ((size_t)(n0 + row) * g.R * g.S + rs) * g.C + cc * BK + q4 * 4);
            } else {
                const int kk = idx & 15, cq = idx >> 4;                  // kk = k inside the slice, cq = float4 of in-channels
                if (cq * 4 < BN && n0 + cq * 4 < g.C)
                    v = *reinterpret_cast<const float4 *>(wgt + ((size_t)(cc * BK + kk) * g.R * g.S + rs) * g.C + n0 + cq * 4);
            }
            rb[i] = v;
        }
    };
    auto store_ab = [&](int buf, const float4 *ra, const float4 *rb) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4 *>(&As[buf][(t >> 2) + 64 * i][c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int idx = t + 256 * i;
            if (MODE == 0) {
                const int row = idx >> 2, q4 = idx & 3;
                if (row < BN) *reinterpret_cast<float4 *>(&Bs[buf][row][q4 * 4]) = rb[i];
            } else {
                const int kk = idx & 15, cq = idx >> 4;
                if (cq * 4 < BN) {
                    Bs[buf][cq * 4 + 0][kk] = rb[i].x;
                    Bs[buf][cq * 4 + 1][kk] = rb[i].y;
                    Bs[buf][cq * 4 + 2][kk] = rb[i].z;
                    Bs[buf][cq * 4 + 3][kk] = rb[i].w;
                }
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

    float4 ra[A_F4], rb[B_F4];
    if (T > 0) {
        load_a(s_beg, ra);
        load_b(s_beg, rb);
        store_ab(0, ra, rb);
    }
    __syncthreads();
    const int row = lane & 31, h = lane >> 5;
    for (int step = 0; step < T; ++step) {
        const int cur = step & 1;
        if (step + 1 < T) {                       // prefetch the next slice into registers
            load_a(s_beg + step + 1, ra);
            load_b(s_beg + step + 1, rb);
        }
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
        if (step + 1 < T) store_ab(cur ^ 1, ra, rb);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int col = n0 + wn0 + j * 32 + row;
        if (col >= Ncols) continue;
        const float bv = (MODE == 0 && bias && zsplits == 1) ? bias[col] : 0.f;
        float *dst = out + (size_t)blockIdx.z * Mrows * Ncols;          // zsplits > 1: `out` is the partial workspace
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < Mrows) {
                    float v = acc[i][j][e] + bv;
                    if (MODE == 0 && act == 1 && zsplits == 1) v = v > 0.f ? v : 0.f;
                    dst[(size_t)m * Ncols + col] = v;
                }
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
            ra[i] = (mv && (q + 16 * i) * 4 < BM && kq < g.K) ? *reinterpret_cast<const float4 *>(dy + (size_t)m * g.K + kq)
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_P; ++i) {
            const int cq = c0 + (q + 16 * i) * 4;
            rb[i] = (xin && (q + 16 * i) * 4 < BN && cq < g.C)
                        ? *reinterpret_cast<const float4 *>(x + ((size_t)(n * g.H + hi) * g.W + wi) * g.C + cq)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
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

__global__ __launch_bounds__(256) void split_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, size_t n,
                                                           int splits) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<const float4 *>(part)[i];
        for (int s = 1; s < splits; ++s) {
            const float4 b = reinterpret_cast<const float4 *>(part + (size_t)s * n)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4 *>(out)[i] = a;
    }
}

// sum of the split-K partial tiles (+ bias, + activation): part [Z][M*Ncols] -> out [M*Ncols]
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                          float *__restrict__ out, size_t n, int Z, int Ncols, int act) {
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
        if (act == 1) {
            a.x = a.x > 0.f ? a.x : 0.f; a.y = a.y > 0.f ? a.y : 0.f; a.z = a.z > 0.f ? a.z : 0.f; a.w = a.w > 0.f ? a.w : 0.f;
        }
        reinterpret_cast<float4 *>(out)[i] = a;
    }
}

// column sums of a [M, K] matrix (bias gradient), deterministic two-level
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ dy, float *__restrict__ part, int M, int K,
                                                     int rows_per_block) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int k = threadIdx.x; k < K; k += 256) {
        float s = 0.f;
        for (int m = r0; m < r1; ++m) s += dy[(size_t)m * K + k];
        part[(size_t)blockIdx.x * K + k] = s;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, float *__restrict__ out, int nblk, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * K + k];
    out[k] = s;
}

int check_geom(const char *who, const ConvGeom &g) {
    SQD_CHECK_ARG(g.N > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.K > 0 && g.R > 0 && g.S > 0 && g.stride > 0 && g.pad >= 0,
                  "%s: bad geometry", who);
    SQD_CHECK_ARG(g.Ho == (g.H + 2 * g.pad - g.R) / g.stride + 1 && g.Wo == (g.W + 2 * g.pad - g.S) / g.stride + 1,
                  "%s: Ho/Wo inconsistent with H/W/R/S/stride/pad", who);
    return SQD_OK;
}
}  // namespace

extern "C" int sqd_conv_supported(int C, int K) { return (C % 16 == 0 && K % 16 == 0) ? 1 : 0; }

struct GemmPlan {
    int bm, bn, z;
    int64_t ws_floats;
};
// tile / split-K choice: enough workgroups to cover 256 CUs a few times over, partial workspace <= 64 MB
static GemmPlan plan_gemm(int mode, const ConvGeom &g) {
    const int Mrows = mode == 0 ? g.N * g.Ho * g.Wo : g.N * g.H * g.W;
    const int Ncols = mode == 0 ? g.K : g.C;
    const int T = g.R * g.S * ((mode == 0 ? g.C : g.K) / BK);
    GemmPlan p;
    p.bn = Ncols >= 128 ? 128 : Ncols >= 64 ? 64 : 32;
    p.bm = 128;
    auto wgs = [&](int bm, int bn) { return ((Mrows + bm - 1) / bm) * ((Ncols + bn - 1) / bn); };
    if (wgs(128, p.bn) < 512 && p.bn >= 64) p.bm = 64;
    if (p.bm == 64 && p.bn == 128 && wgs(64, 128) < 512) p.bn = 64;
    int z = 1;
    const int w = wgs(p.bm, p.bn);
    if (w < 512) {
        z = (768 + w - 1) / w;
        if (z > 16) z = 16;
        if (z > T / 4) z = T / 4 > 1 ? T / 4 : 1;
        while (z > 1 && (int64_t)z * Mrows * Ncols * 4 > (64ll << 20)) --z;
        if (((int64_t)Mrows * Ncols) % 4 != 0) z = 1;
    }
    p.z = z;
    p.ws_floats = z > 1 ? (int64_t)z * Mrows * Ncols : 0;
    return p;
}

#define LAUNCH_GEMM(MODE, BM, BN, WGM, WGN)                                                                               \
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, BM, BN, WGM, WGN>), dim3((Mrows + BM - 1) / BM, (Ncols + BN - 1) / BN, p.z), \
                       dim3(256), 0, st, a_src, w, bias, dst, g, act, p.z)
#define DISPATCH_GEMM(MODE)                                          \
    if (p.bm == 128 && p.bn == 128) LAUNCH_GEMM(MODE, 128, 128, 2, 2); \
    else if (p.bm == 128 && p.bn == 64) LAUNCH_GEMM(MODE, 128, 64, 2, 2); \
    else if (p.bm == 128) LAUNCH_GEMM(MODE, 128, 32, 4, 1);          \
    else if (p.bn == 128) LAUNCH_GEMM(MODE, 64, 128, 2, 2);          \
    else LAUNCH_GEMM(MODE, 64, 64, 2, 2);

static int launch_gemm(int mode, const float *a_src, const float *w, const float *bias, float *out, float *ws, const ConvGeom &g,
                       int act, void *stream) {
    const int Mrows = mode == 0 ? g.N * g.Ho * g.Wo : g.N * g.H * g.W;
    const int Ncols = mode == 0 ? g.K : g.C;
    const GemmPlan p = plan_gemm(mode, g);
    if (p.z > 1 && !ws) {
        sqd::set_error("sqd_conv: this shape needs a split-K workspace of %lld floats (sqd_conv_plan)", (long long)p.ws_floats);
        return SQD_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    float *dst = p.z > 1 ? ws : out;
    (void)hipGetLastError();
    if (mode == 0) { DISPATCH_GEMM(0) } else { DISPATCH_GEMM(1) }
    if (p.z > 1) {
        const size_t n = (size_t)Mrows * Ncols;
        size_t nb = (n / 4 + 255) / 256;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, st, ws, mode == 0 ? bias : nullptr,
                           out, n, p.z, Ncols, mode == 0 ? act : 0);
    }
    return SQD_OK;
}

// workspace (floats) sqd_conv_fwd (mode 0) / sqd_conv_dgrad (mode 1) need for this geometry (0 = none)
extern "C" int sqd_conv_plan(int mode, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                             int64_t *ws_floats) {
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (ws_floats) *ws_floats = plan_gemm(mode, g).ws_floats;
    return SQD_OK;
}

// x [N,H,W,C], w [K,R,S,C], bias [K] or NULL -> y [N,Ho,Wo,K]; act: 0 none, 1 ReLU.  C, K multiples of 16.
extern "C" int sqd_conv_fwd(const float *x, const float *w, const float *bias, float *y, float *ws, int N, int H, int W, int C, int K,
                            int R, int S, int stride, int pad, int Ho, int Wo, int act, void *stream) {
    SQD_CHECK_ARG(x && w && y, "sqd_conv_fwd: null pointer");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_fwd", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(C % 16 == 0, "sqd_conv_fwd: C=%d must be a multiple of 16", C);
    if (launch_gemm(0, x, w, bias, y, ws, g, act, stream)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_conv_fwd");
    return SQD_OK;
}

// dy [N,Ho,Wo,K], w [K,R,S,C] -> dx [N,H,W,C]
extern "C" int sqd_conv_dgrad(const float *dy, const float *w, float *dx, float *ws, int N, int H, int W, int C, int K, int R,
                              int S, int stride, int pad, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(dy && w && dx, "sqd_conv_dgrad: null pointer");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_dgrad", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(K % 16 == 0 && C % 4 == 0, "sqd_conv_dgrad: K=%d must be a multiple of 16 and C=%d of 4", K, C);
    if (launch_gemm(1, dy, w, nullptr, dx, ws, g, 0, stream)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_conv_dgrad");
    return SQD_OK;
}

extern "C" int sqd_conv_wgrad_plan(int N, int Ho, int Wo, int C, int K, int R, int S, int *splits, int64_t *part_floats) {
    const int M = N * Ho * Wo;
    const int bm = K >= 128 ? 128 : 64, bn = C >= 128 ? 128 : 64;
    const int tiles = ((K + bm - 1) / bm) * ((C + bn - 1) / bn) * R * S;
    int sp = (1536 + tiles - 1) / tiles;                         // aim at ~1.5k workgroups
    const int max_by_px = (M + 255) / 256;                      // at least 256 pixels per split
    if (sp > max_by_px) sp = max_by_px;
    const int64_t wsz = (int64_t)K * R * S * C;
    while (sp > 1 && (int64_t)sp * wsz * 4 > (64ll << 20)) --sp; // partial buffer <= 64 MB
    if (sp < 1) sp = 1;
    if (splits) *splits = sp;
    if (part_floats) *part_floats = (int64_t)sp * wsz;
    return SQD_OK;
}

// dy [N,Ho,Wo,K], x [N,H,W,C] -> dw [K,R,S,C]; dbias [K] (may be NULL); part: workspace of sqd_conv_wgrad_plan floats
// (+ colsum scratch: ceil(M/1024)*K floats appended when dbias is requested)
extern "C" int sqd_conv_wgrad(const float *dy, const float *x, float *dw, float *dbias, float *part, int N, int H, int W, int C,
                              int K, int R, int S, int stride, int pad, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(dy && x && dw && part, "sqd_conv_wgrad: null pointer");
    ConvGeom g = {N, H, W, C, K, R, S, stride, pad, Ho, Wo};
    if (check_geom("sqd_conv_wgrad", g)) return SQD_EINVAL;
    SQD_CHECK_ARG(C % 4 == 0 && K % 4 == 0, "sqd_conv_wgrad: C=%d and K=%d must be multiples of 4", C, K);
    int splits;
    int64_t pf;
    sqd_conv_wgrad_plan(N, Ho, Wo, C, K, R, S, &splits, &pf);
    const int M = N * Ho * Wo;
    int pps = (M + splits - 1) / splits;
    pps = ((pps + BK - 1) / BK) * BK;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    const bool bigk = K >= 128, bigc = C >= 128;
    const int bm = bigk ? 128 : 64, bn = bigc ? 128 : 64;
    const dim3 grid(((K + bm - 1) / bm) * ((C + bn - 1) / bn), splits, R * S);
    if (bigk && bigc) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), grid, dim3(256), 0, st, dy, x, part, g, pps);
    else if (bigk) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64>), grid, dim3(256), 0, st, dy, x, part, g, pps);
    else if (bigc) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), grid, dim3(256), 0, st, dy, x, part, g, pps);
    else hipLaunchKernelGGL((conv_wgrad_kernel<64, 64>), grid, dim3(256), 0, st, dy, x, part, g, pps);
    const size_t wsz = (size_t)K * R * S * C;
    size_t nb = (wsz / 4 + 255) / 256;
    hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, st, part, dw, wsz, splits);
    if (dbias) {
        float *cpart = part + (size_t)splits * wsz;
        const int rpb = 1024, nblk = (M + rpb - 1) / rpb;
        hipLaunchKernelGGL(colsum_kernel, dim3(nblk), dim3(256), 0, st, dy, cpart, M, K, rpb);
        hipLaunchKernelGGL(colsum_final_kernel, dim3((K + 255) / 256), dim3(256), 0, st, cpart, dbias, nblk, K);
    }
    SQD_CHECK_LAUNCH("sqd_conv_wgrad");
    return SQD_OK;
}
