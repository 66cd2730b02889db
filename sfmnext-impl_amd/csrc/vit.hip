// vit.hip — the two token-wise blocks of the post-norm TransformerEncoderLayer the depth head runs on its 120 patch tokens
// (reference networks/depth_decoder_QTR.py:31-32,47: nn.TransformerEncoderLayer(E, 4 heads, dim_feedforward 1024 | 512, ReLU,
// dropout 0.1), 4 layers), as fused kernels.  Tokens are rows of a [rows = S*B, E] matrix, E in {16, 32, 64}.
//
//   add + dropout + LayerNorm :  out = LN(x + mask * scale * y)             (norm1 / norm2 with dropout1 / dropout2)
//   feed-forward              :  y = W2 . (mask * scale * relu(W1 . x + b1)) + b2   (linear1, ReLU, dropout, linear2)
//
// The layer is latency-bound, not throughput-bound (46 K token elements, 190 MFLOP): ATen runs its post-attention half as
// ~45 launches forward+backward per layer; these kernels make it 3 + 4.  What matters is the length of the dependent chain
// inside a launch, so the work is cut fine: one wave = 32 tokens x 32 hidden units, every global load it needs issued up
// front, grid = token tiles x hidden groups (360 workgroups for 1440 tokens, F = 1024).  The feed-forward runs on
// v_mfma_f32_32x32x2_f32 in the transposed orientation (hidden units x tokens) so that the second product consumes the
// first one's accumulator registers directly; the hidden activations are never stored — the backward recomputes them.
// Sums over hidden groups (y, g_x) are left as partials that the consuming add+LayerNorm kernel adds while loading;
// sums over token tiles (weight gradients) go through one multi-segment column-sum launch.  Everything is fixed-order.
// Dropout masks are bytes drawn by the caller (torch's generator: graph-safe), so training statistics are torch's.
#include "sqd_common.h"

namespace {
using namespace sqd;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------------
// out = LayerNorm(x + drop(sum_p y[p] + ybias)) over the last dimension E; one half-wave (32 lanes) per row, lane = feature.
// saves xhat [rows,E] and rstd [rows] for the backward.
// ---------------------------------------------------------------------------------------------------
template <int LW>                                      // lanes per row: 32 (E <= 32: one half-wave per row) or 64 (E <= 64)
__global__ __launch_bounds__(256) void addln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ y, int nparts,
                                                        const float *__restrict__ ybias, const unsigned char *__restrict__ mask,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float *__restrict__ out, float *__restrict__ xhat, float *__restrict__ rstd_out,
                                                        int rows, int E, float scale, float eps) {
    const int lane = threadIdx.x & (LW - 1), row = blockIdx.x * (256 / LW) + threadIdx.x / LW;
    if (row >= rows) return;
    const bool on = lane < E;
    const size_t o = (size_t)row * E + lane, pstride = (size_t)rows * E;
    float z = 0.f;
    if (on) {
        float yv = ybias ? ybias[lane] : 0.f;
        for (int p = 0; p < nparts; ++p) yv += y[o + p * pstride];
        if (mask) yv = mask[o] ? yv * scale : 0.f;
        z = x[o] + yv;
    }
    float s = z;
#pragma unroll
    for (int k = LW / 2; k > 0; k >>= 1) s += __shfl_xor(s, k, LW);
    const float mean = s / (float)E;
    const float d = on ? z - mean : 0.f;
    float v = d * d;
#pragma unroll
    for (int k = LW / 2; k > 0; k >>= 1) v += __shfl_xor(v, k, LW);
    const float rstd = rsqrtf(v / (float)E + eps);
    if (on) {
        const float xh = d * rstd;
        xhat[o] = xh;
        out[o] = fmaf(xh, gamma[lane], beta[lane]);
    }
    if (lane == 0) rstd_out[row] = rstd;
}

// g = g_out + sum_p g_extra[p]  ->  g_x (= dz), g_y (= dz * mask * scale), per-block partials of dgamma / dbeta (part [nblk][2][E])
constexpr int LN_ROWS = 16;     // rows per block of the backward
template <int LW>
__global__ __launch_bounds__(256) void addln_bwd_kernel(const float *__restrict__ g, const float *__restrict__ g_extra, int nextra,
                                                        const float *__restrict__ xhat, const float *__restrict__ rstd,
                                                        const unsigned char *__restrict__ mask, const float *__restrict__ gamma,
                                                        float *__restrict__ gx, float *__restrict__ gy, float *__restrict__ part,
                                                        int rows, int E, float scale) {
    constexpr int RP = 256 / LW;                        // rows in flight per block
    __shared__ float red[2][RP][LW];
    const int lane = threadIdx.x & (LW - 1), sub = threadIdx.x / LW;
    const bool on = lane < E;
    const float ga = on ? gamma[lane] : 0.f;
    const size_t pstride = (size_t)rows * E;
    float dg = 0.f, db = 0.f;
    const int r0 = blockIdx.x * LN_ROWS, r1 = min(rows, r0 + LN_ROWS);
#pragma unroll
    for (int k = 0; k < LN_ROWS / RP; ++k) {
        const int row = r0 + sub + RP * k;
        if (row >= r1) break;
        const size_t o = (size_t)row * E + lane;
        float gv = on ? g[o] : 0.f;
        if (on)
            for (int p = 0; p < nextra; ++p) gv += g_extra[o + p * pstride];
        const float xh = on ? xhat[o] : 0.f;
        const float dxh = gv * ga;
        float s1 = dxh, s2 = dxh * xh;
#pragma unroll
        for (int q = LW / 2; q > 0; q >>= 1) {
            s1 += __shfl_xor(s1, q, LW);
            s2 += __shfl_xor(s2, q, LW);
        }
        const float dz = rstd[row] * (dxh - s1 / (float)E - xh * (s2 / (float)E));
        if (on) {
            gx[o] = dz;
            gy[o] = mask ? (mask[o] ? dz * scale : 0.f) : dz;
        }
        dg = fmaf(gv, xh, dg);
        db += gv;
    }
    red[0][sub][lane] = dg;
    red[1][sub][lane] = db;
    __syncthreads();
    if (threadIdx.x < 2 * LW) {
        const int w = threadIdx.x / LW;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < RP; ++k) a += red[w][k][lane];
        if (on) part[((size_t)blockIdx.x * 2 + w) * E + lane] = a;
    }
}

// ---------------------------------------------------------------------------------------------------
// several column sums in one launch: segment s: dst[c] = sum_r src[r * ncols + c] (fixed order), ncols % 4 == 0.
// tr > 0: the columns are a [ncols / tr][tr] matrix written transposed ([tr][ncols / tr]).
// block = 32 float4 columns x 8 row groups.
// ---------------------------------------------------------------------------------------------------
constexpr int CS_MAXSEG = 12;
struct ColsumSegs {
    const float *src[CS_MAXSEG];
    float *dst[CS_MAXSEG];
    int nrows[CS_MAXSEG], ncols[CS_MAXSEG], tr[CS_MAXSEG], blk0[CS_MAXSEG + 1];
    int nseg;
};
// CW float4 columns x 256 / CW row groups per workgroup: 32 x 8 for the few partial rows the token-wise blocks produce, 8 x 32 when
// a segment has hundreds of rows (the ConvNeXt LayerNorm backward at 640 partial rows: 18 workgroups of 80 sequential loads per thread
// took 15 us; 72 workgroups of 20 take 8)
template <int CW>
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumSegs S) {
    constexpr int RG = 256 / CW;
    __shared__ float4 red[RG][CW];
    int s = 0;
    while (s + 1 < S.nseg && (int)blockIdx.x >= S.blk0[s + 1]) ++s;
    const int cl = threadIdx.x % CW, rg = threadIdx.x / CW;
    const int c4 = ((int)blockIdx.x - S.blk0[s]) * CW + cl;
    const int ncols = S.ncols[s], nrows = S.nrows[s];
    const bool on = c4 * 4 < ncols;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
        const float *p = S.src[s] + (size_t)c4 * 4;
        for (int r = rg; r < nrows; r += RG) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)r * ncols);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[rg][cl] = a;
    __syncthreads();
    if (RG > 32) {      // (many row groups: a fixed-order tree down to 32 of them before the serial tail)
        for (int w = RG / 2; w >= 32; w >>= 1) {
            if (rg < w) {
                const float4 v = red[rg + w][cl];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                red[rg][cl] = a;
            }
            __syncthreads();
        }
    }
    if (rg == 0 && on) {
#pragma unroll
        for (int k = 1; k < (RG > 32 ? 32 : RG); ++k) {
            const float4 v = red[k][cl];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float *d = S.dst[s];
        const int tr = S.tr[s];
        if (tr == 0) *reinterpret_cast<float4 *>(d + (size_t)c4 * 4) = a;
        else {
            const int outer = ncols / tr;
            const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c4 * 4 + q;
                d[(size_t)(c % tr) * outer + c / tr] = v[q];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// feed-forward.  Workgroup (tx, gy): tokens [32 tx, 32 tx + 32), hidden units [128 gy, 128 gy + 128): wave w owns the
// chunk f0 = 128 gy + 32 w.  Orientation: hT[f, t] (rows = hidden units, columns = tokens).  Reduction index of the first
// product is the feature e, split between the half-waves as e = 16*half + s (16 consecutive floats per lane: float4 loads).
// ---------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void loadn(const float *__restrict__ p, int E, int kk, bool valid, float (&v)[N]) {
    // N consecutive features starting at N*kk (zeros beyond E)
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const int e = N * kk + 4 * q;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && e < E) t = *reinterpret_cast<const float4 *>(p + e);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

// per-lane side data of a chunk in accumulator layout: register r <-> hidden unit f0 + acc_row(r, half); r = 4*g4 + j are 4
// consecutive units, so bias and mask come as one float4 / one dword per g4
struct ChunkSide {
    float4 b1[4];
    unsigned m[4];
};
__device__ __forceinline__ void load_side(const float *__restrict__ b1, const unsigned char *__restrict__ mask, int F, int f0, int h,
                                          size_t tok, bool tv, ChunkSide &sd) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int f = f0 + 8 * g4 + 4 * h;
        sd.b1[g4] = f < F ? *reinterpret_cast<const float4 *>(b1 + f) : make_float4(0.f, 0.f, 0.f, 0.f);
        sd.m[g4] = 0x01010101u;
        if (mask) sd.m[g4] = (tv && f < F) ? *reinterpret_cast<const unsigned *>(mask + tok * F + f) : 0u;
    }
}
__device__ __forceinline__ float side_b1(const ChunkSide &sd, int r) {
    const float4 b = sd.b1[r >> 2];
    return (r & 3) == 0 ? b.x : (r & 3) == 1 ? b.y : (r & 3) == 2 ? b.z : b.w;
}
__device__ __forceinline__ bool side_keep(const ChunkSide &sd, int r) { return (sd.m[r >> 2] >> (8 * (r & 3))) & 0xffu; }

// EB = blocks of 32 features: E <= 32 EB (1: the 16 / 32-wide heads of the published configurations; 2: model_dim 64).  The first
// product reduces over 32 EB features (16 EB per half-wave), the second one writes EB accumulator tiles of 32 output features.
template <int EB>
__global__ __launch_bounds__(256) void ffn_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W1,
                                                      const float *__restrict__ b1, const float *__restrict__ W2,
                                                      const unsigned char *__restrict__ mask, float *__restrict__ ypart, int rows,
                                                      int E, int F, float scale) {
    constexpr int KE = 16 * EB, EW = 32 * EB;
    __shared__ float yred[4][EW][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * 32, tok = t0 + i;
    const bool tv = tok < rows;
    const int f0 = (blockIdx.y * 4 + wave) * 32;
    f32x16 yT[EB];                                       // yT[eb][e', t]: rows = output features 32 eb + e', columns = tokens
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) yT[eb][r] = 0.f;
    if (f0 < F) {                                        // wave-uniform
        float xt[KE], w1[KE];
        float4 w2[EB][4];
        ChunkSide sd;
        loadn<KE>(x + (size_t)tok * E, E, h, tv, xt);
        loadn<KE>(W1 + (size_t)(f0 + i) * E, E, h, f0 + i < F, w1);
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int f = f0 + 8 * g4 + 4 * h;       // k-slot (r = 4*g4 + j, half) of the second product is hidden unit f + j
                w2[eb][g4] = (32 * eb + i < E && f < F) ? *reinterpret_cast<const float4 *>(W2 + (size_t)(32 * eb + i) * F + f) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        load_side(b1, mask, F, f0, h, (size_t)tok, tv, sd);
        f32x16 hT;
#pragma unroll
        for (int r = 0; r < 16; ++r) hT[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KE; ++s) hT = mfma32(w1[s], xt[s], hT);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = hT[r] + side_b1(sd, r);
            hT[r] = (v > 0.f && side_keep(sd, r)) ? v * scale : 0.f;      // units >= F: W1 row, bias and W2 column are zero
        }
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                yT[eb] = mfma32(w2[eb][g4].x, hT[4 * g4 + 0], yT[eb]);
                yT[eb] = mfma32(w2[eb][g4].y, hT[4 * g4 + 1], yT[eb]);
                yT[eb] = mfma32(w2[eb][g4].z, hT[4 * g4 + 2], yT[eb]);
                yT[eb] = mfma32(w2[eb][g4].w, hT[4 * g4 + 3], yT[eb]);
            }
    }
    // add the 4 waves (fixed order) -> this hidden group's partial of y
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) yred[wave][32 * eb + acc_row(r, h)][i] = yT[eb][r];
    __syncthreads();
    float *yo = ypart + (size_t)blockIdx.y * rows * E;
    for (int idx = threadIdx.x; idx < 32 * EW; idx += 256) {
        const int t = idx / EW, e = idx % EW;
        if (t0 + t < rows && e < E) yo[(size_t)(t0 + t) * E + e] = ((yred[0][e][t] + yred[1][e][t]) + yred[2][e][t]) + yred[3][e][t];
    }
}

// backward: g_y [rows,E] -> gxpart [G][rows,E] (partials over the hidden groups); per-token-tile partials pW1 [T][F,E],
// pW2 [T][F,E] (dW2 transposed), pb1 [T][F], pb2 [T][E]
template <int EB>
__global__ __launch_bounds__(256) void ffn_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                      const float *__restrict__ W1, const float *__restrict__ b1,
                                                      const float *__restrict__ W2, const unsigned char *__restrict__ mask,
                                                      float *__restrict__ gxpart, float *__restrict__ pW1, float *__restrict__ pb1,
                                                      float *__restrict__ pW2, float *__restrict__ pb2, int rows, int E, int F,
                                                      float scale) {
    constexpr int KE = 16 * EB, EW = 32 * EB;
    __shared__ float xs[32][EW + 1], gs[32][EW + 1];     // the token tile: x and g_y, [token][feature], zero padded
    __shared__ float xred[4][EW][33];                    // g_x^T partials of the 4 waves
    __shared__ float tile[4][2][32][36];                 // per wave: [0] hT (after relu*drop), [1] dhT; [f][token]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * 32, tok = t0 + i;
    const bool tv = tok < rows;
    const int f0 = (blockIdx.y * 4 + wave) * 32;
    const bool active = f0 < F;                          // wave-uniform
    // global loads of the chunk first (independent of the staging below)
    float w1row[KE], w2col[KE], w1col[EB][16];
    ChunkSide sd;
    if (active) {
        loadn<KE>(W1 + (size_t)(f0 + i) * E, E, h, f0 + i < F, w1row);
#pragma unroll
        for (int s = 0; s < KE; ++s) {
            const int e = KE * h + s;
            w2col[s] = (f0 + i < F && e < E) ? W2[(size_t)e * F + f0 + i] : 0.f;            // A of dhT: row f = lane, k = e'
        }
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + acc_row(r, h);
                w1col[eb][r] = (f < F && 32 * eb + i < E) ? W1[(size_t)f * E + 32 * eb + i] : 0.f;     // A of g_x^T tile eb: row e = 32 eb + lane, k = f
            }
        load_side(b1, mask, F, f0, h, (size_t)tok, tv, sd);
    }
    for (int idx = threadIdx.x; idx < 32 * EW; idx += 256) {
        const int t = idx / EW, e = idx % EW;
        const bool ok = t0 + t < rows && e < E;
        xs[t][e] = ok ? x[(size_t)(t0 + t) * E + e] : 0.f;
        gs[t][e] = ok ? gy[(size_t)(t0 + t) * E + e] : 0.f;
    }
    __syncthreads();
    f32x16 gxT[EB];                                      // g_x^T[32 eb + e, t]
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) gxT[eb][r] = 0.f;
    if (active) {
        float(*hs)[36] = tile[wave][0], (*ds)[36] = tile[wave][1];
        f32x16 hT, dT;
#pragma unroll
        for (int r = 0; r < 16; ++r) hT[r] = dT[r] = 0.f;
        // hT[f, t] = sum_e W1[f, e] x[t, e];  dhT[f, t] = sum_e' W2[e', f] gy[t, e']
#pragma unroll
        for (int s = 0; s < KE; ++s) {
            hT = mfma32(w1row[s], xs[i][KE * h + s], hT);
            dT = mfma32(w2col[s], gs[i][KE * h + s], dT);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = hT[r] + side_b1(sd, r);
            const bool keep = v > 0.f && side_keep(sd, r);
            const float hv = keep ? v * scale : 0.f;     // dropped, scaled activation (operand of dW2)
            const float dv = keep ? dT[r] * scale : 0.f; // gradient w.r.t. the pre-activation
            dT[r] = dv;
            hs[acc_row(r, h)][i] = hv;
            ds[acc_row(r, h)][i] = dv;
        }
        // g_x^T[e, t] += sum_f W1[f, e] dhT[f, t] : k-slot (r, half) is hidden unit f0 + acc_row(r, half)
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int r = 0; r < 16; ++r) gxT[eb] = mfma32(w1col[eb][r], dT[r], gxT[eb]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // dW1[f, e] = sum_t dhT[f, t] x[t, e],  dW2[e', f] = sum_t gy[t, e'] hT[f, t] — contraction over the 32 tokens;
        // k-step (gq, j): half-wave 0 takes token 8gq+j, half-wave 1 token 8gq+4+j
        f32x16 aW1[EB], aW2T[EB];                         // aW1: rows f, columns e;  aW2T: rows f, columns e'
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int r = 0; r < 16; ++r) aW1[eb][r] = aW2T[eb][r] = 0.f;
        float db1 = 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int tk = 8 * gq + 4 * h;
            const float4 d4 = *reinterpret_cast<const float4 *>(&ds[i][tk]);
            const float4 h4 = *reinterpret_cast<const float4 *>(&hs[i][tk]);
            const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    aW1[eb] = mfma32(dv[j], xs[tk + j][32 * eb + i], aW1[eb]);                  // B[k = token][col = e]
                    aW2T[eb] = mfma32(hv[j], gs[tk + j][32 * eb + i], aW2T[eb]);
                }
                db1 += dv[j];
            }
        }
        db1 += __shfl_xor(db1, 32, 64);
        // each (token tile, hidden unit) is written by exactly one wave: no cross-wave add for dW1 / dW2 / db1
        float *w1o = pW1 + (size_t)blockIdx.x * F * E, *w2o = pW2 + (size_t)blockIdx.x * F * E;
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + acc_row(r, h);
                if (f < F && 32 * eb + i < E) {
                    w1o[(size_t)f * E + 32 * eb + i] = aW1[eb][r];
                    w2o[(size_t)f * E + 32 * eb + i] = aW2T[eb][r];
                }
            }
        if (h == 0 && f0 + i < F) pb1[(size_t)blockIdx.x * F + f0 + i] = db1;
    }
    // g_x: add the 4 waves -> this hidden group's partial; db2 partial = column sums of gy over this tile's tokens
#pragma unroll
    for (int eb = 0; eb < EB; ++eb)
#pragma unroll
        for (int r = 0; r < 16; ++r) xred[wave][32 * eb + acc_row(r, h)][i] = gxT[eb][r];
    __syncthreads();
    float *go = gxpart + (size_t)blockIdx.y * rows * E;
    for (int idx = threadIdx.x; idx < 32 * EW; idx += 256) {
        const int t = idx / EW, e = idx % EW;
        if (t0 + t < rows && e < E) go[(size_t)(t0 + t) * E + e] = ((xred[0][e][t] + xred[1][e][t]) + xred[2][e][t]) + xred[3][e][t];
    }
    if (blockIdx.y == 0 && threadIdx.x < E) {
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < 32; ++t) s += gs[t][threadIdx.x];
        pb2[(size_t)blockIdx.x * E + threadIdx.x] = s;
    }
}

int vit_check(const char *who, int rows, int E) {
    SQD_CHECK_ARG(rows > 0 && E >= 4 && E <= 64 && E % 4 == 0, "%s: rows=%d, E=%d (E must be a multiple of 4, at most 64)", who, rows, E);
    return SQD_OK;
}
}  // namespace

// embedding widths: any multiple of 4 up to 64 — the token-wise kernels mask the lanes / reduction slots beyond E (one half-wave per row up
// to 32 features, a whole wave beyond; feed-forward blocks of 32 features).  The reference's args files use 32 (KITTI), 64 and 56
// (args_files/args_cityscapes_train.txt:9: --model_dim 56).
// ---------------------------------------------------------------------------------------------------------------------------------
// patch tokens + positional encodings (reference networks/depth_decoder_QTR.py:49-51: embedding.flatten(2) + positional_encodings[:T].T,
// then .permute(2, 0, 1)): the embedding convolution's output is channels-last memory [B][T][E]; the encoder wants [T][B][E].  One launch
// each way instead of add + layout copy forward and sum + zero-fill + slice copy + layout copy backward.
//   fwd: out[t][b][e] = emb[b][t][e] + pos[t][e]
//   bwd: g_emb[b][t][e] = g[t][b][e];  g_pos[t][e] = sum_b g[t][b][e] for t < T, 0 for the rows of the table the step did not use
namespace {
__global__ __launch_bounds__(256) void tokens_pos_fwd_kernel(const float *__restrict__ emb, const float *__restrict__ pos, float *__restrict__ out,
                                                             int B, int T, int E) {
    const int n = B * T * E;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int e = idx % E, b = (idx / E) % B, t = idx / (E * B);
        out[idx] = emb[((size_t)b * T + t) * E + e] + pos[(size_t)t * E + e];
    }
}
__global__ __launch_bounds__(256) void tokens_pos_bwd_kernel(const float *__restrict__ g, float *__restrict__ g_emb, float *__restrict__ g_pos,
                                                             int B, int T, int E, int Tmax) {
    const int n = Tmax * E;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int e = idx % E, t = idx / E;
        float s = 0.f;
        if (t < T)
            for (int b = 0; b < B; ++b) {                       // (fixed order: deterministic)
                const float v = g[((size_t)t * B + b) * E + e];
                g_emb[((size_t)b * T + t) * E + e] = v;
                s += v;
            }
        g_pos[idx] = s;
    }
}
}  // namespace
extern "C" int sqd_tokens_pos_fwd(const float *emb, const float *pos, float *out, int B, int T, int E, void *stream) {
    SQD_CHECK_ARG(emb && pos && out && B > 0 && T > 0 && E > 0, "sqd_tokens_pos_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(tokens_pos_fwd_kernel, dim3((B * T * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, emb, pos, out, B, T, E);
    SQD_CHECK_LAUNCH("sqd_tokens_pos_fwd");
    return SQD_OK;
}
extern "C" int sqd_tokens_pos_bwd(const float *g, float *g_emb, float *g_pos, int B, int T, int E, int Tmax, void *stream) {
    SQD_CHECK_ARG(g && g_emb && g_pos && B > 0 && T > 0 && E > 0 && Tmax >= T, "sqd_tokens_pos_bwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(tokens_pos_bwd_kernel, dim3((Tmax * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, g_emb, g_pos, B, T, E, Tmax);
    SQD_CHECK_LAUNCH("sqd_tokens_pos_bwd");
    return SQD_OK;
}

// the first Q tokens of the encoder's output as the query matrix [B,Q,E] (reference networks/depth_decoder_QTR.py:52:
// tokens[:Q].permute(1, 0, 2)) and its adjoint (zero rows for the tokens that are not queries); out = (add ? add : 0) + sum_k parts[k]
namespace {
__global__ __launch_bounds__(256) void first_queries_kernel(const float *__restrict__ src, float *__restrict__ dst, int T, int B, int Q, int E, int adjoint) {
    const int n = adjoint ? T * B * E : B * Q * E;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        if (!adjoint) {
            const int e = idx % E, q = (idx / E) % Q, b = idx / (E * Q);
            dst[idx] = src[((size_t)q * B + b) * E + e];
        } else {
            const int e = idx % E, b = (idx / E) % B, t = idx / (E * B);
            dst[idx] = t < Q ? src[((size_t)b * Q + t) * E + e] : 0.f;
        }
    }
}
__global__ __launch_bounds__(256) void sum_parts_kernel(const float *__restrict__ parts, const float *__restrict__ add, float *__restrict__ out, int nparts,
                                                        int n4) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n4; idx += gridDim.x * 256) {
        float4 a = add ? reinterpret_cast<const float4 *>(add)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 s = reinterpret_cast<const float4 *>(parts)[idx];
        for (int k = 1; k < nparts; ++k) {                      // (torch's sum over dim 0 adds in index order too)
            const float4 v = reinterpret_cast<const float4 *>(parts)[(size_t)k * n4 + idx];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4 *>(out)[idx] = make_float4(s.x + a.x, s.y + a.y, s.z + a.z, s.w + a.w);
    }
}
}  // namespace
extern "C" int sqd_first_queries(const float *src, float *dst, int T, int B, int Q, int E, int adjoint, void *stream) {
    SQD_CHECK_ARG(src && dst && T > 0 && B > 0 && Q > 0 && Q <= T && E > 0, "sqd_first_queries: bad arguments");
    (void)hipGetLastError();
    const int n = adjoint ? T * B * E : B * Q * E;
    hipLaunchKernelGGL(first_queries_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, T, B, Q, E, adjoint);
    SQD_CHECK_LAUNCH("sqd_first_queries");
    return SQD_OK;
}
extern "C" int sqd_sum_parts(const float *parts, const float *add, float *out, int nparts, int64_t n, void *stream) {
    SQD_CHECK_ARG(parts && out && nparts >= 1 && n > 0 && n % 4 == 0 && n / 4 < (1ll << 31), "sqd_sum_parts: bad arguments (n must be a multiple of 4)");
    (void)hipGetLastError();
    const int n4 = (int)(n / 4);
    hipLaunchKernelGGL(sum_parts_kernel, dim3((n4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, parts, add, out, nparts, n4);
    SQD_CHECK_LAUNCH("sqd_sum_parts");
    return SQD_OK;
}

extern "C" int sqd_vit_supported(int E, int F) { return (E >= 4 && E <= 64 && E % 4 == 0 && F >= 4 && F % 4 == 0 && F <= 8192) ? 1 : 0; }

// ---- add + dropout + LayerNorm.  x [rows,E]; y [nparts][rows,E] (summed, + ybias [E] if not NULL); mask [rows,E] bytes
// (1 = keep) or NULL; scale = 1/(1-p)
extern "C" int sqd_addln_fwd(const float *x, const float *y, int nparts, const float *ybias, const unsigned char *mask,
                             const float *gamma, const float *beta, float *out, float *xhat, float *rstd, int rows, int E, float scale,
                             float eps, void *stream) {
    SQD_CHECK_ARG(x && y && gamma && beta && out && xhat && rstd && nparts >= 1, "sqd_addln_fwd: null pointer or nparts < 1");
    if (vit_check("sqd_addln_fwd", rows, E)) return SQD_EINVAL;
    (void)hipGetLastError();
    if (E <= 32)
        hipLaunchKernelGGL(addln_fwd_kernel<32>, dim3((rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, x, y, nparts, ybias, mask, gamma, beta, out,
                           xhat, rstd, rows, E, scale, eps);
    else
        hipLaunchKernelGGL(addln_fwd_kernel<64>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, nparts, ybias, mask, gamma, beta, out,
                           xhat, rstd, rows, E, scale, eps);
    SQD_CHECK_LAUNCH("sqd_addln_fwd");
    return SQD_OK;
}

// g = g_out + sum of nextra tensors g_extra [nextra][rows,E]; part: sqd_addln_nblk(rows) x [2][E] partials of (g_gamma, g_beta),
// to be column-summed (sqd_colsum_multi)
extern "C" int sqd_addln_nblk(int rows) { return (rows + LN_ROWS - 1) / LN_ROWS; }
extern "C" int sqd_addln_bwd(const float *g_out, const float *g_extra, int nextra, const float *xhat, const float *rstd,
                             const unsigned char *mask, const float *gamma, float *g_x, float *g_y, float *part, int rows, int E,
                             float scale, void *stream) {
    SQD_CHECK_ARG(g_out && xhat && rstd && gamma && g_x && g_y && part && nextra >= 0 && (g_extra || nextra == 0),
                  "sqd_addln_bwd: null pointer");
    if (vit_check("sqd_addln_bwd", rows, E)) return SQD_EINVAL;
    (void)hipGetLastError();
    if (E <= 32)
        hipLaunchKernelGGL(addln_bwd_kernel<32>, dim3(sqd_addln_nblk(rows)), dim3(256), 0, (hipStream_t)stream, g_out, g_extra, nextra, xhat, rstd,
                           mask, gamma, g_x, g_y, part, rows, E, scale);
    else
        hipLaunchKernelGGL(addln_bwd_kernel<64>, dim3(sqd_addln_nblk(rows)), dim3(256), 0, (hipStream_t)stream, g_out, g_extra, nextra, xhat, rstd,
                           mask, gamma, g_x, g_y, part, rows, E, scale);
    SQD_CHECK_LAUNCH("sqd_addln_bwd");
    return SQD_OK;
}

// ---- nseg <= 12 column sums in one launch: dst[s][c] = sum_{r < nrows[s]} src[s][r * ncols[s] + c]; ncols % 4 == 0;
// tr[s] > 0: write the [ncols/tr][tr] result transposed
extern "C" int sqd_colsum_multi(const float *const *src, float *const *dst, const int *nrows, const int *ncols, const int *tr, int nseg,
                                void *stream) {
    SQD_CHECK_ARG(src && dst && nrows && ncols && tr && nseg >= 1 && nseg <= CS_MAXSEG, "sqd_colsum_multi: bad arguments (nseg=%d)", nseg);
    ColsumSegs S;
    int blk = 0, max_rows = 0;
    for (int s = 0; s < nseg; ++s) max_rows = nrows[s] > max_rows ? nrows[s] : max_rows;
    // (thousands of partial rows over a few dozen columns — the column sums the BatchNorm backward passes leave for a bias gradient, nnkernels.py —:
    //  2 columns x 128 row groups, 16 sequential loads per thread at 2048 rows; with 8 x 32 one workgroup walked 64 and took 16 us)
    const int CW = max_rows >= 1024 ? 2 : max_rows >= 256 ? 8 : 32;
    for (int s = 0; s < nseg; ++s) {
        SQD_CHECK_ARG(src[s] && dst[s] && nrows[s] >= 1 && ncols[s] >= 4 && ncols[s] % 4 == 0 && tr[s] >= 0 &&
                          (tr[s] == 0 || ncols[s] % tr[s] == 0),
                      "sqd_colsum_multi: segment %d: nrows=%d ncols=%d tr=%d", s, nrows[s], ncols[s], tr[s]);
        S.src[s] = src[s]; S.dst[s] = dst[s]; S.nrows[s] = nrows[s]; S.ncols[s] = ncols[s]; S.tr[s] = tr[s];
        S.blk0[s] = blk;
        blk += (ncols[s] / 4 + CW - 1) / CW;
    }
    S.blk0[nseg] = blk;
    S.nseg = nseg;
    (void)hipGetLastError();
    if (CW == 2) hipLaunchKernelGGL(colsum_multi_kernel<2>, dim3(blk), dim3(256), 0, (hipStream_t)stream, S);
    else if (CW == 8) hipLaunchKernelGGL(colsum_multi_kernel<8>, dim3(blk), dim3(256), 0, (hipStream_t)stream, S);
    else hipLaunchKernelGGL(colsum_multi_kernel<32>, dim3(blk), dim3(256), 0, (hipStream_t)stream, S);
    SQD_CHECK_LAUNCH("sqd_colsum_multi");
    return SQD_OK;
}

// ---- feed-forward.  x [rows,E], W1 [F,E], b1 [F], W2 [E,F]; mask [rows,F] bytes (4-byte aligned) or NULL
//      -> ypart [G][rows,E] with G = sqd_ffn_groups(F): y = sum_g ypart[g] + b2 (added by the consumer, sqd_addln_fwd)
extern "C" int sqd_ffn_groups(int F) { return (F + 127) / 128; }
extern "C" int sqd_ffn_fwd(const float *x, const float *W1, const float *b1, const float *W2, const unsigned char *mask, float *ypart,
                           int rows, int E, int F, float scale, void *stream) {
    SQD_CHECK_ARG(x && W1 && b1 && W2 && ypart, "sqd_ffn_fwd: null pointer");
    SQD_CHECK_ARG(rows > 0 && sqd_vit_supported(E, F), "sqd_ffn_fwd: unsupported dims rows=%d E=%d F=%d", rows, E, F);
    SQD_CHECK_ARG(((uintptr_t)mask & 3) == 0, "sqd_ffn_fwd: mask must be 4-byte aligned");
    (void)hipGetLastError();
    if (E <= 32)
        hipLaunchKernelGGL(ffn_fwd_kernel<1>, dim3((rows + 31) / 32, sqd_ffn_groups(F)), dim3(256), 0, (hipStream_t)stream, x, W1, b1, W2, mask, ypart,
                           rows, E, F, scale);
    else
        hipLaunchKernelGGL(ffn_fwd_kernel<2>, dim3((rows + 31) / 32, sqd_ffn_groups(F)), dim3(256), 0, (hipStream_t)stream, x, W1, b1, W2, mask, ypart,
                           rows, E, F, scale);
    SQD_CHECK_LAUNCH("sqd_ffn_fwd");
    return SQD_OK;
}

// backward: g_y -> gxpart [G][rows,E] (g_x = sum over G) and per-token-tile partials (T = sqd_ffn_tiles(rows)):
// pW1 [T][F,E], pW2T [T][F,E] (g_W2 transposed), pb1 [T][F], pb2 [T][E]; column-sum them with sqd_colsum_multi (tr = E for pW2T)
extern "C" int sqd_ffn_tiles(int rows) { return (rows + 31) / 32; }
extern "C" int sqd_ffn_bwd(const float *x, const float *g_y, const float *W1, const float *b1, const float *W2,
                           const unsigned char *mask, float *gxpart, float *pW1, float *pb1, float *pW2T, float *pb2, int rows, int E,
                           int F, float scale, void *stream) {
    SQD_CHECK_ARG(x && g_y && W1 && b1 && W2 && gxpart && pW1 && pb1 && pW2T && pb2, "sqd_ffn_bwd: null pointer");
    SQD_CHECK_ARG(rows > 0 && sqd_vit_supported(E, F), "sqd_ffn_bwd: unsupported dims rows=%d E=%d F=%d", rows, E, F);
    SQD_CHECK_ARG(((uintptr_t)mask & 3) == 0, "sqd_ffn_bwd: mask must be 4-byte aligned");
    (void)hipGetLastError();
    if (E <= 32)
        hipLaunchKernelGGL(ffn_bwd_kernel<1>, dim3(sqd_ffn_tiles(rows), sqd_ffn_groups(F)), dim3(256), 0, (hipStream_t)stream, x, g_y, W1, b1, W2, mask,
                           gxpart, pW1, pb1, pW2T, pb2, rows, E, F, scale);
    else
        hipLaunchKernelGGL(ffn_bwd_kernel<2>, dim3(sqd_ffn_tiles(rows), sqd_ffn_groups(F)), dim3(256), 0, (hipStream_t)stream, x, g_y, W1, b1, W2, mask,
                           gxpart, pW1, pb1, pW2T, pb2, rows, E, F, scale);
    SQD_CHECK_LAUNCH("sqd_ffn_bwd");
    return SQD_OK;
}

// ===================================================================================================
// multi-head self-attention of the encoder layer (nn.MultiheadAttention, packed in_proj, batch_first = False) for short token
// sequences: S <= 512 tokens, head dimension HD in {4, 8} (E = 64: HD = 16, E = 56: HD = 14, both S <= 256).  One workgroup per (batch element, head), four (two beyond 256 tokens) threads per token
// (each takes every 4th key / query and a quarter of the features; quad shuffles combine them); the whole head (projections,
// S x S scores, softmax, attention dropout, P.V, its slice of the out-projection) is VALU work on operands read from LDS — 120 x 120 x 8 per head is far too small for the matrix cores to matter; what
// counts is that it is ONE launch with no intermediate tensors instead of ~8 (forward) / ~20 (backward).
// The out-projection is left as per-head partials ypart [H][rows,E] (+ bias added by the consumer, sqd_addln_fwd), the
// input gradient as per-head partials gxpart [H][rows,E]; weight gradients as per-batch-element partials for sqd_colsum_multi.
// Token (s, b) is row s*B + b.  Keep-mask of the attention dropout: bytes [B][H][S][SP], SP = S rounded up to 4.
// ===================================================================================================
namespace {
// Workgroup shapes: T tokens x P threads per token (thread P*t + p takes every P-th key / query, 1/P of the projection outputs
// and of the feature columns; shuffles inside the group of P lanes combine them).  S <= 128: 128 x 4; S <= 256: 256 x 4;
// S <= 512: 512 x 2 (the 1024-thread limit) — and there the backward re-reads x / g_sa from global memory (L2) for the
// weight-gradient sums instead of staging them in LDS (2 x 67 KB would not fit next to Q, K, V, dO).
template <int P>
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    if (P == 4) v += __shfl_xor(v, 2, 64);
    return v;
}
template <int P>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    if (P == 4) v = fmaxf(v, __shfl_xor(v, 2, 64));
    return v;
}

// head h's slices of the packed in-projection (rows q | k | v) and of the out-projection, staged in LDS
template <int HD, int EE>
struct MhaWeights {
    float win[3 * HD][EE + 4];                           // + 4: the 4 parts read rows 2p.. at distinct banks
    float bin[3 * HD];
    float wo[EE][HD];                                    // wo[e][d] = Wo[e][h*HD + d]
};
template <int HD, int EE, int NTH>
__device__ __forceinline__ void mha_stage_weights(MhaWeights<HD, EE> &W, const float *__restrict__ Win, const float *__restrict__ bin,
                                                  const float *__restrict__ Wo, int h) {
    for (int idx = threadIdx.x; idx < 3 * HD * EE; idx += NTH) {
        const int j = idx / EE, e = idx % EE;
        W.win[j][e] = Win[(size_t)((j / HD) * EE + h * HD + j % HD) * EE + e];
    }
    for (int idx = threadIdx.x; idx < EE * HD; idx += NTH) W.wo[idx / HD][idx % HD] = Wo[(size_t)(idx / HD) * EE + h * HD + idx % HD];
    if (threadIdx.x < 3 * HD) W.bin[threadIdx.x] = bin[(threadIdx.x / HD) * EE + h * HD + threadIdx.x % HD];
}

// part p of token t projects HD/4 of the query / key / value features and writes them to LDS
template <int HD, int EE, int P>
__device__ __forceinline__ void mha_project(const MhaWeights<HD, EE> &W, const float (&xr)[EE], int t, int p, float qscale,
                                            float (*Qs)[HD], float (*Ks)[HD], float (*Vs)[HD]) {
    constexpr int JP = HD / P;
#pragma unroll
    for (int jj = 0; jj < JP; ++jj) {
        const int j = p * JP + jj;
        float aq = W.bin[j], ak = W.bin[HD + j], av = W.bin[2 * HD + j];
#pragma unroll
        for (int e = 0; e < EE; e += 4) {
            const float4 wq = *reinterpret_cast<const float4 *>(&W.win[j][e]);
            const float4 wk = *reinterpret_cast<const float4 *>(&W.win[HD + j][e]);
            const float4 wv = *reinterpret_cast<const float4 *>(&W.win[2 * HD + j][e]);
            aq = fmaf(xr[e], wq.x, fmaf(xr[e + 1], wq.y, fmaf(xr[e + 2], wq.z, fmaf(xr[e + 3], wq.w, aq))));
            ak = fmaf(xr[e], wk.x, fmaf(xr[e + 1], wk.y, fmaf(xr[e + 2], wk.z, fmaf(xr[e + 3], wk.w, ak))));
            av = fmaf(xr[e], wv.x, fmaf(xr[e + 1], wv.y, fmaf(xr[e + 2], wv.z, fmaf(xr[e + 3], wv.w, av))));
        }
        Qs[t][j] = aq * qscale; Ks[t][j] = ak; Vs[t][j] = av;
    }
}

template <int HD, int EE, int T, int P>
__global__ __launch_bounds__(T * P) void mha_fwd_kernel(const float *__restrict__ x, const float *__restrict__ Win,
                                                              const float *__restrict__ bin, const float *__restrict__ Wo,
                                                              const unsigned char *__restrict__ mask, float *__restrict__ ypart,
                                                              float *__restrict__ o_save, float *__restrict__ ml_save, int S, int B,
                                                              int H, float qscale, float dscale) {
    __shared__ MhaWeights<HD, EE> W;
    __shared__ float Qs[T][HD], Ks[T][HD], Vs[T][HD];
    const int b = blockIdx.x, h = blockIdx.y, t = threadIdx.x / P, p = threadIdx.x % P;
    const bool tv = t < S;
    const int SP = (S + 3) & ~3;
    const size_t row = (size_t)t * B + b;
    mha_stage_weights<HD, EE, T * P>(W, Win, bin, Wo, h);
    float xr[EE];
#pragma unroll
    for (int e = 0; e < EE; e += 4) {
        const float4 v4 = tv ? *reinterpret_cast<const float4 *>(x + row * EE + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        xr[e] = v4.x; xr[e + 1] = v4.y; xr[e + 2] = v4.z; xr[e + 3] = v4.w;
    }
    __syncthreads();
    mha_project<HD, EE, P>(W, xr, t, p, qscale, Qs, Ks, Vs);
    __syncthreads();
    float q[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = Qs[t][d];
    float m = -INFINITY;
    for (int kk = p; kk < S; kk += P) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(q[d], Ks[kk][d], s);
        m = fmaxf(m, s);
    }
    m = group_max<P>(m);                                     // S >= 1: at least part 0 saw a key
    float l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    const unsigned char *mrow = mask ? mask + (((size_t)b * H + h) * S + (tv ? t : 0)) * SP : nullptr;
    for (int kk = p; kk < S; kk += P) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(q[d], Ks[kk][d], s);
        const float pr = __expf(s - m);
        l += pr;
        const float pd = (!mrow || mrow[kk]) ? pr * dscale : 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = fmaf(pd, Vs[kk][d], o[d]);
    }
    l = group_sum<P>(l);
    const float rl = 1.f / l;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = group_sum<P>(o[d]) * rl;
    if (!tv) return;
    const size_t hs = ((size_t)b * H + h) * S + t;
    if (p == 0) {
#pragma unroll
        for (int d = 0; d < HD; ++d) o_save[hs * HD + d] = o[d];
        ml_save[hs * 2] = m;
        ml_save[hs * 2 + 1] = l;
    }
    // this head's part of the out-projection; part p writes features [p*EE/4, (p+1)*EE/4)
    constexpr int EP = EE / P;
    float *yo = ypart + ((size_t)h * S * B + row) * EE + p * EP;
#pragma unroll
    for (int e = 0; e < EP; e += 4) {
        float y4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) a = fmaf(o[d], W.wo[p * EP + e + c][d], a);
            y4[c] = a;
        }
        *reinterpret_cast<float4 *>(yo + e) = make_float4(y4[0], y4[1], y4[2], y4[3]);
    }
}

template <int HD, int EE, int T, int P, bool STAGE>
__global__ __launch_bounds__(T * P) void mha_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gsa,
                                                              const float *__restrict__ Win, const float *__restrict__ bin,
                                                              const float *__restrict__ Wo, const unsigned char *__restrict__ mask,
                                                              const float *__restrict__ o_save, const float *__restrict__ ml_save,
                                                              float *__restrict__ gxpart, float *__restrict__ pWin,
                                                              float *__restrict__ pbin, float *__restrict__ pWo, float *__restrict__ pbo,
                                                              int S, int B, int H, float qscale, float dscale) {
    __shared__ MhaWeights<HD, EE> W;
    __shared__ float xs[STAGE ? T : 1][EE + 1], gs[STAGE ? T : 1][EE + 1];
    __shared__ float qkvd[4][T][HD];                      // Q (scaled), K, V, dO; later aliased by dqkv [T][3*HD]
    __shared__ float os[T][HD];
    __shared__ float st[3][T];                            // row max, 1 / row sum, D = dO . o
    constexpr int JP = HD / P, EP = EE / P, NTH = T * P;
    const int b = blockIdx.x, h = blockIdx.y, t = threadIdx.x / P, p = threadIdx.x % P;
    const bool tv = t < S;
    const int SP = (S + 3) & ~3;
    const size_t row = (size_t)t * B + b, hs = ((size_t)b * H + h) * S + t;
    mha_stage_weights<HD, EE, NTH>(W, Win, bin, Wo, h);
    float xr[EE], gr[EE];
#pragma unroll
    for (int e = 0; e < EE; e += 4) {
        const float4 a = tv ? *reinterpret_cast<const float4 *>(x + row * EE + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 g = tv ? *reinterpret_cast<const float4 *>(gsa + row * EE + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        xr[e] = a.x; xr[e + 1] = a.y; xr[e + 2] = a.z; xr[e + 3] = a.w;
        gr[e] = g.x; gr[e + 1] = g.y; gr[e + 2] = g.z; gr[e + 3] = g.w;
    }
    if (STAGE) {
#pragma unroll
        for (int e = 0; e < EP; ++e) { xs[t][p * EP + e] = xr[p * EP + e]; gs[t][p * EP + e] = gr[p * EP + e]; }
    }
    __syncthreads();
    mha_project<HD, EE, P>(W, xr, t, p, qscale, qkvd[0], qkvd[1], qkvd[2]);
#pragma unroll
    for (int jj = 0; jj < JP; ++jj) {
        const int d = p * JP + jj;
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < EE; ++e) a = fmaf(gr[e], W.wo[e][d], a);
        qkvd[3][t][d] = a;                                // dO = g_sa . Wo[:, head]
        os[t][d] = tv ? o_save[hs * HD + d] : 0.f;
    }
    __syncthreads();
    float q[HD], k[HD], v[HD], dO[HD];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        q[d] = qkvd[0][t][d]; k[d] = qkvd[1][t][d]; v[d] = qkvd[2][t][d]; dO[d] = qkvd[3][t][d];
        D = fmaf(dO[d], os[t][d], D);
    }
    const float m = tv ? ml_save[hs * 2] : 0.f, rl = tv ? 1.f / ml_save[hs * 2 + 1] : 0.f;
    if (p == 0) { st[0][t] = m; st[1][t] = rl; st[2][t] = D; }
    __syncthreads();
    const unsigned char *mbase = mask ? mask + ((size_t)b * H + h) * S * SP : nullptr;
    float dq[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = dk[d] = dv[d] = 0.f;
    // pass A: thread = (query t, every 4th key) -> dq' (gradient of the scaled query)
    {
        const unsigned char *mrow = mbase ? mbase + (size_t)(tv ? t : 0) * SP : nullptr;
        for (int kk = p; kk < S; kk += P) {
            float s = 0.f, dpd = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                s = fmaf(q[d], qkvd[1][kk][d], s);
                dpd = fmaf(dO[d], qkvd[2][kk][d], dpd);
            }
            const float pr = __expf(s - m) * rl;
            const float dp = (!mrow || mrow[kk]) ? dpd * dscale : 0.f;
            const float ds = pr * (dp - D);
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, qkvd[1][kk][d], dq[d]);
        }
    }
    // pass B: thread = (key t, every 4th query) -> dk, dv
    for (int qq = p; qq < S; qq += P) {
        float s = 0.f, dpd = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            s = fmaf(qkvd[0][qq][d], k[d], s);
            dpd = fmaf(qkvd[3][qq][d], v[d], dpd);
        }
        const float pr = __expf(s - st[0][qq]) * st[1][qq];
        const bool keep = tv && (!mbase || mbase[(size_t)qq * SP + t] != 0);
        const float pd = keep ? pr * dscale : 0.f, dp = keep ? dpd * dscale : 0.f;
        const float ds = pr * (dp - st[2][qq]);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            dv[d] = fmaf(pd, qkvd[3][qq][d], dv[d]);
            dk[d] = fmaf(ds, qkvd[0][qq][d], dk[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        dq[d] = tv ? group_sum<P>(dq[d]) * qscale : 0.f;      // gradient of the unscaled projection
        dk[d] = tv ? group_sum<P>(dk[d]) : 0.f;
        dv[d] = tv ? group_sum<P>(dv[d]) : 0.f;
    }
    __syncthreads();                                      // everyone is done with Q, K, V, dO: reuse the area for dqkv
    float(*dqkv)[3 * HD] = reinterpret_cast<float(*)[3 * HD]>(&qkvd[0][0][0]);
    if (p == 0) {
#pragma unroll
        for (int d = 0; d < HD; ++d) { dqkv[t][d] = dq[d]; dqkv[t][HD + d] = dk[d]; dqkv[t][2 * HD + d] = dv[d]; }
    }
    // this head's part of the input gradient; part p writes features [p*EP, (p+1)*EP)
    if (tv) {
        float *go = gxpart + ((size_t)h * S * B + row) * EE + p * EP;
#pragma unroll
        for (int e = 0; e < EP; e += 4) {
            float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < HD; ++j) {
                const float4 wq = *reinterpret_cast<const float4 *>(&W.win[j][p * EP + e]);
                const float4 wk = *reinterpret_cast<const float4 *>(&W.win[HD + j][p * EP + e]);
                const float4 wv = *reinterpret_cast<const float4 *>(&W.win[2 * HD + j][p * EP + e]);
                a[0] = fmaf(dq[j], wq.x, fmaf(dk[j], wk.x, fmaf(dv[j], wv.x, a[0])));
                a[1] = fmaf(dq[j], wq.y, fmaf(dk[j], wk.y, fmaf(dv[j], wv.y, a[1])));
                a[2] = fmaf(dq[j], wq.z, fmaf(dk[j], wk.z, fmaf(dv[j], wv.z, a[2])));
                a[3] = fmaf(dq[j], wq.w, fmaf(dk[j], wk.w, fmaf(dv[j], wv.w, a[3])));
            }
            *reinterpret_cast<float4 *>(go + e) = make_float4(a[0], a[1], a[2], a[3]);
        }
    }
    __syncthreads();
    // weight-gradient partials of this batch element (rows / columns of head h): one output per thread, sums over the tokens
    for (int idx = threadIdx.x; idx < 3 * HD * EE; idx += NTH) {
        const int j = idx / EE, e = idx % EE;             // j: 0..HD-1 query rows, HD.. key rows, 2HD.. value rows
        float a = 0.f;
        if (STAGE) {
#pragma unroll 4
            for (int s = 0; s < S; ++s) a = fmaf(dqkv[s][j], xs[s][e], a);
        } else {
#pragma unroll 8
            for (int s = 0; s < S; ++s) a = fmaf(dqkv[s][j], x[((size_t)s * B + b) * EE + e], a);     // 128-byte rows, L2-resident
        }
        pWin[((size_t)b * 3 * EE + (j / HD) * EE + h * HD + j % HD) * EE + e] = a;
    }
    for (int idx = threadIdx.x; idx < EE * HD; idx += NTH) {
        const int e = idx / HD, d = idx % HD;
        float a = 0.f;
        if (STAGE) {
#pragma unroll 4
            for (int s = 0; s < S; ++s) a = fmaf(gs[s][e], os[s][d], a);
        } else {
#pragma unroll 8
            for (int s = 0; s < S; ++s) a = fmaf(gsa[((size_t)s * B + b) * EE + e], os[s][d], a);
        }
        pWo[((size_t)b * EE + e) * EE + h * HD + d] = a;
    }
    if (threadIdx.x < 3 * HD) {
        const int j = threadIdx.x;
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += dqkv[s][j];
        pbin[(size_t)b * 3 * EE + (j / HD) * EE + h * HD + j % HD] = a;
    }
    if (h == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + EE) {
        const int e = threadIdx.x - 64;
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += STAGE ? gs[s][e] : gsa[((size_t)s * B + b) * EE + e];
        pbo[(size_t)b * EE + e] = a;
    }
}

// model_dim 64 (head dimension 16): 128 tokens x 4 threads, or 256 tokens x 2 threads (512 threads: the 64 + 64 row registers of the
// backward need the 256-register budget); the backward stages x / g_sa in LDS only for the 128-token shape
void mha_launch_fwd64(dim3 grid, hipStream_t st, const float *x, const float *Win, const float *bin, const float *Wo, const unsigned char *mask,
                      float *ypart, float *o_save, float *ml_save, int S, int B, int H, float qscale, float dscale) {
    if (S <= 128)
        hipLaunchKernelGGL((mha_fwd_kernel<16, 64, 128, 4>), grid, dim3(512), 0, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else
        hipLaunchKernelGGL((mha_fwd_kernel<16, 64, 256, 2>), grid, dim3(512), 0, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
}
void mha_launch_bwd64(dim3 grid, hipStream_t st, const float *x, const float *gsa, const float *Win, const float *bin, const float *Wo,
                      const unsigned char *mask, const float *o_save, const float *ml_save, float *gxpart, float *pWin, float *pbin, float *pWo,
                      float *pbo, int S, int B, int H, float qscale, float dscale) {
    if (S <= 128)
        hipLaunchKernelGGL((mha_bwd_kernel<16, 64, 128, 4, true>), grid, dim3(512), 0, st, x, gsa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin,
                           pbin, pWo, pbo, S, B, H, qscale, dscale);
    else
        hipLaunchKernelGGL((mha_bwd_kernel<16, 64, 256, 2, false>), grid, dim3(512), 0, st, x, gsa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin,
                           pbin, pWo, pbo, S, B, H, qscale, dscale);
}
// model_dim 56 with its 4 heads of 14 (reference args_files/args_cityscapes_train.txt:9): two threads per token (14 = 2 x 7 projection
// outputs, 56 = 2 x 28 feature columns per part), up to 256 tokens
void mha_launch_fwd56(dim3 grid, hipStream_t st, const float *x, const float *Win, const float *bin, const float *Wo, const unsigned char *mask,
                      float *ypart, float *o_save, float *ml_save, int S, int B, int H, float qscale, float dscale) {
    if (S <= 128)
        hipLaunchKernelGGL((mha_fwd_kernel<14, 56, 128, 2>), grid, dim3(256), 0, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else
        hipLaunchKernelGGL((mha_fwd_kernel<14, 56, 256, 2>), grid, dim3(512), 0, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
}
void mha_launch_bwd56(dim3 grid, hipStream_t st, const float *x, const float *gsa, const float *Win, const float *bin, const float *Wo,
                      const unsigned char *mask, const float *o_save, const float *ml_save, float *gxpart, float *pWin, float *pbin, float *pWo,
                      float *pbo, int S, int B, int H, float qscale, float dscale) {
    if (S <= 128)
        hipLaunchKernelGGL((mha_bwd_kernel<14, 56, 128, 2, true>), grid, dim3(256), 0, st, x, gsa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin,
                           pbin, pWo, pbo, S, B, H, qscale, dscale);
    else
        hipLaunchKernelGGL((mha_bwd_kernel<14, 56, 256, 2, false>), grid, dim3(512), 0, st, x, gsa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin,
                           pbin, pWo, pbo, S, B, H, qscale, dscale);
}
template <int HD, int EE>
void mha_launch_fwd(dim3 grid, hipStream_t st, const float *x, const float *Win, const float *bin, const float *Wo, const unsigned char *mask,
                    float *ypart, float *o_save, float *ml_save, int S, int B, int H, float qscale, float dscale) {
    if (S <= 128)
        hipLaunchKernelGGL((mha_fwd_kernel<HD, EE, 128, 4>), grid, dim3(512), 0, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else if (S <= 256)
        hipLaunchKernelGGL((mha_fwd_kernel<HD, EE, 256, 4>), grid, dim3(1024), 0, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else
        hipLaunchKernelGGL((mha_fwd_kernel<HD, EE, 512, 2>), grid, dim3(1024), 0, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
}
template <int HD, int EE>
void mha_launch_bwd(dim3 grid, hipStream_t st, const float *x, const float *gsa, const float *Win, const float *bin, const float *Wo,
                    const unsigned char *mask, const float *o_save, const float *ml_save, float *gxpart, float *pWin, float *pbin, float *pWo,
                    float *pbo, int S, int B, int H, float qscale, float dscale) {
    if (S <= 128)
        hipLaunchKernelGGL((mha_bwd_kernel<HD, EE, 128, 4, true>), grid, dim3(512), 0, st, x, gsa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin,
                           pbin, pWo, pbo, S, B, H, qscale, dscale);
    else if (S <= 256)
        hipLaunchKernelGGL((mha_bwd_kernel<HD, EE, 256, 4, true>), grid, dim3(1024), 0, st, x, gsa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin,
                           pbin, pWo, pbo, S, B, H, qscale, dscale);
    else
        hipLaunchKernelGGL((mha_bwd_kernel<HD, EE, 512, 2, false>), grid, dim3(1024), 0, st, x, gsa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin,
                           pbin, pWo, pbo, S, B, H, qscale, dscale);
}
}  // namespace

extern "C" int sqd_mha_supported(int S, int E, int H) {
    if (!(E == 16 || E == 32 || E == 56 || E == 64) || H < 1 || E % H) return 0;
    const int hd = E / H;
    if (E == 64) return (S >= 1 && S <= 256 && hd == 16) ? 1 : 0;          // model_dim 64 with its 4 heads
    if (E == 56) return (S >= 1 && S <= 256 && hd == 14) ? 1 : 0;          // model_dim 56 with its 4 heads (the Cityscapes args files)
    return (S >= 1 && S <= 512 && (hd == 4 || hd == 8)) ? 1 : 0;
}

// x [S*B, E] (token (s,b) = row s*B+b), Win [3E,E], bin [3E], Wo [E,E]; mask [B][H][S][SP] keep bytes (SP = S rounded up to 4)
// or NULL, dscale = 1/(1-p) -> ypart [H][S*B,E] (attention output through the out-projection, per head, without its bias),
// o_save [B][H][S][E/H], ml_save [B][H][S][2] (row max, row sum) for the backward
extern "C" int sqd_mha_fwd(const float *x, const float *Win, const float *bin, const float *Wo, const unsigned char *mask, float *ypart,
                           float *o_save, float *ml_save, int S, int B, int E, int H, float dscale, void *stream) {
    SQD_CHECK_ARG(x && Win && bin && Wo && ypart && o_save && ml_save, "sqd_mha_fwd: null pointer");
    SQD_CHECK_ARG(B >= 1 && sqd_mha_supported(S, E, H), "sqd_mha_fwd: unsupported dims S=%d B=%d E=%d heads=%d", S, B, E, H);
    SQD_CHECK_ARG(((uintptr_t)mask & 3) == 0, "sqd_mha_fwd: mask must be 4-byte aligned");
    const int hd = E / H;
    const float qscale = 1.f / sqrtf((float)hd);
    const dim3 grid(B, H);
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    if (E == 64) mha_launch_fwd64(grid, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else if (E == 56) mha_launch_fwd56(grid, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else if (hd == 8 && E == 32) mha_launch_fwd<8, 32>(grid, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else if (hd == 4 && E == 32) mha_launch_fwd<4, 32>(grid, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else if (hd == 8 && E == 16) mha_launch_fwd<8, 16>(grid, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    else mha_launch_fwd<4, 16>(grid, st, x, Win, bin, Wo, mask, ypart, o_save, ml_save, S, B, H, qscale, dscale);
    SQD_CHECK_LAUNCH("sqd_mha_fwd");
    return SQD_OK;
}

// g_sa [S*B,E] (gradient of the attention block's output, after the out-projection) -> gxpart [H][S*B,E] (g_x = sum over heads)
// and per-batch-element partials pWin [B][3E,E], pbin [B][3E], pWo [B][E,E], pbo [B][E] for sqd_colsum_multi
extern "C" int sqd_mha_bwd(const float *x, const float *g_sa, const float *Win, const float *bin, const float *Wo,
                           const unsigned char *mask, const float *o_save, const float *ml_save, float *gxpart, float *pWin, float *pbin,
                           float *pWo, float *pbo, int S, int B, int E, int H, float dscale, void *stream) {
    SQD_CHECK_ARG(x && g_sa && Win && bin && Wo && o_save && ml_save && gxpart && pWin && pbin && pWo && pbo, "sqd_mha_bwd: null pointer");
    SQD_CHECK_ARG(B >= 1 && sqd_mha_supported(S, E, H), "sqd_mha_bwd: unsupported dims S=%d B=%d E=%d heads=%d", S, B, E, H);
    SQD_CHECK_ARG(((uintptr_t)mask & 3) == 0, "sqd_mha_bwd: mask must be 4-byte aligned");
    const int hd = E / H;
    const float qscale = 1.f / sqrtf((float)hd);
    const dim3 grid(B, H);
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
#define SQD_MHA_BWD(HD_, EE_) \
    mha_launch_bwd<HD_, EE_>(grid, st, x, g_sa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin, pbin, pWo, pbo, S, B, H, qscale, dscale)
    if (E == 64) mha_launch_bwd64(grid, st, x, g_sa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin, pbin, pWo, pbo, S, B, H, qscale, dscale);
    else if (E == 56) mha_launch_bwd56(grid, st, x, g_sa, Win, bin, Wo, mask, o_save, ml_save, gxpart, pWin, pbin, pWo, pbo, S, B, H, qscale, dscale);
    else if (hd == 8 && E == 32) SQD_MHA_BWD(8, 32);
    else if (hd == 4 && E == 32) SQD_MHA_BWD(4, 32);
    else if (hd == 8 && E == 16) SQD_MHA_BWD(8, 16);
    else SQD_MHA_BWD(4, 16);
#undef SQD_MHA_BWD
    SQD_CHECK_LAUNCH("sqd_mha_bwd");
    return SQD_OK;
}
