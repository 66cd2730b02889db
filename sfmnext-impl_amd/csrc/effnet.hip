// effnet.hip — the operators of the EfficientNet-b5 trunk (reference networks/base_encoder.py:76-107: torch.hub
// 'rwightman/gen-efficientnet-pytorch' tf_efficientnet_b5_ap — third-party, restated from the public architecture) that the
// ResNet path does not have: depthwise k x k convolution with TensorFlow "SAME" padding, and the squeeze-and-excite gate.
// Channels-last activations ([N,H,W,C] memory), fp32.  Everything here is HBM / latency bound (a depthwise convolution does
// 2 k^2 flop per element moved): one thread = one pixel x four channels, float4 accesses, lanes along the channel axis.
// Reductions (weight gradients, pooled means) are two-level with fixed order: deterministic.
#include "sqd_common.h"

namespace {
using namespace sqd;

struct DwGeom {
    int N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo;
};
constexpr int DW_MAXK = 7;

// ---------------------------------------------------------------------------------------------------
// depthwise convolution.  w: [k*k][C] (tap-major: sqd_dw_weight_to_taps), y[n,ho,wo,c] = sum_{r,s} x[n,ho*st+r-pt,wo*st+s-pl,c] w[r*k+s][c]
// MODE 0: forward;  MODE 1: data gradient (in = dy, out = dx: gather over the taps whose stride phase matches)
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void dw_conv_kernel(const float *__restrict__ in, const float *__restrict__ w, float *__restrict__ out,
                                                      DwGeom g) {
    const int V = g.C / 4;
    const size_t nout = MODE == 0 ? (size_t)g.N * g.Ho * g.Wo * V : (size_t)g.N * g.H * g.W * V;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % V);
        const size_t p = i / V;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            const int wo = (int)(p % g.Wo), ho = (int)((p / g.Wo) % g.Ho), n = (int)(p / ((size_t)g.Wo * g.Ho));
            const int h0 = ho * g.stride - g.pad_t, w0 = wo * g.stride - g.pad_l;
            for (int r = 0; r < g.k; ++r) {
                const int hi = h0 + r;
                if ((unsigned)hi >= (unsigned)g.H) continue;
                for (int s = 0; s < g.k; ++s) {
                    const int wi = w0 + s;
                    if ((unsigned)wi >= (unsigned)g.W) continue;
                    const float4 xv = *reinterpret_cast<const float4 *>(in + (((size_t)n * g.H + hi) * g.W + wi) * g.C + cg * 4);
                    const float4 wv = *reinterpret_cast<const float4 *>(w + (size_t)(r * g.k + s) * g.C + cg * 4);
                    acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
                    acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
                }
            }
        } else {
            const int wi = (int)(p % g.W), hi = (int)((p / g.W) % g.H), n = (int)(p / ((size_t)g.W * g.H));
            for (int r = 0; r < g.k; ++r) {
                const int hn = hi + g.pad_t - r;
                if (hn < 0 || hn % g.stride) continue;
                const int ho = hn / g.stride;
                if (ho >= g.Ho) continue;
                for (int s = 0; s < g.k; ++s) {
                    const int wn = wi + g.pad_l - s;
                    if (wn < 0 || wn % g.stride) continue;
                    const int wo = wn / g.stride;
                    if (wo >= g.Wo) continue;
                    const float4 gv = *reinterpret_cast<const float4 *>(in + (((size_t)n * g.Ho + ho) * g.Wo + wo) * g.C + cg * 4);
                    const float4 wv = *reinterpret_cast<const float4 *>(w + (size_t)(r * g.k + s) * g.C + cg * 4);
                    acc.x = fmaf(gv.x, wv.x, acc.x); acc.y = fmaf(gv.y, wv.y, acc.y);
                    acc.z = fmaf(gv.z, wv.z, acc.z); acc.w = fmaf(gv.w, wv.w, acc.w);
                }
            }
        }
        *reinterpret_cast<float4 *>(out + i * 4) = acc;
    }
}

// The same arithmetic with a thread on a RUN of four consecutive output columns (four channels each): per filter row it loads the
// 3 * stride + k input columns the run touches once and uses each for up to k taps — 70 instead of 196 float4 loads per four outputs
// of a 7x7 filter (the one-pixel kernel above is bound by its L1 loads: 69 us per ConvNeXt-L block where the tensors move in 25).
// DGRAD (stride 1 only): in = dy, the taps run backwards over the columns (dx[h][w] = sum dy[h + pt - r][w + pl - s] w[r][s]); the order
// of the k^2 additions of an output is that of the one-pixel kernel in both modes: the same bits.
template <int K, int ST, bool DGRAD>
__global__ __launch_bounds__(256) void dw_conv_run_kernel(const float *__restrict__ in, const float *__restrict__ w, float *__restrict__ out,
                                                          DwGeom g, const float *__restrict__ addend = nullptr) {
    constexpr int P = 4, NC = (P - 1) * ST + K;
    const int V = g.C / 4;
    const int OH = DGRAD ? g.H : g.Ho, OW = DGRAD ? g.W : g.Wo, IH = DGRAD ? g.Ho : g.H, IW = DGRAD ? g.Wo : g.W;
    const int wruns = (OW + P - 1) / P;
    const size_t total = (size_t)g.N * OH * wruns * V;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % V);
        const size_t q = i / V;
        const int wr = (int)(q % wruns), oh = (int)((q / wruns) % OH), n = (int)(q / ((size_t)wruns * OH));
        const int ow0 = wr * P;
        const int c0 = DGRAD ? ow0 + g.pad_l - (K - 1) : ow0 * ST - g.pad_l;          // first input column of the run
        float4 acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = zero;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int ih = DGRAD ? oh + g.pad_t - r : oh * ST - g.pad_t + r;
            if ((unsigned)ih >= (unsigned)IH) continue;
            const float *row = in + ((size_t)n * IH + ih) * IW * g.C + cg * 4;
            float4 xv[NC], wv[K];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int iw = c0 + c;
                xv[c] = (unsigned)iw < (unsigned)IW ? *reinterpret_cast<const float4 *>(row + (size_t)iw * g.C) : zero;
            }
#pragma unroll
            for (int sx = 0; sx < K; ++sx) wv[sx] = *reinterpret_cast<const float4 *>(w + (size_t)(r * K + sx) * g.C + cg * 4);
#pragma unroll
            for (int sx = 0; sx < K; ++sx)
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float4 xq = xv[DGRAD ? p + (K - 1 - sx) : p * ST + sx];
                    acc[p].x = fmaf(xq.x, wv[sx].x, acc[p].x); acc[p].y = fmaf(xq.y, wv[sx].y, acc[p].y);
                    acc[p].z = fmaf(xq.z, wv[sx].z, acc[p].z); acc[p].w = fmaf(xq.w, wv[sx].w, acc[p].w);
                }
        }
#pragma unroll
        for (int p = 0; p < P; ++p)
            if (ow0 + p < OW) {
                const size_t o = (((size_t)n * OH + oh) * OW + ow0 + p) * g.C + cg * 4;
                if (DGRAD && addend) {          // the gradient that reaches the input over its other path (the block's shortcut)
                    const float4 u = *reinterpret_cast<const float4 *>(addend + o);
                    acc[p].x += u.x; acc[p].y += u.y; acc[p].z += u.z; acc[p].w += u.w;
                }
                *reinterpret_cast<float4 *>(out + o) = acc[p];
            }
    }
}

// weight gradient: block (chunk, channel band of 64) sums dy * x over its share of the output for every tap; part [nchunk][k*k][C].
// A thread walks RUNS of four consecutive output columns: per filter row it loads the 3*stride + k input columns the run touches
// once and uses each for up to k taps (a tap-by-tap gather re-reads every input element k^2 times through L1, which — not the
// arithmetic — bounds this kernel: 16.7 ms of a 75 ms EfficientNet-b5 step before, profiles/r02b).
// RS (row split, the 5x5 and 7x7 filters): a workgroup takes ONE filter row (blockIdx.z) — K accumulators per thread instead of K^2
// (the 49 float4 accumulators of a 7x7 filter are 196 registers: one wave per SIMD, 177 us per ConvNeXt-L block where the tensors
// move in 25); dy is read once per filter row, from L2.  Every tap is still summed over the same runs in the same order: same bits.
template <int K, int ST, bool RS>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float *__restrict__ dy, const float *__restrict__ x, float *__restrict__ part,
                                                       DwGeom g, int runs_per_chunk) {
    __shared__ float4 red[16][16];
    constexpr int P = 4, NC = (P - 1) * ST + K;
    constexpr int KR = RS ? 1 : K;                        // filter rows of this workgroup
    const int rbase = RS ? (int)blockIdx.z : 0;
    const int cgl = threadIdx.x & 15, pl = threadIdx.x >> 4;          // 16 channel groups (64 channels) x 16 run lanes
    const int cg = blockIdx.y * 16 + cgl;
    const bool con = cg * 4 < g.C;
    const int wruns = (g.Wo + P - 1) / P;
    const size_t nruns = (size_t)g.N * g.Ho * wruns;
    const size_t q0 = (size_t)blockIdx.x * runs_per_chunk, q1 = q0 + runs_per_chunk < nruns ? q0 + runs_per_chunk : nruns;
    float4 acc[KR * K];
#pragma unroll
    for (int t = 0; t < KR * K; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (con)
        for (size_t q = q0 + pl; q < q1; q += 16) {
            const int wr = (int)(q % wruns), ho = (int)((q / wruns) % g.Ho), n = (int)(q / ((size_t)wruns * g.Ho));
            const int wo0 = wr * P;
            float4 gv[P];
#pragma unroll
            for (int p = 0; p < P; ++p)
                gv[p] = wo0 + p < g.Wo ? *reinterpret_cast<const float4 *>(dy + (((size_t)n * g.Ho + ho) * g.Wo + wo0 + p) * g.C + cg * 4) : zero;
            const int w0 = wo0 * ST - g.pad_l, h0 = ho * ST - g.pad_t;
#pragma unroll
            for (int rl = 0; rl < KR; ++rl) {
                const int r = rbase + rl;
                const int hi = h0 + r;
                if ((unsigned)hi >= (unsigned)g.H) continue;
                const float *xrow = x + ((size_t)n * g.H + hi) * g.W * g.C + cg * 4;
                float4 xv[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int wi = w0 + c;
                    xv[c] = (unsigned)wi < (unsigned)g.W ? *reinterpret_cast<const float4 *>(xrow + (size_t)wi * g.C) : zero;
                }
#pragma unroll
                for (int sx = 0; sx < K; ++sx) {
                    float4 &a = acc[rl * K + sx];
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const float4 xq = xv[p * ST + sx];
                        a.x = fmaf(gv[p].x, xq.x, a.x); a.y = fmaf(gv[p].y, xq.y, a.y);
                        a.z = fmaf(gv[p].z, xq.z, a.z); a.w = fmaf(gv[p].w, xq.w, a.w);
                    }
                }
            }
        }
    // fixed-order sum over the 16 run lanes, tap by tap
#pragma unroll
    for (int t = 0; t < KR * K; ++t) {
        red[pl][cgl] = acc[t];
        __syncthreads();
        if (pl == 0 && con) {
            float4 a = red[0][cgl];
            for (int q = 1; q < 16; ++q) {
                const float4 b = red[q][cgl];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            *reinterpret_cast<float4 *>(part + ((size_t)blockIdx.x * K * K + rbase * K + t) * g.C + cg * 4) = a;
        }
        __syncthreads();
    }
}

// [C][k*k] (torch depthwise filter [C,1,k,k]) <-> [k*k][C]
__global__ __launch_bounds__(256) void dw_weight_transpose_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int kk,
                                                                  int to_taps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * kk) return;
    const int c = i / kk, t = i % kk;
    if (to_taps) dst[(size_t)t * C + c] = src[i];
    else dst[i] = src[(size_t)t * C + c];
}

// ---------------------------------------------------------------------------------------------------
// squeeze-and-excite
// ---------------------------------------------------------------------------------------------------
// per-(image, pixel chunk) channel sums of a (or of a * b when b != NULL): part [B][nchunk][C]
__global__ __launch_bounds__(256) void se_pool_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ part,
                                                      int HW, int C, int px_per_chunk, int nchunk) {
    __shared__ float4 red[16][16];
    const int cgl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int cg = blockIdx.y * 16 + cgl, img = blockIdx.z, chunk = blockIdx.x;
    const bool con = cg * 4 < C;
    const int p0 = chunk * px_per_chunk, p1 = min(HW, p0 + px_per_chunk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (con)
        for (int p = p0 + pl; p < p1; p += 16) {
            const size_t o = ((size_t)img * HW + p) * C + cg * 4;
            float4 v = *reinterpret_cast<const float4 *>(a + o);
            if (b) {
                const float4 u = *reinterpret_cast<const float4 *>(b + o);
                v.x *= u.x; v.y *= u.y; v.z *= u.z; v.w *= u.w;
            }
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    red[pl][cgl] = acc;
    __syncthreads();
    if (pl == 0 && con) {
        float4 s = red[0][cgl];
        for (int q = 1; q < 16; ++q) {
            const float4 v = red[q][cgl];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4 *>(part + ((size_t)img * nchunk + chunk) * C + cg * 4) = s;
    }
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// The gate of one image: s = mean over pixels (sum of the chunk partials / HW); r = swish(W1 s + b1); gate = sigmoid(W2 r + b2).
// W1 [R][C], W2T [R][C] (the expansion filter transposed: lanes run along C in every loop).  saves s [B][C], pre1 [B][R].
// Two launches, both spread over the chip (one workgroup per image — 8 of 256 CUs, each reading its 1.5 MB of W1 alone — took 39 us per
// block of the trunk): (1) SE_SPLIT workgroups per image share the R squeeze dots (one wave per reduced channel; every workgroup forms
// s itself — the partials are small —, the first one stores it); (2) one thread per channel for the excite dots.  Same order of every
// sum as the one-workgroup kernel: the same bits.
constexpr int SE_SPLIT = 16;
__global__ __launch_bounds__(256) void se_squeeze_kernel(const float *__restrict__ part, int nchunk, const float *__restrict__ W1,
                                                         const float *__restrict__ b1, float *__restrict__ s_out, float *__restrict__ pre1_out,
                                                         int C, int R, float inv_hw) {
    extern __shared__ float sh[];            // s [C]
    float *s = sh;
    const int img = blockIdx.y, blk = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int k = 0; k < nchunk; ++k) a += part[((size_t)img * nchunk + k) * C + c];
        a *= inv_hw;
        s[c] = a;
        if (blk == 0) s_out[(size_t)img * C + c] = a;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (R + SE_SPLIT - 1) / SE_SPLIT, j0 = blk * per, j1 = min(R, j0 + per);
    for (int j = j0 + wave; j < j1; j += 4) {      // one wave per reduced channel
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(W1[(size_t)j * C + c], s[c], a);
        a = wave_sum(a) + b1[j];
        if (lane == 0) pre1_out[(size_t)img * R + j] = a;
    }
}
__global__ __launch_bounds__(256) void se_excite_kernel(const float *__restrict__ pre1, const float *__restrict__ W2, const float *__restrict__ b2,
                                                        float *__restrict__ gate, int C, int R) {
    __shared__ float r[256];                 // R <= 252
    const int img = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if ((int)threadIdx.x < R) {
        const float a = pre1[(size_t)img * R + threadIdx.x];
        r[threadIdx.x] = a * sigmoidf(a);
    }
    __syncthreads();
    if (c >= C) return;
    float a = b2[c];
    for (int j = 0; j < R; ++j) a = fmaf(W2[(size_t)j * C + c], r[j], a);
    gate[(size_t)img * C + c] = sigmoidf(a);
}

// backward of the gate for one image: dgate [C] = sum over pixels of dy * x (chunk partials) ->
// dW2part [B][C][R], db2part [B][C], dW1part [B][R][C], db1part [B][R rounded up to 4], ds [B][C] (gradient w.r.t. the pooled mean, already / HW)
// Two launches, as the forward (one workgroup of 1024 threads per image took 64 us per block of the trunk): (1) SE_SPLIT workgroups per
// image: every one forms dpre2 [C] and r [R] itself, writes its slice of channels of dW2part / db2part, then the dpre1 of its slice of
// reduced channels (one wave per dot over C) -> db1part and its rows of dW1part; (2) one thread per channel for ds, reading dpre1 back
// from db1part.  The same order of every sum as before: the same bits.
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float *__restrict__ dgpart, int nchunk, const float *__restrict__ W2,
                                                         const float *__restrict__ s_in, const float *__restrict__ pre1_in,
                                                         const float *__restrict__ gate, float *__restrict__ dW1p, float *__restrict__ db1p,
                                                         float *__restrict__ dW2p, float *__restrict__ db2p, int C, int R) {
    extern __shared__ float sh[];            // dpre2 [C], r [R], dpre1 [R]
    float *dpre2 = sh, *r = sh + C, *dpre1 = r + R;
    const int img = blockIdx.y, blk = blockIdx.x, RP = (R + 3) & ~3;
    const int cper = (C + SE_SPLIT - 1) / SE_SPLIT, c0 = blk * cper, c1 = min(C, c0 + cper);
    const int jper = (R + SE_SPLIT - 1) / SE_SPLIT, j0 = blk * jper, j1 = min(R, j0 + jper);
    for (int j = threadIdx.x; j < R; j += 256) {
        const float a = pre1_in[(size_t)img * R + j];
        r[j] = a * sigmoidf(a);
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        float dg = 0.f;
        for (int k = 0; k < nchunk; ++k) dg += dgpart[((size_t)img * nchunk + k) * C + c];
        const float gt = gate[(size_t)img * C + c];
        const float d = dg * gt * (1.f - gt);
        dpre2[c] = d;
        if (c >= c0 && c < c1) db2p[(size_t)img * C + c] = d;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (c1 - c0) * R; i += 256) {
        const int c = c0 + i / R, j = i % R;
        dW2p[((size_t)img * C + c) * R + j] = dpre2[c] * r[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = j0 + wave; j < j1; j += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(W2[(size_t)j * C + c], dpre2[c], a);
        a = wave_sum(a);
        if (lane == 0) {
            const float p = pre1_in[(size_t)img * R + j], sg = sigmoidf(p);
            const float d = a * (sg * (1.f + p * (1.f - sg)));            // swish'
            dpre1[j] = d;
            db1p[(size_t)img * RP + j] = d;
        }
    }
    if (blk == 0 && (int)threadIdx.x >= R && (int)threadIdx.x < RP) db1p[(size_t)img * RP + threadIdx.x] = 0.f;     // padding columns of the partial rows
    __syncthreads();
    for (int i = threadIdx.x; i < (j1 - j0) * C; i += 256) {
        const int j = j0 + i / C, c = i % C;
        dW1p[((size_t)img * R + j) * C + c] = dpre1[j] * s_in[(size_t)img * C + c];
    }
}
__global__ __launch_bounds__(256) void se_gate_bwd_ds_kernel(const float *__restrict__ W1, const float *__restrict__ db1p, float *__restrict__ ds,
                                                            int C, int R, float inv_hw) {
    __shared__ float dpre1[256];
    const int img = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x, RP = (R + 3) & ~3;
    if ((int)threadIdx.x < R) dpre1[threadIdx.x] = db1p[(size_t)img * RP + threadIdx.x];
    __syncthreads();
    if (c >= C) return;
    float a = 0.f;
    for (int j = 0; j < R; ++j) a = fmaf(W1[(size_t)j * C + c], dpre1[j], a);
    ds[(size_t)img * C + c] = a * inv_hw;
}

// y = x * gate[img, c]   |   backward: dx = dy * gate + ds[img, c]
template <bool BWD>
__global__ __launch_bounds__(256) void se_scale_kernel(const float *__restrict__ x, const float *__restrict__ gate, const float *__restrict__ ds,
                                                       float *__restrict__ y, size_t total4, int HW, int C) {
    const int V = C / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % V);
        const int img = (int)(i / ((size_t)V * HW));
        const float4 xv = reinterpret_cast<const float4 *>(x)[i];
        const float4 gv = *reinterpret_cast<const float4 *>(gate + (size_t)img * C + cg * 4);
        float4 o = make_float4(xv.x * gv.x, xv.y * gv.y, xv.z * gv.z, xv.w * gv.w);
        if (BWD) {
            const float4 dv = *reinterpret_cast<const float4 *>(ds + (size_t)img * C + cg * 4);
            o.x += dv.x; o.y += dv.y; o.z += dv.z; o.w += dv.w;
        }
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}

int dw_check(const char *who, const DwGeom &g) {
    SQD_CHECK_ARG(g.N > 0 && g.H > 0 && g.W > 0 && g.C >= 4 && g.C % 4 == 0 && (g.k == 3 || g.k == 5 || g.k == 7) && (g.stride == 1 || g.stride == 2) &&
                      g.pad_t >= 0 && g.pad_l >= 0 && g.Ho > 0 && g.Wo > 0,
                  "%s: unsupported depthwise geometry (C %% 4 == 0, k in {3,5,7}, stride in {1,2})", who);
    SQD_CHECK_ARG((long long)g.N * g.H * g.W * g.C < (1ll << 31) && (long long)g.N * g.Ho * g.Wo * g.C < (1ll << 31), "%s: tensor too large", who);
    return SQD_OK;
}
int ew_blocks(size_t n) {
    const size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}
}  // namespace

extern "C" int sqd_dw_weight_layout(const float *src, float *dst, int C, int k, int to_taps, void *stream) {
    SQD_CHECK_ARG(src && dst && C > 0 && k > 0, "sqd_dw_weight_layout: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(dw_weight_transpose_kernel, dim3((C * k * k + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, C, k * k, to_taps);
    SQD_CHECK_LAUNCH("sqd_dw_weight_layout");
    return SQD_OK;
}

extern "C" int sqd_dw_conv_fwd(const float *x, const float *w_taps, float *y, int N, int H, int W, int C, int k, int stride, int pad_t,
                               int pad_l, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(x && w_taps && y, "sqd_dw_conv_fwd: null pointer");
    const DwGeom g = {N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo};
    if (dw_check("sqd_dw_conv_fwd", g)) return SQD_EINVAL;
    (void)hipGetLastError();
    const dim3 grid(ew_blocks((size_t)N * Ho * ((Wo + 3) / 4) * C / 4));
    hipStream_t st = (hipStream_t)stream;
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_conv_run_kernel<3, 1, false>), grid, dim3(256), 0, st, x, w_taps, y, g, (const float *)nullptr);
    else if (k == 3) hipLaunchKernelGGL((dw_conv_run_kernel<3, 2, false>), grid, dim3(256), 0, st, x, w_taps, y, g, (const float *)nullptr);
    else if (k == 5 && stride == 1) hipLaunchKernelGGL((dw_conv_run_kernel<5, 1, false>), grid, dim3(256), 0, st, x, w_taps, y, g, (const float *)nullptr);
    else if (k == 5) hipLaunchKernelGGL((dw_conv_run_kernel<5, 2, false>), grid, dim3(256), 0, st, x, w_taps, y, g, (const float *)nullptr);
    else if (stride == 1) hipLaunchKernelGGL((dw_conv_run_kernel<7, 1, false>), grid, dim3(256), 0, st, x, w_taps, y, g, (const float *)nullptr);
    else hipLaunchKernelGGL((dw_conv_run_kernel<7, 2, false>), grid, dim3(256), 0, st, x, w_taps, y, g, (const float *)nullptr);
    SQD_CHECK_LAUNCH("sqd_dw_conv_fwd");
    return SQD_OK;
}

extern "C" int sqd_dw_conv_dgrad(const float *dy, const float *w_taps, float *dx, int N, int H, int W, int C, int k, int stride, int pad_t,
                                 int pad_l, int Ho, int Wo, void *stream) {
    return sqd_dw_conv_dgrad_add(dy, w_taps, nullptr, dx, N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, stream);
}

// dx = data gradient + addend [N,H,W,C] (NULL: none; stride 1 only): the gradient arriving over the input's second path — a block's
// shortcut — is added here instead of by a separate accumulation pass
extern "C" int sqd_dw_conv_dgrad_add(const float *dy, const float *w_taps, const float *addend, float *dx, int N, int H, int W, int C, int k,
                                     int stride, int pad_t, int pad_l, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(dy && w_taps && dx, "sqd_dw_conv_dgrad: null pointer");
    SQD_CHECK_ARG(!addend || (stride == 1 && addend != dx), "sqd_dw_conv_dgrad_add: the addend needs stride 1 and must not alias dx");
    const DwGeom g = {N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo};
    if (dw_check("sqd_dw_conv_dgrad", g)) return SQD_EINVAL;
    (void)hipGetLastError();
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ew_blocks((size_t)N * H * ((W + 3) / 4) * C / 4));
    if (stride == 1 && k == 3) hipLaunchKernelGGL((dw_conv_run_kernel<3, 1, true>), grid, dim3(256), 0, st, dy, w_taps, dx, g, addend);
    else if (stride == 1 && k == 5) hipLaunchKernelGGL((dw_conv_run_kernel<5, 1, true>), grid, dim3(256), 0, st, dy, w_taps, dx, g, addend);
    else if (stride == 1) hipLaunchKernelGGL((dw_conv_run_kernel<7, 1, true>), grid, dim3(256), 0, st, dy, w_taps, dx, g, addend);
    else     // stride 2: the taps of an input pixel depend on its stride phase — the one-pixel gather
        hipLaunchKernelGGL((dw_conv_kernel<1>), dim3(ew_blocks((size_t)N * H * W * C / 4)), dim3(256), 0, st, dy, w_taps, dx, g);
    SQD_CHECK_LAUNCH("sqd_dw_conv_dgrad");
    return SQD_OK;
}

extern "C" int sqd_dw_conv_wgrad_chunks(int N, int Ho, int Wo) {
    const long long M = (long long)N * Ho * Wo;
    long long n = (M + 2047) / 2048;                 // >= 2048 output pixels per chunk
    return (int)(n < 1 ? 1 : n > 512 ? 512 : n);
}
// part [chunks][k*k][C] (chunks = sqd_dw_conv_wgrad_chunks); sum over the chunks (sqd_colsum_multi) gives dw in tap-major layout
extern "C" int sqd_dw_conv_wgrad(const float *dy, const float *x, float *part, int N, int H, int W, int C, int k, int stride, int pad_t,
                                 int pad_l, int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(dy && x && part, "sqd_dw_conv_wgrad: null pointer");
    const DwGeom g = {N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo};
    if (dw_check("sqd_dw_conv_wgrad", g)) return SQD_EINVAL;
    const int chunks = sqd_dw_conv_wgrad_chunks(N, Ho, Wo);
    const long long nruns = (long long)N * Ho * ((Wo + 3) / 4);
    const int rpc = (int)((nruns + chunks - 1) / chunks);
    const dim3 grid(chunks, (C / 4 + 15) / 16), grid_rs(chunks, (C / 4 + 15) / 16, k);      // (row split: one filter row per workgroup)
    (void)hipGetLastError();
    hipStream_t st = (hipStream_t)stream;
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_wgrad_kernel<3, 1, true>), grid_rs, dim3(256), 0, st, dy, x, part, g, rpc);
    else if (k == 3) hipLaunchKernelGGL((dw_wgrad_kernel<3, 2, true>), grid_rs, dim3(256), 0, st, dy, x, part, g, rpc);
    else if (k == 5 && stride == 1) hipLaunchKernelGGL((dw_wgrad_kernel<5, 1, true>), grid_rs, dim3(256), 0, st, dy, x, part, g, rpc);
    else if (k == 5) hipLaunchKernelGGL((dw_wgrad_kernel<5, 2, true>), grid_rs, dim3(256), 0, st, dy, x, part, g, rpc);
    else if (stride == 1) hipLaunchKernelGGL((dw_wgrad_kernel<7, 1, true>), grid_rs, dim3(256), 0, st, dy, x, part, g, rpc);
    else hipLaunchKernelGGL((dw_wgrad_kernel<7, 2, true>), grid_rs, dim3(256), 0, st, dy, x, part, g, rpc);
    SQD_CHECK_LAUNCH("sqd_dw_conv_wgrad");
    return SQD_OK;
}

extern "C" int sqd_se_chunks(int HW) {
    const int n = (HW + 1023) / 1024;
    return n < 1 ? 1 : n > 256 ? 256 : n;
}
// part [B][chunks][C] = per-chunk channel sums of a (b == NULL) or of a * b
extern "C" int sqd_se_pool(const float *a, const float *b, float *part, int B, int HW, int C, void *stream) {
    SQD_CHECK_ARG(a && part && B > 0 && HW > 0 && C >= 4 && C % 4 == 0, "sqd_se_pool: bad arguments (C %% 4 == 0)");
    const int nchunk = sqd_se_chunks(HW), ppc = (HW + nchunk - 1) / nchunk;
    (void)hipGetLastError();
    hipLaunchKernelGGL(se_pool_kernel, dim3(nchunk, (C / 4 + 15) / 16, B), dim3(256), 0, (hipStream_t)stream, a, b, part, HW, C, ppc, nchunk);
    SQD_CHECK_LAUNCH("sqd_se_pool");
    return SQD_OK;
}

extern "C" int sqd_se_gate_fwd(const float *part, const float *W1, const float *b1, const float *W2, const float *b2, float *s, float *pre1,
                               float *gate, int B, int HW, int C, int R, void *stream) {
    SQD_CHECK_ARG(part && W1 && b1 && W2 && b2 && s && pre1 && gate && B > 0 && C > 0 && R > 0 && (size_t)(C + R) * 4 <= 64 * 1024,
                  "sqd_se_gate_fwd: bad arguments");
    (void)hipGetLastError();
    SQD_CHECK_ARG(R <= 252, "sqd_se_gate_fwd: R=%d (at most 252 reduced channels)", R);
    hipLaunchKernelGGL(se_squeeze_kernel, dim3(SE_SPLIT, B), dim3(256), C * sizeof(float), (hipStream_t)stream, part, sqd_se_chunks(HW), W1, b1, s,
                       pre1, C, R, 1.0f / (float)HW);
    hipLaunchKernelGGL(se_excite_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, pre1, W2, b2, gate, C, R);
    SQD_CHECK_LAUNCH("sqd_se_gate_fwd");
    return SQD_OK;
}

extern "C" int sqd_se_gate_bwd(const float *dgpart, const float *W1, const float *W2, const float *s, const float *pre1, const float *gate,
                               float *dW1part, float *db1part, float *dW2part, float *db2part, float *ds, int B, int HW, int C, int R,
                               void *stream) {
    SQD_CHECK_ARG(dgpart && W1 && W2 && s && pre1 && gate && dW1part && db1part && dW2part && db2part && ds && B > 0 && C > 0 && R > 0 &&
                      (size_t)(C + 2 * R) * 4 <= 64 * 1024 && R <= 252,
                  "sqd_se_gate_bwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(SE_SPLIT, B), dim3(256), (C + 2 * R) * sizeof(float), (hipStream_t)stream, dgpart, sqd_se_chunks(HW), W2,
                       s, pre1, gate, dW1part, db1part, dW2part, db2part, C, R);
    hipLaunchKernelGGL(se_gate_bwd_ds_kernel, dim3((C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, W1, db1part, ds, C, R, 1.0f / (float)HW);
    SQD_CHECK_LAUNCH("sqd_se_gate_bwd");
    return SQD_OK;
}

// forward: y = x * gate (ds == NULL);  backward: y = x * gate + ds  (x = dy)
extern "C" int sqd_se_scale(const float *x, const float *gate, const float *ds, float *y, int B, int HW, int C, void *stream) {
    SQD_CHECK_ARG(x && gate && y && B > 0 && HW > 0 && C >= 4 && C % 4 == 0, "sqd_se_scale: bad arguments");
    const size_t total4 = (size_t)B * HW * C / 4;
    (void)hipGetLastError();
    if (ds) hipLaunchKernelGGL((se_scale_kernel<true>), dim3(ew_blocks(total4)), dim3(256), 0, (hipStream_t)stream, x, gate, ds, y, total4, HW, C);
    else hipLaunchKernelGGL((se_scale_kernel<false>), dim3(ew_blocks(total4)), dim3(256), 0, (hipStream_t)stream, x, gate, ds, y, total4, HW, C);
    SQD_CHECK_LAUNCH("sqd_se_scale");
    return SQD_OK;
}
