// bn_act.hip — BatchNorm2d (training and eval) fused with the activation and the residual add, for
// activations stored channels-last ([N,H,W,C] in memory = a row-major [M = N*H*W, C] matrix).
//
// replaces, per conv block of the networks (reference networks/resnet_encoder.py:89-117 through
// torchvision's Bottleneck/BasicBlock, and UpSampleBN): MIOpenBatchNormFwdTrainSpatial + ReLU/LeakyReLU
// (+ residual add) forward, MIOpenBatchNormBwdSpatial + threshold_backward (+ add) backward.
//
// Forward, training:  stats pass (sum, sum of squares per channel, deterministic two-level reduction)
//                     -> finalize (mean, rstd, running-stat update: momentum 0.1, unbiased variance)
//                     -> apply   y = act(gamma * (x - mean) * rstd + beta [+ residual])
// Backward, training: reduce pass (sum dz, sum dz * xhat with dz = dy * act'(y))
//                     -> apply   dx = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat)),  dres = dz
// Roofline: HBM — forward 2 reads + 1 write of x per element (3 with a residual), backward 5 accesses.
#include "sqd_common.h"

namespace {
using namespace sqd;

constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SWISH = 3;      // swish = x * sigmoid(x) (EfficientNet)
constexpr float LEAKY_SLOPE = 0.01f;

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LEAKY) return v > 0.f ? v : v * LEAKY_SLOPE;
    if (act == ACT_SWISH) return v / (1.f + __expf(-v));
    return v;
}
// swish has no sign-mask shortcut: its derivative needs the pre-activation v = gamma * xhat + beta, recomputed from x
__device__ __forceinline__ float swish_bwd(float v) {
    const float sg = 1.f / (1.f + __expf(-v));
    return sg * (1.f + v * (1.f - sg));
}
// derivative expressed through the OUTPUT y (what the backward has at hand)
__device__ __forceinline__ float act_bwd(float y, int act) {
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_LEAKY) return y > 0.f ? 1.f : LEAKY_SLOPE;
    return 1.f;
}
// ... or through the sign bit the forward stored (mask byte of a float4 group: bit j = element j was positive): the
// backward then reads 1 byte instead of 16 to know the activation's derivative
__device__ __forceinline__ float act_bwd_bit(unsigned bits, int j, int act) {
    const bool pos = (bits >> j) & 1u;
    if (act == ACT_RELU) return pos ? 1.f : 0.f;
    if (act == ACT_LEAKY) return pos ? 1.f : LEAKY_SLOPE;
    return 1.f;
}
// the four activation derivatives of float4 group `idx`: from the mask when there is one, else from y (not read at all
// when there is no activation)
__device__ __forceinline__ float4 act_bwd4(const unsigned char *__restrict__ mask, const float *__restrict__ y, size_t idx, int act) {
    if (act == ACT_NONE) return make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) {
        const unsigned bits = mask[idx];
        return make_float4(act_bwd_bit(bits, 0, act), act_bwd_bit(bits, 1, act), act_bwd_bit(bits, 2, act), act_bwd_bit(bits, 3, act));
    }
    const float4 yv = reinterpret_cast<const float4 *>(y)[idx];
    return make_float4(act_bwd(yv.x, act), act_bwd(yv.y, act), act_bwd(yv.z, act), act_bwd(yv.w, act));
}

struct Geom {
    int V, TPR, RP, rows_per_block, nblk;
};
__host__ __device__ inline Geom geom(int M, int C) {
    Geom g;
    g.V = C / 4;
    g.TPR = g.V < 256 ? g.V : 256;             // threads across one row (each a float4 of channels)
    g.RP = 256 / g.TPR;                        // rows processed in parallel by a block
    int rpb = (M + 767) / 768;                 // aim at <= 768 blocks
    rpb = ((rpb + g.RP - 1) / g.RP) * g.RP;
    if (rpb < g.RP * 4) rpb = g.RP * 4;
    g.rows_per_block = rpb;
    g.nblk = (M + rpb - 1) / rpb;
    return g;
}

// per-channel partial sums of two row-wise quantities: MODE 0: (x, x^2);  MODE 1: (dz, dz * xhat)
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                        const float *__restrict__ y, const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, float *__restrict__ part, int M,
                                                        int C, int act, Geom g, const unsigned char *__restrict__ mask,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta) {
    const int t = threadIdx.x;
    const int cg0 = t % g.TPR, rr = t / g.TPR;
    const int r0 = blockIdx.x * g.rows_per_block, r1 = min(M, r0 + g.rows_per_block);
    __shared__ float4 sh[2][256];
    for (int cg = cg0; cg < g.V; cg += g.TPR) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        float4 mu, rs, ga, be;
        if (MODE == 1) {
            mu = *reinterpret_cast<const float4 *>(mean + cg * 4);
            rs = *reinterpret_cast<const float4 *>(rstd + cg * 4);
            if (act == ACT_SWISH) {
                ga = *reinterpret_cast<const float4 *>(gamma + cg * 4);
                be = *reinterpret_cast<const float4 *>(beta + cg * 4);
            }
        }
        // (C / 4 need not divide 256 — EfficientNet's 24, 40, 48, 144 ... channels: the 256 - RP * TPR left-over threads idle)
        for (int row = r0 + rr; rr < g.RP && row < r1; row += g.RP) {
            const size_t o = (size_t)row * C + cg * 4;
            const float4 xv = *reinterpret_cast<const float4 *>(x + o);
            if (MODE == 0) {
                a.x += xv.x; a.y += xv.y; a.z += xv.z; a.w += xv.w;
                b.x = fmaf(xv.x, xv.x, b.x); b.y = fmaf(xv.y, xv.y, b.y); b.z = fmaf(xv.z, xv.z, b.z); b.w = fmaf(xv.w, xv.w, b.w);
            } else {
                const float4 gy = *reinterpret_cast<const float4 *>(dy + o);
                float4 da;
                if (act == ACT_SWISH)
                    da = make_float4(swish_bwd(fmaf((xv.x - mu.x) * rs.x, ga.x, be.x)), swish_bwd(fmaf((xv.y - mu.y) * rs.y, ga.y, be.y)),
                                     swish_bwd(fmaf((xv.z - mu.z) * rs.z, ga.z, be.z)), swish_bwd(fmaf((xv.w - mu.w) * rs.w, ga.w, be.w)));
                else
                    da = act_bwd4(mask, y, o / 4, act);
                const float d0 = gy.x * da.x, d1 = gy.y * da.y, d2 = gy.z * da.z, d3 = gy.w * da.w;
                a.x += d0; a.y += d1; a.z += d2; a.w += d3;
                b.x = fmaf(d0, (xv.x - mu.x) * rs.x, b.x); b.y = fmaf(d1, (xv.y - mu.y) * rs.y, b.y);
                b.z = fmaf(d2, (xv.z - mu.z) * rs.z, b.z); b.w = fmaf(d3, (xv.w - mu.w) * rs.w, b.w);
            }
        }
        sh[0][t] = a;
        sh[1][t] = b;
        __syncthreads();
        if (rr == 0) {
            for (int k = 1; k < g.RP; ++k) {
                const float4 a2 = sh[0][k * g.TPR + cg0], b2 = sh[1][k * g.TPR + cg0];
                a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
                b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
            }
            float *o = part + ((size_t)blockIdx.x * C + cg * 4) * 2;
            o[0] = a.x; o[1] = b.x; o[2] = a.y; o[3] = b.y; o[4] = a.z; o[5] = b.z; o[6] = a.w; o[7] = b.w;
        }
        __syncthreads();
    }
}

// Sum the per-block partials of FIN_CH = 4 consecutive channels per workgroup: the (sum, sum of squares) pairs of four channels
// are 32 contiguous bytes of a partial row, so each of the 256 threads strides over the rows with two 16-byte loads per row (four
// rows in flight), then a fixed-order reduction: DPP butterflies inside each wave, the four waves through LDS (deterministic).
// History: one thread per channel over up to 768 partials 131 us per call; 16 lanes per channel + an LDS tree 12 us; one wave per
// channel with 8-byte loads 5.2 us.  Round 3 put all 256 threads on the four channels and eight rows in flight per thread (2880
// partial rows = two rounds of loads instead of eleven): the in-step average stayed at 5.0 us per call — the kernel is not bound by
// its load rounds (a replayed graph's chain of EMPTY kernels costs 1.5 us per node, tools/ubench_graph_nodes.py; the rest is this
// kernel's launch ramp over C / 4 workgroups, its first touch of partials another XCD's L2 wrote, two barriers and the running-stat
// read-modify-write).  120 calls x 5 us per step remain: only fewer launches remove them.
constexpr int FIN_CH = 4;
__device__ __forceinline__ bool finalize_sums(const float *__restrict__ part, int nblk, int C, int &c, float &s, float &ss) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * FIN_CH;                  // C % 4 == 0: the four channels exist
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), b0 = a0, a1 = a0, b1 = a0;
    const float *p = part + (size_t)c0 * 2;
    const size_t stride = (size_t)C * 2;
    // eight rows (sixteen 16-byte loads) in flight per thread and round
    constexpr int DEPTH = 8;
    for (int k0 = threadIdx.x; k0 < nblk; k0 += 256 * DEPTH) {
        float4 u[DEPTH], v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int k = k0 + 256 * d;
            u[d] = v[d] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < nblk) {
                u[d] = *reinterpret_cast<const float4 *>(p + (size_t)k * stride);
                v[d] = *reinterpret_cast<const float4 *>(p + (size_t)k * stride + 4);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d += 2) {
            a0.x += u[d].x; a0.y += u[d].y; a0.z += u[d].z; a0.w += u[d].w; b0.x += v[d].x; b0.y += v[d].y; b0.z += v[d].z; b0.w += v[d].w;
            a1.x += u[d + 1].x; a1.y += u[d + 1].y; a1.z += u[d + 1].z; a1.w += u[d + 1].w;
            b1.x += v[d + 1].x; b1.y += v[d + 1].y; b1.z += v[d + 1].z; b1.w += v[d + 1].w;
        }
    }
    // (sum, sum of squares) of channels c0 .. c0+3 in this thread: a = (s0, ss0, s1, ss1), b = (s2, ss2, s3, ss3)
    float v[8] = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w, b0.x + b1.x, b0.y + b1.y, b0.z + b1.z, b0.w + b1.w};
    __shared__ float sh[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = wave_sum_to_lane63(v[j]);
        if (lane == 63) sh[wave][j] = t;
    }
    __syncthreads();
    if (threadIdx.x >= FIN_CH) return false;
    c = c0 + threadIdx.x;
    s = (sh[0][2 * threadIdx.x] + sh[1][2 * threadIdx.x]) + (sh[2][2 * threadIdx.x] + sh[3][2 * threadIdx.x]);
    ss = (sh[0][2 * threadIdx.x + 1] + sh[1][2 * threadIdx.x + 1]) + (sh[2][2 * threadIdx.x + 1] + sh[3][2 * threadIdx.x + 1]);
    return c < C;
}

// forward finalize: mean / rstd from the partials + running statistics (nn.BatchNorm2d semantics)
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float *__restrict__ part, int nblk, int M, int C,
                                                              float eps, float momentum, float *__restrict__ mean,
                                                              float *__restrict__ rstd, float *__restrict__ rmean,
                                                              float *__restrict__ rvar) {
    int c;
    float s, ss;
    // the running statistics are fetched before the partials are summed, not after: one dependent memory latency less at the end of a
    // kernel that is nothing but a latency chain
    float rm = 0.f, rv = 0.f;
    if (rmean && threadIdx.x < FIN_CH && (int)(blockIdx.x * FIN_CH + threadIdx.x) < C) {
        rm = rmean[blockIdx.x * FIN_CH + threadIdx.x];
        rv = rvar[blockIdx.x * FIN_CH + threadIdx.x];
    }
    if (!finalize_sums(part, nblk, C, c, s, ss)) return;
    const float m = s / (float)M;
    float var = ss / (float)M - m * m;                    // biased (normalisation) variance
    var = var < 0.f ? 0.f : var;
    mean[c] = m;
    rstd[c] = rsqrtf(var + eps);
    if (rmean) {
        const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
        rmean[c] = (1.f - momentum) * rm + momentum * m;
        rvar[c] = (1.f - momentum) * rv + momentum * unbiased;
    }
}

// backward finalize: dgamma, dbeta (and the two means the apply pass needs, stored in `red`)
// Workgroups beyond the nfin finalize ones carry a pending split reduction along (a weight gradient's pixel splits: one launch less
// per layer; red_part == NULL: none).
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float *__restrict__ part, int nblk, int M, int C,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta, int nfin,
                                                              const float *__restrict__ red_part, float *__restrict__ red_out, size_t red_n,
                                                              int red_splits) {
    if ((int)blockIdx.x >= nfin) {                  // (workgroup-uniform)
        __shared__ float4 red[16][16];
        sqd::split_reduce_block(red_part, red_out, red_n, red_splits, (int)blockIdx.x - nfin, red);
        return;
    }
    int c;
    float s, ss;
    if (!finalize_sums(part, nblk, C, c, s, ss)) return;
    dbeta[c] = s;
    dgamma[c] = ss;
}

// EVAL: mean/rstd arguments hold running_mean / running_var (rstd computed on the fly)
template <bool EVAL>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           float *__restrict__ y, size_t total4, int C, float eps, int act,
                                                           unsigned char *__restrict__ mask, unsigned *__restrict__ amax_y) {
    unsigned am = 0u;                                           // max |y| of this thread's elements (amax_y == NULL: not recorded)
    // the channel group of a thread's element advances by (grid stride mod V) per iteration — zero whenever V divides 256, i.e.
    // for every C <= 1024: the per-channel constants are then loaded once, and no 64-bit modulo runs inside the loop
    const unsigned V = (unsigned)C / 4u;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned cg = (unsigned)(i % V);
    const unsigned cstep = (unsigned)(stride % V);
    float4 ga, be, mu, rs;
    auto params = [&]() {
        ga = reinterpret_cast<const float4 *>(gamma)[cg];
        be = reinterpret_cast<const float4 *>(beta)[cg];
        mu = reinterpret_cast<const float4 *>(mean)[cg];
        rs = reinterpret_cast<const float4 *>(rstd)[cg];
        if (EVAL) rs = make_float4(rsqrtf(rs.x + eps), rsqrtf(rs.y + eps), rsqrtf(rs.z + eps), rsqrtf(rs.w + eps));
    };
    params();
    for (; i < total4; i += stride) {
        const float4 xv = reinterpret_cast<const float4 *>(x)[i];
        float4 o;
        o.x = fmaf((xv.x - mu.x) * rs.x, ga.x, be.x);
        o.y = fmaf((xv.y - mu.y) * rs.y, ga.y, be.y);
        o.z = fmaf((xv.z - mu.z) * rs.z, ga.z, be.z);
        o.w = fmaf((xv.w - mu.w) * rs.w, ga.w, be.w);
        if (res) {
            const float4 rv = reinterpret_cast<const float4 *>(res)[i];
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (!EVAL && mask)
            mask[i] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
        o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
        reinterpret_cast<float4 *>(y)[i] = o;
        am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
        if (cstep) {                                            // uniform
            cg += cstep;
            cg -= cg >= V ? V : 0u;
            params();
        }
    }
    amax_commit(am, amax_y);
}

__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                           const float *__restrict__ y, const float *__restrict__ gamma,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                           float *__restrict__ dx, float *__restrict__ dres, size_t total4,
                                                           int C, float invM, int act, const unsigned char *__restrict__ mask,
                                                           const float *__restrict__ beta, unsigned *__restrict__ amax_dx,
                                                           unsigned *__restrict__ amax_dres, const float *__restrict__ x2 = nullptr,
                                                           const float *__restrict__ mean2 = nullptr, const float *__restrict__ rstd2 = nullptr,
                                                           float *__restrict__ part2 = nullptr, float *__restrict__ cpart = nullptr) {
    // cpart != NULL (same launch rule): the column sums of dx, one partial row [C] per workgroup — the bias gradient of the convolution that
    // produced x, when its weight-gradient kernel leaves none behind (a pass over dx of its own otherwise: colsum_kernel, two launches)
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    // x2 != NULL (launched with 256 % (C / 4) == 0: a thread keeps its channel group): dres = dz is ALSO the whole gradient of the BatchNorm that
    // produced the residual branch (a bottleneck's down-sample BatchNorm: no activation, one consumer) — its two backward sums
    // (sum dz, sum dz * xhat2) are taken here, one partial row per workgroup, instead of by a pass of its own over dres and x2
    __shared__ float4 red2[2][256];
    float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f), q2 = s2, mu2 = s2, rs2 = s2;
    unsigned am = 0u, amr = 0u;
    const unsigned V = (unsigned)C / 4u;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned cg = (unsigned)(i % V);
    const unsigned cstep = (unsigned)(stride % V);              // 0 whenever V divides 256 (C <= 1024): constants loaded once
    float4 k0, k1, k2, mu, be = make_float4(0.f, 0.f, 0.f, 0.f);   // dx = k0 * dz - k1 - (x - mu) * k2
    auto params = [&]() {
        const float4 ga = reinterpret_cast<const float4 *>(gamma)[cg], rs = reinterpret_cast<const float4 *>(rstd)[cg];
        if (act == ACT_SWISH) be = reinterpret_cast<const float4 *>(beta)[cg];
        const float4 dg = reinterpret_cast<const float4 *>(dgamma)[cg], db = reinterpret_cast<const float4 *>(dbeta)[cg];
        mu = reinterpret_cast<const float4 *>(mean)[cg];
        k0 = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);
        k1 = make_float4(db.x * invM, db.y * invM, db.z * invM, db.w * invM);
        k2 = make_float4(rs.x * (dg.x * invM), rs.y * (dg.y * invM), rs.z * (dg.z * invM), rs.w * (dg.w * invM));
    };
    params();
    if (x2) {
        mu2 = reinterpret_cast<const float4 *>(mean2)[cg];
        rs2 = reinterpret_cast<const float4 *>(rstd2)[cg];
    }
    for (; i < total4; i += stride) {
        const float4 gy = reinterpret_cast<const float4 *>(dy)[i];
        const float4 xv = reinterpret_cast<const float4 *>(x)[i];
        float4 da;
        if (act == ACT_SWISH)          // pre-activation = gamma * xhat + beta = k0 * (x - mu) + beta
            da = make_float4(swish_bwd(fmaf(xv.x - mu.x, k0.x, be.x)), swish_bwd(fmaf(xv.y - mu.y, k0.y, be.y)),
                             swish_bwd(fmaf(xv.z - mu.z, k0.z, be.z)), swish_bwd(fmaf(xv.w - mu.w, k0.w, be.w)));
        else
            da = act_bwd4(mask, y, i, act);
        float4 dz, o;
        dz.x = gy.x * da.x; dz.y = gy.y * da.y; dz.z = gy.z * da.z; dz.w = gy.w * da.w;
        o.x = k0.x * (dz.x - k1.x - (xv.x - mu.x) * k2.x);
        o.y = k0.y * (dz.y - k1.y - (xv.y - mu.y) * k2.y);
        o.z = k0.z * (dz.z - k1.z - (xv.z - mu.z) * k2.z);
        o.w = k0.w * (dz.w - k1.w - (xv.w - mu.w) * k2.w);
        reinterpret_cast<float4 *>(dx)[i] = o;
        if (dres) reinterpret_cast<float4 *>(dres)[i] = dz;
        if (x2) {
            const float4 xw = reinterpret_cast<const float4 *>(x2)[i];
            s2.x += dz.x; s2.y += dz.y; s2.z += dz.z; s2.w += dz.w;
            q2.x = fmaf(dz.x, (xw.x - mu2.x) * rs2.x, q2.x); q2.y = fmaf(dz.y, (xw.y - mu2.y) * rs2.y, q2.y);
            q2.z = fmaf(dz.z, (xw.z - mu2.z) * rs2.z, q2.z); q2.w = fmaf(dz.w, (xw.w - mu2.w) * rs2.w, q2.w);
        }
        cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
        am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
        amr = max(max(amr, abs_bits(dz.x)), max(abs_bits(dz.y), max(abs_bits(dz.z), abs_bits(dz.w))));
        if (cstep) {                                            // uniform
            cg += cstep;
            cg -= cg >= V ? V : 0u;
            params();
        }
    }
    if (x2) {      // (uniform) the workgroup's partial row: the threads of a channel group added in a fixed order
        red2[0][threadIdx.x] = s2;
        red2[1][threadIdx.x] = q2;
        __syncthreads();
        if (threadIdx.x < V) {
            for (unsigned k = threadIdx.x + V; k < 256u; k += V) {
                const float4 u = red2[0][k], v = red2[1][k];
                s2.x += u.x; s2.y += u.y; s2.z += u.z; s2.w += u.w;
                q2.x += v.x; q2.y += v.y; q2.z += v.z; q2.w += v.w;
            }
            float *o = part2 + ((size_t)blockIdx.x * C + threadIdx.x * 4) * 2;
            reinterpret_cast<float4 *>(o)[0] = make_float4(s2.x, q2.x, s2.y, q2.y);
            reinterpret_cast<float4 *>(o)[1] = make_float4(s2.z, q2.z, s2.w, q2.w);
        }
    }
    if (cpart) {   // (uniform)
        __syncthreads();
        red2[0][threadIdx.x] = cs;
        __syncthreads();
        if (threadIdx.x < V) {
            for (unsigned k = threadIdx.x + V; k < 256u; k += V) {
                const float4 u = red2[0][k];
                cs.x += u.x; cs.y += u.y; cs.z += u.z; cs.w += u.w;
            }
            reinterpret_cast<float4 *>(cpart + (size_t)blockIdx.x * C)[threadIdx.x] = cs;
        }
    }
    amax_commit(am, amax_dx);
    if (dres) amax_commit(amr, amax_dres, 1);
}

// bn_apply_fwd_kernel for a BatchNorm whose output goes into a squeeze-and-excite gate: the same values, and on the way the per-image
// channel sums the gate's pooled mean needs (part [B][nchunk][C], the layout, chunking and summation order of se_pool_kernel in
// effnet.hip: bit-identical to pooling y afterwards) — the gate does not read the activation a second time.
// grid (nchunk, ceil(C / 64), B): thread (channel group t & 15, pixel lane t >> 4) walks pixels p0 + lane, + 16, ... of its chunk.
__global__ __launch_bounds__(256) void bn_apply_pool_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, const float *__restrict__ mean,
                                                                const float *__restrict__ rstd, float *__restrict__ y,
                                                                unsigned char *__restrict__ mask, float *__restrict__ part, int HW, int C,
                                                                int act, int px_per_chunk, int nchunk) {
    __shared__ float4 red[16][16];
    const int cgl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int cg = blockIdx.y * 16 + cgl, img = blockIdx.z, chunk = blockIdx.x;
    const bool con = cg * 4 < C;
    const int p0 = chunk * px_per_chunk, p1 = min(HW, p0 + px_per_chunk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (con) {
        const float4 ga = reinterpret_cast<const float4 *>(gamma)[cg], be = reinterpret_cast<const float4 *>(beta)[cg];
        const float4 mu = reinterpret_cast<const float4 *>(mean)[cg], rs = reinterpret_cast<const float4 *>(rstd)[cg];
#pragma unroll 4
        for (int p = p0 + pl; p < p1; p += 16) {
            const size_t i4 = (((size_t)img * HW + p) * C) / 4 + cg;
            const float4 xv = reinterpret_cast<const float4 *>(x)[i4];
            float4 o;
            o.x = fmaf((xv.x - mu.x) * rs.x, ga.x, be.x);
            o.y = fmaf((xv.y - mu.y) * rs.y, ga.y, be.y);
            o.z = fmaf((xv.z - mu.z) * rs.z, ga.z, be.z);
            o.w = fmaf((xv.w - mu.w) * rs.w, ga.w, be.w);
            if (mask) mask[i4] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
            o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
            reinterpret_cast<float4 *>(y)[i4] = o;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
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

// ---------------------------------------------------------------------------------------------------
// Finalize + element-wise pass in ONE launch, for layers whose statistics arrive as FEW partial rows (the 12x40 and 6x20 maps of
// layer3 / layer4: 90 / 23 rows from the convolution's epilogue).  No cross-workgroup dependency: a workgroup owns 64 channels x a
// block of rows and sums the partial rows of ITS 64 channels itself, in its prologue (rows x 512 bytes out of L2, fixed order — every
// workgroup of a channel chunk computes the same bits); the workgroup of row block 0 publishes mean / rstd (or dgamma / dbeta) and the
// running statistics.  Redundant L2 reads instead of a 5 us launch + a kernel boundary per layer and direction.  Not for
// many partial rows (360 at 24x80: 184 KB per workgroup — more than its slice of the tensor) and not with a cross-workgroup wait
// (round 3 measured that: 1.4 - 3.7 x slower steps).
// ---------------------------------------------------------------------------------------------------
#ifndef SQD_BN_FUSE_MAX_ROWS
#define SQD_BN_FUSE_MAX_ROWS 160
#endif
constexpr int FUSE_MAX_ROWS = SQD_BN_FUSE_MAX_ROWS;
// (sum, sum2) of the thread's 4 channels over all partial rows -> s[0..3], q[0..3]; thread = (channel group t & 15, row lane t >> 4)
__device__ __forceinline__ void chunk_sums(const float *__restrict__ part, int rows, int C, int c0, float4 &s, float4 &q, float4 (*red)[16][2]) {
    const int cgl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;        // a = (s0, q0, s1, q1), b = (s2, q2, s3, q3)
    const float *p = part + ((size_t)c0 + cgl * 4) * 2;
    for (int r = rl; r < rows; r += 16) {
        const float4 u = *reinterpret_cast<const float4 *>(p + (size_t)r * C * 2), v = *reinterpret_cast<const float4 *>(p + (size_t)r * C * 2 + 4);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    red[rl][cgl][0] = a;
    red[rl][cgl][1] = b;
    __syncthreads();
    a = red[0][cgl][0];
    b = red[0][cgl][1];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        const float4 u = red[k][cgl][0], v = red[k][cgl][1];
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    s = make_float4(a.x, a.z, b.x, b.z);
    q = make_float4(a.y, a.w, b.y, b.w);
}

__global__ __launch_bounds__(256) void bn_fused_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const float *__restrict__ part, int rows,
                                                           float *__restrict__ mean_out, float *__restrict__ rstd_out, float *__restrict__ rmean,
                                                           float *__restrict__ rvar, float *__restrict__ y, unsigned char *__restrict__ mask, int M,
                                                           int C, float eps, float momentum, int act, int rows_per_block, unsigned *__restrict__ amax_y) {
    __shared__ float4 red[16][16][2];
    const int cgl = threadIdx.x & 15, rl = threadIdx.x >> 4, c0 = blockIdx.y * 64, c = c0 + cgl * 4;
    float4 s, q;
    chunk_sums(part, rows, C, c0, s, q, red);
    float4 mu, rs;
    {
        const float m0 = s.x / (float)M, m1 = s.y / (float)M, m2 = s.z / (float)M, m3 = s.w / (float)M;
        float v0 = q.x / (float)M - m0 * m0, v1 = q.y / (float)M - m1 * m1, v2 = q.z / (float)M - m2 * m2, v3 = q.w / (float)M - m3 * m3;
        v0 = v0 < 0.f ? 0.f : v0; v1 = v1 < 0.f ? 0.f : v1; v2 = v2 < 0.f ? 0.f : v2; v3 = v3 < 0.f ? 0.f : v3;
        mu = make_float4(m0, m1, m2, m3);
        rs = make_float4(rsqrtf(v0 + eps), rsqrtf(v1 + eps), rsqrtf(v2 + eps), rsqrtf(v3 + eps));
        if (blockIdx.x == 0 && rl == 0) {
            *reinterpret_cast<float4 *>(mean_out + c) = mu;
            *reinterpret_cast<float4 *>(rstd_out + c) = rs;
            if (rmean) {
                const float ub = M > 1 ? (float)M / (float)(M - 1) : 1.f;
                const float4 rm = *reinterpret_cast<const float4 *>(rmean + c), rv = *reinterpret_cast<const float4 *>(rvar + c);
                *reinterpret_cast<float4 *>(rmean + c) = make_float4((1.f - momentum) * rm.x + momentum * m0, (1.f - momentum) * rm.y + momentum * m1,
                                                                      (1.f - momentum) * rm.z + momentum * m2, (1.f - momentum) * rm.w + momentum * m3);
                *reinterpret_cast<float4 *>(rvar + c) = make_float4((1.f - momentum) * rv.x + momentum * (v0 * ub), (1.f - momentum) * rv.y + momentum * (v1 * ub),
                                                                     (1.f - momentum) * rv.z + momentum * (v2 * ub), (1.f - momentum) * rv.w + momentum * (v3 * ub));
            }
        }
    }
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + c), be = *reinterpret_cast<const float4 *>(beta + c);
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    unsigned am = 0u;
    for (int r = r0 + rl; r < r1; r += 16) {
        const size_t i = ((size_t)r * C + c) / 4;
        const float4 xv = reinterpret_cast<const float4 *>(x)[i];
        float4 o;
        o.x = fmaf((xv.x - mu.x) * rs.x, ga.x, be.x);
        o.y = fmaf((xv.y - mu.y) * rs.y, ga.y, be.y);
        o.z = fmaf((xv.z - mu.z) * rs.z, ga.z, be.z);
        o.w = fmaf((xv.w - mu.w) * rs.w, ga.w, be.w);
        if (res) {
            const float4 rv = reinterpret_cast<const float4 *>(res)[i];
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
        }
        if (mask) mask[i] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
        o.x = act_fwd(o.x, act); o.y = act_fwd(o.y, act); o.z = act_fwd(o.z, act); o.w = act_fwd(o.w, act);
        reinterpret_cast<float4 *>(y)[i] = o;
        am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
    }
    amax_commit(am, amax_y);
}

// backward: dgamma / dbeta from the partial rows (sum dz, sum dz * xhat), then dx = gamma rstd (dz - mean(dz) - xhat mean(dz xhat)), dres = dz.
// Workgroups with blockIdx.y >= nchunks carry a pending split reduction along (as bn_finalize_bwd_kernel does).
__global__ __launch_bounds__(256) void bn_fused_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                           const float *__restrict__ part, int rows, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                           float *__restrict__ dx, float *__restrict__ dres, int M, int C, int act,
                                                           const unsigned char *__restrict__ mask, int rows_per_block, int nchunks, int nrb,
                                                           const float *__restrict__ red_part, float *__restrict__ red_out, size_t red_n, int red_splits,
                                                           unsigned *__restrict__ amax_dx, unsigned *__restrict__ amax_dres,
                                                           const float *__restrict__ x2 = nullptr, const float *__restrict__ mean2 = nullptr,
                                                           const float *__restrict__ rstd2 = nullptr, float *__restrict__ part2 = nullptr,
                                                           float *__restrict__ cpart = nullptr) {
    __shared__ float4 red[16][16][2];
    if ((int)blockIdx.y >= nchunks) {              // (workgroup-uniform) the pending sum of a weight gradient's pixel splits
        sqd::split_reduce_block(red_part, red_out, red_n, red_splits, ((int)blockIdx.y - nchunks) * nrb + (int)blockIdx.x,
                                reinterpret_cast<float4(*)[16]>(&red[0][0][0]));
        return;
    }
    const int cgl = threadIdx.x & 15, rl = threadIdx.x >> 4, c0 = blockIdx.y * 64, c = c0 + cgl * 4;
    float4 db, dg;
    chunk_sums(part, rows, C, c0, db, dg, red);
    if (blockIdx.x == 0 && rl == 0) {
        *reinterpret_cast<float4 *>(dbeta + c) = db;
        *reinterpret_cast<float4 *>(dgamma + c) = dg;
    }
    const float invM = 1.0f / (float)M;
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + c), rs = *reinterpret_cast<const float4 *>(rstd + c);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c);
    const float4 k0 = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);
    const float4 k1 = make_float4(db.x * invM, db.y * invM, db.z * invM, db.w * invM);
    const float4 k2 = make_float4(rs.x * (dg.x * invM), rs.y * (dg.y * invM), rs.z * (dg.z * invM), rs.w * (dg.w * invM));
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    unsigned am = 0u, amr = 0u;
    // x2 != NULL: the backward sums of the BatchNorm behind the residual branch, one partial row per row block (bn_apply_bwd_kernel)
    float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f), q2 = s2, mu2 = s2, rs2 = s2, cs = s2;
    if (x2) {
        mu2 = *reinterpret_cast<const float4 *>(mean2 + c);
        rs2 = *reinterpret_cast<const float4 *>(rstd2 + c);
    }
    for (int r = r0 + rl; r < r1; r += 16) {
        const size_t i = ((size_t)r * C + c) / 4;
        const float4 gy = reinterpret_cast<const float4 *>(dy)[i];
        const float4 xv = reinterpret_cast<const float4 *>(x)[i];
        const float4 da = act_bwd4(mask, y, i, act);
        float4 dz, o;
        dz.x = gy.x * da.x; dz.y = gy.y * da.y; dz.z = gy.z * da.z; dz.w = gy.w * da.w;
        o.x = k0.x * (dz.x - k1.x - (xv.x - mu.x) * k2.x);
        o.y = k0.y * (dz.y - k1.y - (xv.y - mu.y) * k2.y);
        o.z = k0.z * (dz.z - k1.z - (xv.z - mu.z) * k2.z);
        o.w = k0.w * (dz.w - k1.w - (xv.w - mu.w) * k2.w);
        reinterpret_cast<float4 *>(dx)[i] = o;
        if (dres) reinterpret_cast<float4 *>(dres)[i] = dz;
        if (x2) {
            const float4 xw = reinterpret_cast<const float4 *>(x2)[i];
            s2.x += dz.x; s2.y += dz.y; s2.z += dz.z; s2.w += dz.w;
            q2.x = fmaf(dz.x, (xw.x - mu2.x) * rs2.x, q2.x); q2.y = fmaf(dz.y, (xw.y - mu2.y) * rs2.y, q2.y);
            q2.z = fmaf(dz.z, (xw.z - mu2.z) * rs2.z, q2.z); q2.w = fmaf(dz.w, (xw.w - mu2.w) * rs2.w, q2.w);
        }
        cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
        am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
        amr = max(max(amr, abs_bits(dz.x)), max(abs_bits(dz.y), max(abs_bits(dz.z), abs_bits(dz.w))));
    }
    if (cpart) {   // (uniform) column sums of dx, one partial row per row block (bn_apply_bwd_kernel)
        __syncthreads();
        red[rl][cgl][0] = cs;
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                const float4 u = red[k][cgl][0];
                cs.x += u.x; cs.y += u.y; cs.z += u.z; cs.w += u.w;
            }
            *reinterpret_cast<float4 *>(cpart + (size_t)blockIdx.x * C + c) = cs;
        }
    }
    if (x2) {      // (uniform; chunk_sums is done with `red`: every thread has passed its barrier and read its sums)
        __syncthreads();
        red[rl][cgl][0] = s2;
        red[rl][cgl][1] = q2;
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                const float4 u = red[k][cgl][0], v = red[k][cgl][1];
                s2.x += u.x; s2.y += u.y; s2.z += u.z; s2.w += u.w;
                q2.x += v.x; q2.y += v.y; q2.z += v.z; q2.w += v.w;
            }
            float *o = part2 + ((size_t)blockIdx.x * C + c) * 2;
            reinterpret_cast<float4 *>(o)[0] = make_float4(s2.x, q2.x, s2.y, q2.y);
            reinterpret_cast<float4 *>(o)[1] = make_float4(s2.z, q2.z, s2.w, q2.w);
        }
    }
    amax_commit(am, amax_dx);
    if (dres) amax_commit(amr, amax_dres, 1);
}
// rows a workgroup of the fused kernels owns: ~1024 workgroups per launch, at least 64 rows each and at least as many as there are
// partial rows (the prologue — `prows` x 512 bytes — is paid per workgroup: it stays below the workgroup's own slice of the tensor)
inline int fused_rows_per_block(int M, int C, int prows) {
    const int chunks = C / 64;
    int nrb = (1024 + chunks - 1) / chunks;
    int rpb = (M + nrb - 1) / nrb;
    rpb = rpb < 64 ? 64 : rpb;
    rpb = rpb < prows ? prows : rpb;
    return (rpb + 15) / 16 * 16;
}

int check(const char *who, int M, int C) {
    SQD_CHECK_ARG(M > 0 && C >= 4 && C % 4 == 0, "%s: need C %% 4 == 0 (C=%d, M=%d)", who, C, M);
    return SQD_OK;
}
int ew_grid(size_t total4) {
    size_t b = (total4 + 255) / 256;
    return (int)(b > 4096 ? 4096 : b);
}
}  // namespace

extern "C" int sqd_bn_nblk(int M, int C) {
    if (C < 4 || C % 4) return -1;
    return geom(M, C).nblk;
}

extern "C" int sqd_bn_train_fwd(const float *x, const float *res, const float *gamma, const float *beta, float *running_mean,
                                float *running_var, float *y, unsigned char *mask, float *save_mean, float *save_rstd, float *part,
                                int pre_rows, int M, int C, float eps, float momentum, int act, void *stream) {
    return sqd_bn_train_fwd_pool(x, res, gamma, beta, running_mean, running_var, y, mask, save_mean, save_rstd, part, pre_rows, M, C, eps, momentum,
                                 act, nullptr, 0, stream);
}

// pool_part != NULL (B images of M / B pixels, no residual): the element-wise pass also writes the per-image channel sums of y,
// pool_part [B][sqd_se_chunks(M / B)][C] — what sqd_se_pool(y) would compute, for sqd_se_gate_fwd
extern "C" int sqd_bn_train_fwd_pool(const float *x, const float *res, const float *gamma, const float *beta, float *running_mean,
                                     float *running_var, float *y, unsigned char *mask, float *save_mean, float *save_rstd, float *part,
                                     int pre_rows, int M, int C, float eps, float momentum, int act, float *pool_part, int B, void *stream) {
    return sqd_bn_train_fwd_amax(x, res, gamma, beta, running_mean, running_var, y, mask, save_mean, save_rstd, part, pre_rows, M, C, eps, momentum,
                                 act, pool_part, B, nullptr, stream);
}

// ... and amax_y (may be NULL; cleared by the caller on the stream before the call): the element-wise pass records the bit pattern of
// max |y| there — the operand scale of a convolution that reads y on two-term fp16 operands (sqd_conv_fwd_scaled).  Not with pool_part.
extern "C" int sqd_bn_train_fwd_amax(const float *x, const float *res, const float *gamma, const float *beta, float *running_mean,
                                     float *running_var, float *y, unsigned char *mask, float *save_mean, float *save_rstd, float *part,
                                     int pre_rows, int M, int C, float eps, float momentum, int act, float *pool_part, int B, float *amax_y,
                                     void *stream) {
    SQD_CHECK_ARG(!(amax_y && pool_part), "sqd_bn_train_fwd_amax: the pooled variant records no max |y|");
    SQD_CHECK_ARG(x && gamma && beta && y && save_mean && save_rstd && part, "sqd_bn_train_fwd: null pointer");
    SQD_CHECK_ARG(!pool_part || (B > 0 && M % B == 0 && !res), "sqd_bn_train_fwd_pool: B=%d must divide M=%d; no residual", B, M);
    SQD_CHECK_ARG(act != ACT_SWISH || !res, "sqd_bn_train_fwd: swish takes no residual (its backward recomputes the pre-activation from x)");
    if (check("sqd_bn_train_fwd", M, C)) return SQD_EINVAL;
    const Geom g = geom(M, C);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    // pre_rows > 0: `part` already holds that many rows of (sum, sum of squares) partials — written by the producing
    // convolution's epilogue (sqd_conv_fwd's stats) — and the reduction pass over x is skipped
    if (pre_rows > 0 && pre_rows <= FUSE_MAX_ROWS && C % 64 == 0 && !pool_part && act != ACT_SWISH) {
        // few partial rows from the producing convolution: finalize + element-wise pass in one launch
        const int rpb = fused_rows_per_block(M, C, pre_rows);
        hipLaunchKernelGGL(bn_fused_fwd_kernel, dim3((M + rpb - 1) / rpb, C / 64), dim3(256), 0, s, x, res, gamma, beta, part, pre_rows, save_mean, save_rstd,
                           running_mean, running_var, y, mask, M, C, eps, momentum, act, rpb, (unsigned *)amax_y);
        SQD_CHECK_LAUNCH("sqd_bn_train_fwd");
        return SQD_OK;
    }
    if (pre_rows <= 0)
        hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(g.nblk), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, part, M, C, act, g,
                           (const unsigned char *)nullptr, (const float *)nullptr, (const float *)nullptr);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(256), 0, s, part, pre_rows > 0 ? pre_rows : g.nblk, M,
                       C, eps, momentum, save_mean, save_rstd, running_mean, running_var);
    const size_t total4 = (size_t)M * C / 4;
    if (pool_part) {
        const int HW = M / B, nchunk = sqd_se_chunks(HW), ppc = (HW + nchunk - 1) / nchunk;
        hipLaunchKernelGGL(bn_apply_pool_fwd_kernel, dim3(nchunk, (C / 4 + 15) / 16, B), dim3(256), 0, s, x, gamma, beta, save_mean, save_rstd, y, mask,
                           pool_part, HW, C, act, ppc, nchunk);
    } else {
        hipLaunchKernelGGL((bn_apply_fwd_kernel<false>), dim3(ew_grid(total4)), dim3(256), 0, s, x, res, gamma, beta, save_mean,
                           save_rstd, y, total4, C, eps, act, mask, (unsigned *)amax_y);
    }
    SQD_CHECK_LAUNCH("sqd_bn_train_fwd");
    return SQD_OK;
}

extern "C" int sqd_bn_eval_fwd(const float *x, const float *res, const float *gamma, const float *beta,
                               const float *running_mean, const float *running_var, float *y, int M, int C, float eps, int act,
                               void *stream) {
    SQD_CHECK_ARG(x && gamma && beta && running_mean && running_var && y, "sqd_bn_eval_fwd: null pointer");
    if (check("sqd_bn_eval_fwd", M, C)) return SQD_EINVAL;
    const size_t total4 = (size_t)M * C / 4;
    (void)hipGetLastError();
    hipLaunchKernelGGL((bn_apply_fwd_kernel<true>), dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, x, res, gamma,
                       beta, running_mean, running_var, y, total4, C, eps, act, (unsigned char *)nullptr, (unsigned *)nullptr);
    SQD_CHECK_LAUNCH("sqd_bn_eval_fwd");
    return SQD_OK;
}

extern "C" int sqd_bn_train_bwd(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                                const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                                float *dbeta, float *part, int M, int C, int act, void *stream) {
    return sqd_bn_train_bwd_pre(dy, x, y, mask, gamma, beta, save_mean, save_rstd, dx, dres, dgamma, dbeta, part, 0, M, C, act, stream);
}

// pre_rows > 0: `part` already holds that many rows of (sum dz, sum dz * xhat) partials — written by the epilogue of the data gradient
// that produced dy (sqd_conv_dgrad_bn) — and the reduction pass over dy and x is skipped
extern "C" int sqd_bn_train_bwd_pre(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                                    const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                                    float *dbeta, float *part, int pre_rows, int M, int C, int act, void *stream) {
    return sqd_bn_train_bwd_pre_red(dy, x, y, mask, gamma, beta, save_mean, save_rstd, dx, dres, dgamma, dbeta, part, pre_rows, M, C, act, nullptr,
                                    nullptr, 0, 0, stream);
}

// ... and red_out[i] = sum_{s < red_splits} red_part[s * red_n + i] (sqd_split_reduce's arithmetic) as extra workgroups of the finalize
// launch: the pending sum of a weight gradient's pixel splits (sqd_conv_wgrad_partials) rides along.  red_part == NULL: none.
extern "C" int sqd_bn_train_bwd_pre_red(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                                        const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                                        float *dbeta, float *part, int pre_rows, int M, int C, int act, const float *red_part, float *red_out,
                                        int64_t red_n, int red_splits, void *stream) {
    return sqd_bn_train_bwd_amax(dy, x, y, mask, gamma, beta, save_mean, save_rstd, dx, dres, dgamma, dbeta, part, pre_rows, M, C, act, red_part, red_out,
                                 red_n, red_splits, nullptr, nullptr, stream);
}

// ... and amax_dx / amax_dres (may be NULL; cleared by the caller): bit patterns of max |dx| and max |dres|, for the data / weight gradients
// that read them on two-term fp16 operands
extern "C" int sqd_bn_train_bwd_amax(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                                     const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                                     float *dbeta, float *part, int pre_rows, int M, int C, int act, const float *red_part, float *red_out,
                                     int64_t red_n, int red_splits, float *amax_dx, float *amax_dres, void *stream) {
    return sqd_bn_train_bwd_res(dy, x, y, mask, gamma, beta, save_mean, save_rstd, dx, dres, dgamma, dbeta, part, pre_rows, M, C, act, red_part, red_out,
                                red_n, red_splits, amax_dx, amax_dres, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

// the partial rows sqd_bn_train_bwd_res writes — for the residual branch's BatchNorm and for the column sums of dx — at this shape (0: this shape
// does not take them)
extern "C" int sqd_bn_bwd_res_rows(int M, int C, int pre_rows, int act) {
    if (M <= 0 || C < 4 || C % 4 || act == ACT_SWISH) return 0;
    if (pre_rows > 0 && pre_rows <= FUSE_MAX_ROWS && C % 64 == 0) {
        const int rpb = fused_rows_per_block(M, C, pre_rows);
        return (M + rpb - 1) / rpb;
    }
    if (256 % (C / 4)) return 0;                        // (the element-wise kernel's threads keep their channel group)
    const int g = ew_grid((size_t)M * C / 4);
    return g < 2048 ? g : 2048;
}

// ... and, with x2 / mean2 / rstd2 / part2 (all or none; dres required), the two backward sums of the BatchNorm that produced the residual branch
// (no activation, this node its only consumer: dres IS its incoming gradient): x2 [M,C] that BatchNorm's input, mean2 / rstd2 its saved
// statistics -> part2 [sqd_bn_bwd_res_rows(M, C, pre_rows, act)][C][2] = (sum dres, sum dres * xhat2), to be handed to its own backward as
// precomputed partials (sqd_bn_train_bwd_pre) — the pass over dres and x2 that would take them is not run.
// dx_colsum_part (may be NULL, independent of x2): [sqd_bn_bwd_res_rows(...)][C] partial column sums of dx — summed over the rows they are the bias
// gradient of the convolution that produced x
extern "C" int sqd_bn_train_bwd_res(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                                    const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                                    float *dbeta, float *part, int pre_rows, int M, int C, int act, const float *red_part, float *red_out,
                                    int64_t red_n, int red_splits, float *amax_dx, float *amax_dres, const float *x2, const float *mean2,
                                    const float *rstd2, float *part2, float *dx_colsum_part, void *stream) {
    SQD_CHECK_ARG((!x2 && !mean2 && !rstd2 && !part2) || (x2 && mean2 && rstd2 && part2 && dres && sqd_bn_bwd_res_rows(M, C, pre_rows, act) > 0),
                  "sqd_bn_train_bwd_res: the residual branch's sums need x2, mean2, rstd2, part2, dres and a shape sqd_bn_bwd_res_rows accepts");
    SQD_CHECK_ARG(!dx_colsum_part || sqd_bn_bwd_res_rows(M, C, pre_rows, act) > 0, "sqd_bn_train_bwd_res: dx_colsum_part at a shape sqd_bn_bwd_res_rows does not accept");
    SQD_CHECK_ARG(!red_part || (red_out && red_n > 0 && red_n % 4 == 0 && red_splits >= 1), "sqd_bn_train_bwd_pre_red: bad pending reduction");
    SQD_CHECK_ARG(dy && x && gamma && save_mean && save_rstd && dx && dgamma && dbeta && part, "sqd_bn_train_bwd: null pointer");
    SQD_CHECK_ARG(pre_rows >= 0 && (pre_rows == 0 || act != ACT_SWISH), "sqd_bn_train_bwd_pre: pre_rows=%d (no precomputed partials with swish)", pre_rows);
    SQD_CHECK_ARG(act == ACT_NONE || act == ACT_SWISH || y || mask, "sqd_bn_train_bwd: ReLU / LeakyReLU need y or the sign mask of the forward");
    SQD_CHECK_ARG(act != ACT_SWISH || (beta && !dres), "sqd_bn_train_bwd: swish needs beta (pre-activation is recomputed) and takes no residual");
    if (check("sqd_bn_train_bwd", M, C)) return SQD_EINVAL;
    const Geom g = geom(M, C);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    if (pre_rows > 0 && pre_rows <= FUSE_MAX_ROWS && C % 64 == 0 && act != ACT_SWISH) {
        const int rpb = fused_rows_per_block(M, C, pre_rows), nrb = (M + rpb - 1) / rpb, nchunks = C / 64;
        const int nredb = red_part ? (int)((red_n / 4 + 15) / 16) : 0;             // blocks of the pending split reduction, nrb per grid row
        hipLaunchKernelGGL(bn_fused_bwd_kernel, dim3(nrb, nchunks + (nredb + nrb - 1) / nrb), dim3(256), 0, s, dy, x, y, gamma, save_mean, save_rstd, part,
                           pre_rows, dgamma, dbeta, dx, dres, M, C, act, mask, rpb, nchunks, nrb, red_part, red_out, (size_t)red_n, red_splits,
                           (unsigned *)amax_dx, (unsigned *)amax_dres, x2, mean2, rstd2, part2, dx_colsum_part);
        SQD_CHECK_LAUNCH("sqd_bn_train_bwd");
        return SQD_OK;
    }
    if (pre_rows <= 0)
        hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(g.nblk), dim3(256), 0, s, x, dy, y, save_mean, save_rstd, part, M, C, act, g, mask, gamma, beta);
    const int nfin = (C + FIN_CH - 1) / FIN_CH, nred = red_part ? (int)((red_n / 4 + 15) / 16) : 0;
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(nfin + nred), dim3(256), 0, s, part, pre_rows > 0 ? pre_rows : g.nblk, M, C, dgamma, dbeta, nfin,
                       red_part, red_out, (size_t)red_n, red_splits);
    const size_t total4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(x2 || dx_colsum_part ? sqd_bn_bwd_res_rows(M, C, pre_rows, act) : ew_grid(total4)), dim3(256), 0, s, dy, x, y, gamma,
                       save_mean, save_rstd, dgamma, dbeta, dx, dres, total4, C, 1.0f / (float)M, act, mask, beta, (unsigned *)amax_dx, (unsigned *)amax_dres, x2, mean2,
                       rstd2, part2, dx_colsum_part);
    SQD_CHECK_LAUNCH("sqd_bn_train_bwd");
    return SQD_OK;
}
