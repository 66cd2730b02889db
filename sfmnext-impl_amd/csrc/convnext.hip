// convnext.hip — the operators of the ConvNeXt-L trunk and the U-Net decoder (reference networks/Unet.py:9-312 builds the encoder with
// timm.create_model('convnext_large', features_only=True) — third-party, restated from the public architecture) that the other paths
// do not have.  Channels-last activations seen as rows: [M = N*H*W][C], fp32.
//   sqd_ln_rows_fwd/bwd      LayerNorm over the channels of every pixel (timm LayerNorm2d / nn.LayerNorm in the block, eps 1e-6), with an
//                            optional per-channel bias added first (the bias of the depthwise convolution that feeds it)
//   sqd_gelu_fwd/bwd         exact GELU (erf) of the block's MLP
//   sqd_scale_residual_*     out = shortcut + gamma[c] * z   (layer scale + residual), and the reductions of its backward
//   sqd_upsample2x_fwd/bwd   F.interpolate(scale_factor=2, mode='bilinear') (align_corners=False) of the decoder block without skip
//                            (Unet.py:250)
// All HBM-bound element-wise / row work: float4 accesses, lanes along the channel axis, fixed-order reductions (deterministic).
#include "sqd_common.h"

namespace {
using namespace sqd;
constexpr int LN_MAXQ = 8;            // float4 per lane: C <= 64 * 4 * 8 = 2048
constexpr int LN_RPW = 16;            // rows per wave in the backward, at most (partial column sums stay in registers that long)
// ... fewer where the tensor has few rows: stage 3 of ConvNeXt-L is 27 of the 36 blocks at 5120 rows — 80 workgroups of 64 rows each
// left two thirds of the chip idle (65 us per call where the bytes move in 10).  Aim at >= 1024 workgroups.
__host__ __device__ inline int ln_rpw(int M) {
    int r = M / (4 * 1024);
    return r < 2 ? 2 : r > LN_RPW ? LN_RPW : r;
}

// one wave per row; lane holds float4 q = lane + 64 j of the row
__global__ __launch_bounds__(256) void ln_rows_fwd_kernel(const float *__restrict__ x, const float *__restrict__ pre_bias,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float *__restrict__ y, float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                          int M, int C, float eps, unsigned *__restrict__ amax_y) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, Q = C / 4;
    unsigned am = 0u;                                      // max |y| of this thread's elements (amax_y == NULL: not recorded)
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        float4 v[LN_MAXQ];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXQ; ++j) {
            const int q = lane + 64 * j;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < Q) {
                v[j] = reinterpret_cast<const float4 *>(x + (size_t)row * C)[q];
                if (pre_bias) {
                    const float4 b = reinterpret_cast<const float4 *>(pre_bias)[q];
                    v[j].x += b.x; v[j].y += b.y; v[j].z += b.z; v[j].w += b.w;
                }
                s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXQ; ++j) {
            if (lane + 64 * j < Q) {
                const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
                ss += (a * a + b * b) + (c * c + d * d);
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
        for (int j = 0; j < LN_MAXQ; ++j) {
            const int q = lane + 64 * j;
            if (q < Q) {
                const float4 g = reinterpret_cast<const float4 *>(gamma)[q], b = reinterpret_cast<const float4 *>(beta)[q];
                float4 o;
                o.x = (v[j].x - mean) * rstd * g.x + b.x; o.y = (v[j].y - mean) * rstd * g.y + b.y;
                o.z = (v[j].z - mean) * rstd * g.z + b.z; o.w = (v[j].w - mean) * rstd * g.w + b.w;
                reinterpret_cast<float4 *>(y + (size_t)row * C)[q] = o;
                am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
            }
        }
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
    amax_commit(am, amax_y);
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; partial column sums per workgroup:
// part[blk][0][C] = sum dy * xhat (dgamma), part[blk][1][C] = sum dy (dbeta), part[blk][2][C] = sum dx (gradient of the pre-bias)
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                          const float *__restrict__ pre_bias, const float *__restrict__ gamma,
                                                          const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                          float *__restrict__ dx, float *__restrict__ part, int M, int C) {
    extern __shared__ float red[];                         // [4 waves][3][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, Q = C / 4;
    float4 ag[LN_MAXQ], ab[LN_MAXQ], ax[LN_MAXQ];
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) ag[j] = ab[j] = ax[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int rpw = ln_rpw(M);
    const int row0 = (blockIdx.x * 4 + wave) * rpw;
    for (int r = 0; r < rpw; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        const float mean = mean_in[row], rstd = rstd_in[row];
        float4 xh[LN_MAXQ], gv[LN_MAXQ];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXQ; ++j) {
            const int q = lane + 64 * j;
            xh[j] = gv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < Q) {
                float4 xv = reinterpret_cast<const float4 *>(x + (size_t)row * C)[q];
                if (pre_bias) {
                    const float4 b = reinterpret_cast<const float4 *>(pre_bias)[q];
                    xv.x += b.x; xv.y += b.y; xv.z += b.z; xv.w += b.w;
                }
                const float4 d = reinterpret_cast<const float4 *>(dy + (size_t)row * C)[q];
                const float4 g = reinterpret_cast<const float4 *>(gamma)[q];
                xh[j] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                gv[j] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
                s1 += (gv[j].x + gv[j].y) + (gv[j].z + gv[j].w);
                s2 += (gv[j].x * xh[j].x + gv[j].y * xh[j].y) + (gv[j].z * xh[j].z + gv[j].w * xh[j].w);
                ag[j].x += d.x * xh[j].x; ag[j].y += d.y * xh[j].y; ag[j].z += d.z * xh[j].z; ag[j].w += d.w * xh[j].w;
                ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
            }
        }
        const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int j = 0; j < LN_MAXQ; ++j) {
            const int q = lane + 64 * j;
            if (q < Q) {
                float4 o;
                o.x = rstd * (gv[j].x - m1 - xh[j].x * m2); o.y = rstd * (gv[j].y - m1 - xh[j].y * m2);
                o.z = rstd * (gv[j].z - m1 - xh[j].z * m2); o.w = rstd * (gv[j].w - m1 - xh[j].w * m2);
                reinterpret_cast<float4 *>(dx + (size_t)row * C)[q] = o;
                ax[j].x += o.x; ax[j].y += o.y; ax[j].z += o.z; ax[j].w += o.w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < LN_MAXQ; ++j) {
        const int q = lane + 64 * j;
        if (q < Q) {
            reinterpret_cast<float4 *>(red + ((size_t)wave * 3 + 0) * C)[q] = ag[j];
            reinterpret_cast<float4 *>(red + ((size_t)wave * 3 + 1) * C)[q] = ab[j];
            reinterpret_cast<float4 *>(red + ((size_t)wave * 3 + 2) * C)[q] = ax[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += 256)
        part[(size_t)blockIdx.x * 3 * C + i] = ((red[i] + red[3 * C + i]) + red[6 * C + i]) + red[9 * C + i];
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_d(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
template <bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ out, size_t n4,
                                                   unsigned *__restrict__ amax_out) {
    unsigned am = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        float4 o;
        if (BWD) {
            const float4 d = reinterpret_cast<const float4 *>(dy)[i];
            o = make_float4(d.x * gelu_d(v.x), d.y * gelu_d(v.y), d.z * gelu_d(v.z), d.w * gelu_d(v.w));
        } else {
            o = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
        }
        reinterpret_cast<float4 *>(out)[i] = o;
        am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
    }
    amax_commit(am, amax_out);
}

// MODE 0: out = res + gamma * z.   MODE 1: dz = gamma * dy, and per-workgroup partial column sums of dy * z (dgamma) -> part[blk][C]
// and (part2 != NULL) of dz -> part2[blk][C]: the bias gradient of the Linear layer whose output z is, taken by the pass that writes dz.
// MODE 2 (a = dy, z = x, no gamma): dx = dy * gelu'(x) and partial column sums of dx -> part[blk][C] (the bias gradient of the Linear
// layer in front of the GELU).
template <int MODE>
__global__ __launch_bounds__(256) void scale_residual_kernel(const float *__restrict__ a, const float *__restrict__ z,
                                                             const float *__restrict__ gamma, float *__restrict__ out,
                                                             float *__restrict__ part, int M, int C, int rows_per_blk,
                                                             unsigned *__restrict__ amax_out, float *__restrict__ part2) {
    const int Q = C / 4;
    if (MODE == 0) {
        const size_t total = (size_t)M * Q;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const float4 r = reinterpret_cast<const float4 *>(a)[i], zz = reinterpret_cast<const float4 *>(z)[i];
            const float4 g = reinterpret_cast<const float4 *>(gamma)[i % Q];
            reinterpret_cast<float4 *>(out)[i] = make_float4(r.x + g.x * zz.x, r.y + g.y * zz.y, r.z + g.z * zz.z, r.w + g.w * zz.w);
        }
        return;
    }
    // backward: a = dy.  Wave w takes rows r0 + w, r0 + w + 4, ... of the block, lane l the float4 columns l, l + 64, ...; the four
    // waves' column sums are added through LDS in wave order
    extern __shared__ float red[];                         // [4][C] (MODE 1 with part2: [8][C])
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
    const bool two = MODE == 1 && part2 != nullptr;
    unsigned am = 0u;                                      // max |out| (amax_out == NULL: not recorded)
    for (int q = lane; q < Q; q += 64) {
        const float4 g = MODE == 1 ? reinterpret_cast<const float4 *>(gamma)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = acc;
        // four rows of the wave per step: their eight loads are in flight together (rows beyond the block read row r1 - 1 again and are
        // not stored or summed); the sums run over the rows in order
        for (int row = r0 + wave; row < r1; row += 16) {
            float4 d[4], zz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = min(row + 4 * u, r1 - 1);
                d[u] = reinterpret_cast<const float4 *>(a + (size_t)rr * C)[q];
                zz[u] = reinterpret_cast<const float4 *>(z + (size_t)rr * C)[q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (row + 4 * u >= r1) break;
                float4 o;
                if (MODE == 1) o = make_float4(g.x * d[u].x, g.y * d[u].y, g.z * d[u].z, g.w * d[u].w);
                else o = make_float4(d[u].x * gelu_d(zz[u].x), d[u].y * gelu_d(zz[u].y), d[u].z * gelu_d(zz[u].z), d[u].w * gelu_d(zz[u].w));
                reinterpret_cast<float4 *>(out + (size_t)(row + 4 * u) * C)[q] = o;
                am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
                if (MODE == 1) {
                    acc.x += d[u].x * zz[u].x; acc.y += d[u].y * zz[u].y; acc.z += d[u].z * zz[u].z; acc.w += d[u].w * zz[u].w;
                    acc2.x += o.x; acc2.y += o.y; acc2.z += o.z; acc2.w += o.w;
                } else {
                    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                }
            }
        }
        reinterpret_cast<float4 *>(red + (size_t)wave * C)[q] = acc;
        if (two) reinterpret_cast<float4 *>(red + (size_t)(4 + wave) * C)[q] = acc2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
        part[(size_t)blockIdx.x * C + i] = ((red[i] + red[C + i]) + red[2 * C + i]) + red[3 * C + i];
        if (two) part2[(size_t)blockIdx.x * C + i] = ((red[4 * C + i] + red[5 * C + i]) + red[6 * C + i]) + red[7 * C + i];
    }
    amax_commit(am, amax_out);
}

// ATen upsample_bilinear2d, align_corners = False, scale 2: src = max(0.5 * (dst + 0.5) - 0.5, 0)
__device__ __forceinline__ void up2_src(int d, int n_in, int &i0, int &i1, float &l1) {
    float s = 0.5f * ((float)d + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int N, int H, int W, int C) {
    const int Q = C / 4, Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)N * Ho * Wo * Q;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % Q);
        const size_t p = i / Q;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
        int y0, y1, x0, x1;
        float ly, lx;
        up2_src(yo, H, y0, y1, ly);
        up2_src(xo, W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float4 *b = reinterpret_cast<const float4 *>(x + (size_t)n * H * W * C);
        const float4 v00 = b[((size_t)y0 * W + x0) * Q + q], v01 = b[((size_t)y0 * W + x1) * Q + q];
        const float4 v10 = b[((size_t)y1 * W + x0) * Q + q], v11 = b[((size_t)y1 * W + x1) * Q + q];
        float4 o;
        o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
        o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
        o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
        o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}
// gather form of the adjoint: input pixel (yi, xi) collects from the output rows / columns whose taps touch it (fixed order)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, int N, int H, int W, int C) {
    const int Q = C / 4, Ho = 2 * H, Wo = 2 * W;
    const size_t total = (size_t)N * H * W * Q;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i % Q);
        const size_t p = i / Q;
        const int xi = (int)(p % W), yi = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
        const float4 *g = reinterpret_cast<const float4 *>(dy + (size_t)n * Ho * Wo * C);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int yo = max(0, 2 * yi - 2); yo <= min(Ho - 1, 2 * yi + 2); ++yo) {
            int y0, y1;
            float ly;
            up2_src(yo, H, y0, y1, ly);
            const float wy = (y0 == yi ? 1.f - ly : 0.f) + (y1 == yi ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int xo = max(0, 2 * xi - 2); xo <= min(Wo - 1, 2 * xi + 2); ++xo) {
                int x0, x1;
                float lx;
                up2_src(xo, W, x0, x1, lx);
                const float wx = (x0 == xi ? 1.f - lx : 0.f) + (x1 == xi ? lx : 0.f);
                if (wx == 0.f) continue;
                const float4 d = g[((size_t)yo * Wo + xo) * Q + q];
                const float w = wy * wx;
                acc.x += w * d.x; acc.y += w * d.y; acc.z += w * d.z; acc.w += w * d.w;
            }
        }
        reinterpret_cast<float4 *>(dx)[i] = acc;
    }
}

int ew_grid(size_t n) {
    const size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 16384 ? 16384 : b);
}
int ln_check(const char *who, int M, int C) {
    SQD_CHECK_ARG(M > 0 && C >= 4 && C % 4 == 0 && C <= 64 * 4 * LN_MAXQ, "%s: rows=%d, C=%d (C a multiple of 4, <= 2048)", who, M, C);
    return SQD_OK;
}
}  // namespace

// x, y [M,C]; pre_bias [C] or NULL (added to x first); gamma, beta [C]; mean, rstd [M] (saved for the backward)
extern "C" int sqd_ln_rows_fwd(const float *x, const float *pre_bias, const float *gamma, const float *beta, float *y, float *mean,
                               float *rstd, int M, int C, float eps, void *stream) {
    return sqd_ln_rows_fwd_amax(x, pre_bias, gamma, beta, y, mean, rstd, M, C, eps, nullptr, stream);
}
// ... and amax_y (may be NULL; a record cleared by the caller on the stream before the call, include/sqd.h section 10b): the bit pattern of
// max |y| — the operand scale of the block's first Linear layer when it runs on two-term fp16 operands (no sqd_amax pass over y)
extern "C" int sqd_ln_rows_fwd_amax(const float *x, const float *pre_bias, const float *gamma, const float *beta, float *y, float *mean,
                                    float *rstd, int M, int C, float eps, float *amax_y, void *stream) {
    SQD_CHECK_ARG(x && gamma && beta && y && mean && rstd, "sqd_ln_rows_fwd: null pointer");
    if (ln_check("sqd_ln_rows_fwd", M, C)) return SQD_EINVAL;
    const int blocks = (M + 3) / 4 > 8192 ? 8192 : (M + 3) / 4;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ln_rows_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, pre_bias, gamma, beta, y, mean, rstd, M, C, eps,
                       (unsigned *)amax_y);
    SQD_CHECK_LAUNCH("sqd_ln_rows_fwd");
    return SQD_OK;
}
extern "C" int sqd_ln_rows_nblk(int M) { return (M + 4 * ln_rpw(M) - 1) / (4 * ln_rpw(M)); }
// dy, x [M,C] -> dx [M,C]; part [sqd_ln_rows_nblk(M)][3][C]: per-block column sums of dy*xhat, dy, dx (sum them over the blocks:
// dgamma, dbeta, gradient of pre_bias)
extern "C" int sqd_ln_rows_bwd(const float *dy, const float *x, const float *pre_bias, const float *gamma, const float *mean,
                               const float *rstd, float *dx, float *part, int M, int C, void *stream) {
    SQD_CHECK_ARG(dy && x && gamma && mean && rstd && dx && part, "sqd_ln_rows_bwd: null pointer");
    if (ln_check("sqd_ln_rows_bwd", M, C)) return SQD_EINVAL;
    const size_t shmem = (size_t)4 * 3 * C * sizeof(float);
    (void)hipGetLastError();
    if (shmem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ln_rows_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(ln_rows_bwd_kernel, dim3(sqd_ln_rows_nblk(M)), dim3(256), shmem, (hipStream_t)stream, dy, x, pre_bias, gamma, mean, rstd,
                       dx, part, M, C);
    SQD_CHECK_LAUNCH("sqd_ln_rows_bwd");
    return SQD_OK;
}
// exact (erf) GELU: y = gelu(x);  backward: dx = dy * gelu'(x).  n a multiple of 4, 16-byte aligned pointers
extern "C" int sqd_gelu_fwd(const float *x, float *y, int64_t n, void *stream) { return sqd_gelu_fwd_amax(x, y, n, nullptr, stream); }
extern "C" int sqd_gelu_bwd(const float *x, const float *dy, float *dx, int64_t n, void *stream) {
    return sqd_gelu_bwd_amax(x, dy, dx, n, nullptr, stream);
}
// ... and amax_y / amax_dx (may be NULL; cleared records): max |y| for the second Linear layer's forward, max |dx| for the first one's
// gradients, recorded by the pass that writes the tensor
extern "C" int sqd_gelu_fwd_amax(const float *x, float *y, int64_t n, float *amax_y, void *stream) {
    SQD_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "sqd_gelu_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL((gelu_kernel<false>), dim3(ew_grid((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, x, (const float *)nullptr, y, (size_t)n / 4,
                       (unsigned *)amax_y);
    SQD_CHECK_LAUNCH("sqd_gelu_fwd");
    return SQD_OK;
}
extern "C" int sqd_gelu_bwd_amax(const float *x, const float *dy, float *dx, int64_t n, float *amax_dx, void *stream) {
    SQD_CHECK_ARG(x && dy && dx && n > 0 && n % 4 == 0, "sqd_gelu_bwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL((gelu_kernel<true>), dim3(ew_grid((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, (size_t)n / 4, (unsigned *)amax_dx);
    SQD_CHECK_LAUNCH("sqd_gelu_bwd");
    return SQD_OK;
}
// out [M,C] = res + gamma[c] * z
extern "C" int sqd_scale_residual_fwd(const float *res, const float *z, const float *gamma, float *out, int M, int C, void *stream) {
    SQD_CHECK_ARG(res && z && gamma && out && M > 0 && C >= 4 && C % 4 == 0, "sqd_scale_residual_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL((scale_residual_kernel<0>), dim3(ew_grid((size_t)M * C / 4)), dim3(256), 0, (hipStream_t)stream, res, z, gamma, out,
                       (float *)nullptr, M, C, 0, (unsigned *)nullptr, (float *)nullptr);
    SQD_CHECK_LAUNCH("sqd_scale_residual_fwd");
    return SQD_OK;
}
// rows per workgroup of the backward: >= 1024 workgroups where the rows allow, 8 .. 256 rows each (the partial rows a launch leaves are
// summed by sqd_colsum_multi: at 64 rows the 327 680-row maps of stage 1 left 5120 of them per tensor)
static int scale_residual_rows(int M) {
    int r = M / 1024;
    r = (r + 3) & ~3;
    return r < 8 ? 8 : r > 256 ? 256 : r;
}
extern "C" int sqd_scale_residual_nblk(int M) { return (M + scale_residual_rows(M) - 1) / scale_residual_rows(M); }
// dy, z [M,C] -> dz = gamma * dy; part [sqd_scale_residual_nblk(M)][C] per-block column sums of dy * z (sum over the blocks: dgamma)
extern "C" int sqd_scale_residual_bwd(const float *dy, const float *z, const float *gamma, float *dz, float *part, int M, int C, void *stream) {
    return sqd_scale_residual_bwd_amax(dy, z, gamma, dz, part, M, C, nullptr, stream);
}
// ... and amax_dz (may be NULL; a cleared record): max |dz| for the second Linear layer's gradients
extern "C" int sqd_scale_residual_bwd_amax(const float *dy, const float *z, const float *gamma, float *dz, float *part, int M, int C, float *amax_dz,
                                           void *stream) {
    return sqd_scale_residual_bwd_sums(dy, z, gamma, dz, part, nullptr, M, C, amax_dz, stream);
}
// ... and part2 [sqd_scale_residual_nblk(M)][C] (may be NULL): per-block column sums of dz — summed over the blocks, the bias gradient of
// the Linear layer that produced z (its weight-gradient call then takes dbias = NULL: no column-sum pass over dz of its own)
extern "C" int sqd_scale_residual_bwd_sums(const float *dy, const float *z, const float *gamma, float *dz, float *part, float *part2, int M, int C,
                                           float *amax_dz, void *stream) {
    SQD_CHECK_ARG(dy && z && gamma && dz && part && M > 0 && C >= 4 && C % 4 == 0, "sqd_scale_residual_bwd: bad arguments");
    const size_t shmem = (size_t)(part2 ? 8 : 4) * C * sizeof(float);
    SQD_CHECK_ARG(shmem <= 160 * 1024, "sqd_scale_residual_bwd: C=%d needs more LDS than a workgroup has", C);
    (void)hipGetLastError();
    if (shmem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&scale_residual_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL((scale_residual_kernel<1>), dim3(sqd_scale_residual_nblk(M)), dim3(256), shmem, (hipStream_t)stream, dy, z, gamma, dz, part, M, C,
                       scale_residual_rows(M), (unsigned *)amax_dz, part2);
    SQD_CHECK_LAUNCH("sqd_scale_residual_bwd");
    return SQD_OK;
}
// GELU backward on rows: x, dy, dx [M,C]; part [sqd_scale_residual_nblk(M)][C] = per-block column sums of dx (summed over the blocks: the bias
// gradient of the Linear layer in front of the GELU); amax_dx as in sqd_gelu_bwd_amax.  The values of sqd_gelu_bwd.
extern "C" int sqd_gelu_bwd_rows(const float *x, const float *dy, float *dx, float *part, int M, int C, float *amax_dx, void *stream) {
    SQD_CHECK_ARG(x && dy && dx && part && M > 0 && C >= 4 && C % 4 == 0, "sqd_gelu_bwd_rows: bad arguments");
    const size_t shmem = (size_t)4 * C * sizeof(float);
    SQD_CHECK_ARG(shmem <= 160 * 1024, "sqd_gelu_bwd_rows: C=%d needs more LDS than a workgroup has", C);
    (void)hipGetLastError();
    if (shmem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&scale_residual_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL((scale_residual_kernel<2>), dim3(sqd_scale_residual_nblk(M)), dim3(256), shmem, (hipStream_t)stream, dy, x, (const float *)nullptr, dx,
                       part, M, C, scale_residual_rows(M), (unsigned *)amax_dx, (float *)nullptr);
    SQD_CHECK_LAUNCH("sqd_gelu_bwd_rows");
    return SQD_OK;
}
// x [N,H,W,C] -> y [N,2H,2W,C], bilinear, align_corners = False;  backward: dy [N,2H,2W,C] -> dx [N,H,W,C]
extern "C" int sqd_upsample2x_fwd(const float *x, float *y, int N, int H, int W, int C, void *stream) {
    SQD_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "sqd_upsample2x_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(ew_grid((size_t)N * 4 * H * W * C / 4)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
    SQD_CHECK_LAUNCH("sqd_upsample2x_fwd");
    return SQD_OK;
}
extern "C" int sqd_upsample2x_bwd(const float *dy, float *dx, int N, int H, int W, int C, void *stream) {
    SQD_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "sqd_upsample2x_bwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid((size_t)N * H * W * C / 4)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, C);
    SQD_CHECK_LAUNCH("sqd_upsample2x_bwd");
    return SQD_OK;
}
