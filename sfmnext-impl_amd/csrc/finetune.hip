// finetune.hip — the per-step arithmetic of the reference's supervised metric-depth finetune loop
// (finetune/train_ft_SQLdepth.py:219-285, finetune/loss.py:24-42) on the device:
//   sqd_resize_ac_fwd/bwd     nn.functional.interpolate(pred, depth.shape[-2:], mode='bilinear', align_corners=True) of the [B,1,h,w]
//                             prediction (:233) and its adjoint
//   sqd_median_ratio          the per-sample median rescale of :234-266, computed on the host with numpy in the reference:
//                             ratio = median(depth[valid]) / median(pred[valid]), valid = min_eval < depth < max_eval inside the
//                             Garg / Eigen crop — exact medians by radix select, float32 like np.median of float32 arrays
//   sqd_silog_fwd/bwd         SILogLoss (loss.py:24-42) over the pixels with depth > min_depth of the (ratio-scaled) prediction:
//                             g = log(pred) - log(depth), loss = 10 * sqrt(var(g) + 0.15 * mean(g)^2) (unbiased variance)
// One workgroup per sample for the medians, fixed-order reductions: deterministic.  Not part of the self-supervised step.
#include "sqd_common.h"

namespace {
using namespace sqd;
constexpr int NT = 1024, BINS = 2048;

__device__ __forceinline__ void ac_src(int d, int n_in, int n_out, int &i0, int &i1, float &l1) {
    // ATen area_pixel_compute_source_index, align_corners=True: src = d * (n_in - 1) / (n_out - 1)
    const float scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
    const float s = scale * (float)d;
    i0 = (int)s;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void resize_ac_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int h, int w, int H,
                                                            int W) {
    const size_t total = (size_t)B * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xo = (int)(i % W), yo = (int)((i / W) % H), b = (int)(i / ((size_t)W * H));
        int y0, y1, x0, x1;
        float ly, lx;
        ac_src(yo, h, H, y0, y1, ly);
        ac_src(xo, w, W, x0, x1, lx);
        const float *p = x + (size_t)b * h * w;
        const float hy = 1.f - ly, hx = 1.f - lx;
        y[i] = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
    }
}
// gather form of the adjoint: source pixel (ys, xs) collects from the destination rows / columns whose taps touch it
__global__ __launch_bounds__(256) void resize_ac_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ rowscale,
                                                            float *__restrict__ dx, int B, int h, int w, int H, int W) {
    const size_t total = (size_t)B * h * w;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xs = (int)(i % w), ys = (int)((i / w) % h), b = (int)(i / ((size_t)w * h));
        // destination indices d with floor(s * d) in {src - 1, src}: d in ((src - 1) / s, (src + 1) / s)
        const int ylo = sy > 0.f ? max(0, (int)floorf((float)(ys - 1) / sy)) : 0, yhi = sy > 0.f ? min(H - 1, (int)ceilf((float)(ys + 1) / sy)) : H - 1;
        const int xlo = sx > 0.f ? max(0, (int)floorf((float)(xs - 1) / sx)) : 0, xhi = sx > 0.f ? min(W - 1, (int)ceilf((float)(xs + 1) / sx)) : W - 1;
        const float *g = dy + (size_t)b * H * W;
        float acc = 0.f;
        for (int yo = ylo; yo <= yhi; ++yo) {
            int y0, y1;
            float ly;
            ac_src(yo, h, H, y0, y1, ly);
            const float wy = (y0 == ys ? 1.f - ly : 0.f) + (y1 == ys ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int xo = xlo; xo <= xhi; ++xo) {
                int x0, x1;
                float lx;
                ac_src(xo, w, W, x0, x1, lx);
                const float wx = (x0 == xs ? 1.f - lx : 0.f) + (x1 == xs ? lx : 0.f);
                if (wx != 0.f) acc += wy * wx * g[(size_t)yo * W + xo];
            }
        }
        dx[i] = acc * (rowscale ? rowscale[b] : 1.f);
    }
}

struct Crop {
    int y0, y1, x0, x1;
};
__device__ __forceinline__ bool valid_px(float d, int y, int x, float lo, float hi, const Crop &c) {
    return d > lo && d < hi && y >= c.y0 && y < c.y1 && x >= c.x0 && x < c.x1;
}
// element of rank `rank` among the valid pixels of v (WHICH: 0 depth, 1 pred): 3-level radix select on the float bit patterns (> 0)
template <int WHICH>
__device__ float select_rank_f32(const float *pred, const float *depth, int H, int W, float lo, float hi, const Crop &c, long long rank,
                                 unsigned *hist, unsigned *shared) {
    unsigned prefix = 0;
    int decided = 0;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    const int total = H * W;
    for (int lvl = 0; lvl < 3; ++lvl) {
        for (int i = threadIdx.x; i < BINS; i += NT) hist[i] = 0;
        __syncthreads();
        for (int p = threadIdx.x; p < total; p += NT) {
            const float d = depth[p];
            if (!valid_px(d, p / W, p % W, lo, hi, c)) continue;
            const unsigned key = __float_as_uint(WHICH == 0 ? d : pred[p]);
            if (decided && (key >> (32 - decided)) != prefix) continue;
            atomicAdd(&hist[(key >> shifts[lvl]) & ((1u << widths[lvl]) - 1)], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            long long r = rank;
            unsigned b = 0;
            for (; b < (1u << widths[lvl]); ++b) {
                if (r < (long long)hist[b]) break;
                r -= hist[b];
            }
            shared[0] = b;
            shared[1] = (unsigned)r;
        }
        __syncthreads();
        prefix = (prefix << widths[lvl]) | shared[0];
        rank = (long long)shared[1];
        decided += widths[lvl];
        __syncthreads();
    }
    return __uint_as_float(prefix);
}

// one workgroup per sample b < nscale: ratio[b] = median(depth[valid]) / median(pred[valid]) (1 when a median is NaN / nothing valid)
__global__ __launch_bounds__(NT) void median_ratio_kernel(const float *__restrict__ pred, const float *__restrict__ depth,
                                                          float *__restrict__ ratio, int H, int W, float lo, float hi, Crop c) {
    __shared__ unsigned hist[BINS];
    __shared__ unsigned shared[2];
    __shared__ int cnt_red[NT / 64];
    const int b = blockIdx.x;
    const float *p = pred + (size_t)b * H * W, *d = depth + (size_t)b * H * W;
    int cnt = 0;
    for (int i = threadIdx.x; i < H * W; i += NT) cnt += valid_px(d[i], i / W, i % W, lo, hi, c) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) cnt_red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    int n = 0;
    for (int i = 0; i < NT / 64; ++i) n += cnt_red[i];
    __syncthreads();
    if (n == 0) {                                       // np.median of an empty array is NaN -> ratio = 1 (train_ft_SQLdepth.py:261-262)
        if (threadIdx.x == 0) ratio[b] = 1.f;
        return;
    }
    const long long r1 = (n - 1) / 2, r2 = n / 2;
    const float g1 = select_rank_f32<0>(p, d, H, W, lo, hi, c, r1, hist, shared), g2 = r2 == r1 ? g1 : select_rank_f32<0>(p, d, H, W, lo, hi, c, r2, hist, shared);
    const float p1 = select_rank_f32<1>(p, d, H, W, lo, hi, c, r1, hist, shared), p2 = r2 == r1 ? p1 : select_rank_f32<1>(p, d, H, W, lo, hi, c, r2, hist, shared);
    if (threadIdx.x == 0) ratio[b] = ((g1 + g2) * 0.5f) / ((p1 + p2) * 0.5f);      // np.median(float32): float32 mean of the middle pair
}

// SILog partial sums over chunks: part[blk] = (n, sum g, sum g^2) in double, g = log(scale_b * pred) - log(depth) where depth > min_depth
__global__ __launch_bounds__(256) void silog_sums_kernel(const float *__restrict__ pred, const float *__restrict__ depth,
                                                         const float *__restrict__ scale, double *__restrict__ part, int HW, size_t total,
                                                         float min_depth) {
    __shared__ double red[3][4];
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float d = depth[i];
        if (d > min_depth) {
            const float g = logf(pred[i] * (scale ? scale[i / HW] : 1.f)) - logf(d);
            n += 1.0; s1 += (double)g; s2 += (double)g * (double)g;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { n += __shfl_down(n, o, 64); s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = n; red[1][threadIdx.x >> 6] = s1; red[2][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) part[(size_t)blockIdx.x * 3 + threadIdx.x] = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
}
// stats [4] = (n, mean, Dg, loss)
__global__ __launch_bounds__(64) void silog_finish_kernel(const double *__restrict__ part, int nblk, float *__restrict__ stats) {
    if (threadIdx.x != 0) return;
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < nblk; ++i) { n += part[3 * i]; s1 += part[3 * i + 1]; s2 += part[3 * i + 2]; }
    const double mean = s1 / n, var = (s2 - s1 * s1 / n) / (n - 1.0);             // torch.var: unbiased
    const double Dg = var + 0.15 * mean * mean;
    stats[0] = (float)n; stats[1] = (float)mean; stats[2] = (float)Dg; stats[3] = (float)(10.0 * sqrt(Dg));
}
// d loss / d pred (of the UN-scaled prediction: d log(r p) / dp = 1 / p) times the upstream gradient gl[0]
__global__ __launch_bounds__(256) void silog_bwd_kernel(const float *__restrict__ pred, const float *__restrict__ depth,
                                                        const float *__restrict__ scale, const float *__restrict__ stats,
                                                        const float *__restrict__ gl, float *__restrict__ dpred, int HW, size_t total,
                                                        float min_depth) {
    const float n = stats[0], mean = stats[1], Dg = stats[2];
    const float k = gl[0] * 10.f * 0.5f / sqrtf(Dg);                          // d(10 sqrt(Dg)) / dDg
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float d = depth[i];
        float o = 0.f;
        if (d > min_depth) {
            const float p = pred[i];
            const float g = logf(p * (scale ? scale[i / HW] : 1.f)) - logf(d);
            o = k * (2.f * (g - mean) / (n - 1.f) + 0.3f * mean / n) / p;
        }
        dpred[i] = o;
    }
}
int ew_grid(size_t n, int cap) {
    const size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > (size_t)cap ? (size_t)cap : b);
}
}  // namespace

// x [B,1,h,w] -> y [B,1,H,W], bilinear, align_corners = True;  backward: dy -> dx, each sample's rows multiplied by rowscale[b] (or NULL)
extern "C" int sqd_resize_ac_fwd(const float *x, float *y, int B, int h, int w, int H, int W, void *stream) {
    SQD_CHECK_ARG(x && y && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "sqd_resize_ac_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(resize_ac_fwd_kernel, dim3(ew_grid((size_t)B * H * W, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, B, h, w, H, W);
    SQD_CHECK_LAUNCH("sqd_resize_ac_fwd");
    return SQD_OK;
}
extern "C" int sqd_resize_ac_bwd(const float *dy, const float *rowscale, float *dx, int B, int h, int w, int H, int W, void *stream) {
    SQD_CHECK_ARG(dy && dx && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "sqd_resize_ac_bwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(resize_ac_bwd_kernel, dim3(ew_grid((size_t)B * h * w, 8192)), dim3(256), 0, (hipStream_t)stream, dy, rowscale, dx, B, h, w,
                       H, W);
    SQD_CHECK_LAUNCH("sqd_resize_ac_bwd");
    return SQD_OK;
}
// pred, depth [B,H,W]; ratio [nscale] for the first nscale samples; crop: 0 none, 1 Garg, 2 Eigen (KITTI) — train_ft_SQLdepth.py:239-251
extern "C" int sqd_median_ratio(const float *pred, const float *depth, float *ratio, int nscale, int H, int W, float min_eval, float max_eval,
                                int crop, void *stream) {
    SQD_CHECK_ARG(pred && depth && ratio && nscale > 0 && H > 0 && W > 0 && crop >= 0 && crop <= 2, "sqd_median_ratio: bad arguments");
    Crop c = {0, H, 0, W};
    if (crop == 1) c = Crop{(int)(0.40810811 * H), (int)(0.99189189 * H), (int)(0.03594771 * W), (int)(0.96405229 * W)};
    if (crop == 2) c = Crop{(int)(0.3324324 * H), (int)(0.91351351 * H), (int)(0.0359477 * W), (int)(0.96405229 * W)};
    (void)hipGetLastError();
    hipLaunchKernelGGL(median_ratio_kernel, dim3(nscale), dim3(NT), 0, (hipStream_t)stream, pred, depth, ratio, H, W, min_eval, max_eval, c);
    SQD_CHECK_LAUNCH("sqd_median_ratio");
    return SQD_OK;
}
extern "C" int sqd_silog_nblk(int64_t total) { return ew_grid((size_t)total, 1024); }
// pred, depth [B,H,W]; scale [B] or NULL (the median ratios; samples beyond nscale carry 1); part [3 * sqd_silog_nblk] doubles;
// stats [4] floats = (valid pixels, mean g, Dg, loss)
extern "C" int sqd_silog_fwd(const float *pred, const float *depth, const float *scale, double *part, float *stats, int B, int HW,
                             float min_depth, void *stream) {
    SQD_CHECK_ARG(pred && depth && part && stats && B > 0 && HW > 0, "sqd_silog_fwd: bad arguments");
    const size_t total = (size_t)B * HW;
    const int nblk = sqd_silog_nblk((int64_t)total);
    (void)hipGetLastError();
    hipLaunchKernelGGL(silog_sums_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, pred, depth, scale, part, HW, total, min_depth);
    hipLaunchKernelGGL(silog_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, nblk, stats);
    SQD_CHECK_LAUNCH("sqd_silog_fwd");
    return SQD_OK;
}
// g_loss: device scalar (the upstream gradient of the loss) -> dpred [B,H,W] (gradient w.r.t. the un-scaled prediction)
extern "C" int sqd_silog_bwd(const float *pred, const float *depth, const float *scale, const float *stats, const float *g_loss, float *dpred,
                             int B, int HW, float min_depth, void *stream) {
    SQD_CHECK_ARG(pred && depth && stats && g_loss && dpred && B > 0 && HW > 0, "sqd_silog_bwd: bad arguments");
    const size_t total = (size_t)B * HW;
    (void)hipGetLastError();
    hipLaunchKernelGGL(silog_bwd_kernel, dim3(ew_grid(total, 8192)), dim3(256), 0, (hipStream_t)stream, pred, depth, scale, stats, g_loss, dpred,
                       HW, total, min_depth);
    SQD_CHECK_LAUNCH("sqd_silog_bwd");
    return SQD_OK;
}
