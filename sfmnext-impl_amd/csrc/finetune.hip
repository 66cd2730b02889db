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
// One wave finds, in a histogram of nb bins (nb a multiple of 64), the bin holding the element of rank r: -> (bin, rank inside the bin,
// count of the bin) in out[0..2].  Lane l owns nb/64 consecutive bins; an inclusive scan over the lanes locates the owner.
__device__ void wave_find_bin(const unsigned *hist, int nb, unsigned r, unsigned *out) {
    const int lane = threadIdx.x & 63, per = nb >> 6;
    unsigned s = 0;
    for (int i = 0; i < per; ++i) s += hist[lane * per + i];
    unsigned incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const unsigned long long owners = __ballot(incl > r);
    const int owner = __ffsll((long long)owners) - 1;            // r < total count, so some lane owns it
    if (lane == owner) {
        unsigned rr = r - (incl - s);
        int bin = lane * per;
        for (;; ++bin) {
            const unsigned h = hist[bin];
            if (rr < h) break;
            rr -= h;
        }
        out[0] = (unsigned)bin;
        out[1] = rr;
        out[2] = hist[bin];
    }
}

// one workgroup per sample b < nscale: ratio[b] = median(depth[valid]) / median(pred[valid]) (1 when nothing is valid).  Exact
// order statistics by a 3-level radix select (11 + 11 + 10 bits) on the float bit patterns — both arrays are positive, so the bit
// patterns order as the values do — of the ground truth and of the prediction at once (two histograms per pass over the crop
// rectangle); for an even count the upper middle element is the lower one again when its value repeats, else the smallest value
// above it (one more pass).
// the two medians (ground truth, prediction) over the valid pixels of the crop rectangle; false when nothing is valid.  Block-wide
// (NT threads); uses hist / found / nextkey / cnt_red in LDS; n_out = number of valid pixels
__device__ bool joint_medians(const float *__restrict__ p, const float *__restrict__ d, int W, float lo, float hi, const Crop &c,
                              unsigned (*hist)[BINS], unsigned (*found)[3], unsigned *nextkey, int *cnt_red, float &med_gt, float &med_pred,
                              int &n_out) {
    const int cw = c.x1 - c.x0, npx = (c.y1 - c.y0) * cw;
    int cnt = 0;
    for (int q = threadIdx.x; q < npx; q += NT) {
        const float dv = d[(size_t)(c.y0 + q / cw) * W + c.x0 + q % cw];
        cnt += (dv > lo && dv < hi) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) cnt_red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    int n = 0;
    for (int i = 0; i < NT / 64; ++i) n += cnt_red[i];
    n_out = n;
    if (n == 0) return false;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    unsigned prefix[2] = {0, 0}, rank[2] = {(unsigned)((n - 1) / 2), (unsigned)((n - 1) / 2)};
    int decided = 0;
    for (int lvl = 0; lvl < 3; ++lvl) {
        for (int i = threadIdx.x; i < 2 * BINS; i += NT) (&hist[0][0])[i] = 0;
        __syncthreads();
        const unsigned mask = (1u << widths[lvl]) - 1;
        for (int q = threadIdx.x; q < npx; q += NT) {
            const size_t at = (size_t)(c.y0 + q / cw) * W + c.x0 + q % cw;
            const float dv = d[at];
            if (!(dv > lo && dv < hi)) continue;
            const unsigned kd = __float_as_uint(dv), kp = __float_as_uint(p[at]);
            if (!decided || (kd >> (32 - decided)) == prefix[0]) atomicAdd(&hist[0][(kd >> shifts[lvl]) & mask], 1u);
            if (!decided || (kp >> (32 - decided)) == prefix[1]) atomicAdd(&hist[1][(kp >> shifts[lvl]) & mask], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) wave_find_bin(hist[0], 1 << widths[lvl], rank[0], found[0]);
        else if (threadIdx.x < 128) wave_find_bin(hist[1], 1 << widths[lvl], rank[1], found[1]);
        __syncthreads();
        for (int a = 0; a < 2; ++a) {
            prefix[a] = (prefix[a] << widths[lvl]) | found[a][0];
            rank[a] = found[a][1];
        }
        decided += widths[lvl];
    }
    // prefix[] = the elements of rank (n-1)/2; found[a][1..2] = that rank inside the run of equal values and the run's length
    unsigned upper[2] = {prefix[0], prefix[1]};
    const bool need[2] = {(n & 1) == 0 && found[0][1] + 1 >= found[0][2], (n & 1) == 0 && found[1][1] + 1 >= found[1][2]};
    if (need[0] || need[1]) {                           // block-uniform
        if (threadIdx.x < 2) nextkey[threadIdx.x] = 0xffffffffu;
        __syncthreads();
        unsigned best[2] = {0xffffffffu, 0xffffffffu};
        for (int q = threadIdx.x; q < npx; q += NT) {
            const size_t at = (size_t)(c.y0 + q / cw) * W + c.x0 + q % cw;
            const float dv = d[at];
            if (!(dv > lo && dv < hi)) continue;
            const unsigned kd = __float_as_uint(dv), kp = __float_as_uint(p[at]);
            if (kd > prefix[0] && kd < best[0]) best[0] = kd;
            if (kp > prefix[1] && kp < best[1]) best[1] = kp;
        }
        for (int a = 0; a < 2; ++a) {
            for (int o = 32; o > 0; o >>= 1) best[a] = min(best[a], (unsigned)__shfl_down(best[a], o, 64));
            if ((threadIdx.x & 63) == 0) atomicMin(&nextkey[a], best[a]);
        }
        __syncthreads();
        for (int a = 0; a < 2; ++a)
            if (need[a]) upper[a] = nextkey[a];
    }
    // np.median(float32): float32 mean of the middle pair
    med_gt = (__uint_as_float(prefix[0]) + __uint_as_float(upper[0])) * 0.5f;
    med_pred = (__uint_as_float(prefix[1]) + __uint_as_float(upper[1])) * 0.5f;
    return true;
}

__global__ __launch_bounds__(NT) void median_ratio_kernel(const float *__restrict__ pred, const float *__restrict__ depth,
                                                          float *__restrict__ ratio, int H, int W, float lo, float hi, Crop c) {
    __shared__ unsigned hist[2][BINS];
    __shared__ unsigned found[2][3];
    __shared__ unsigned nextkey[2];
    __shared__ int cnt_red[NT / 64];
    const int b = blockIdx.x;
    float mg, mp;
    int n;
    const bool any = joint_medians(pred + (size_t)b * H * W, depth + (size_t)b * H * W, W, lo, hi, c, hist, found, nextkey, cnt_red, mg, mp, n);
    if (threadIdx.x == 0) ratio[b] = any ? mg / mp : 1.f;      // np.median of an empty array is NaN -> ratio = 1 (train_ft_SQLdepth.py:261-262)
}

// validation metrics of one image (train_ft_SQLdepth.py:347-375 + utils.py:76-96): median-scaled, clamped prediction against the ground
// truth over the valid pixels of the crop; the per-pixel terms in float32 as numpy evaluates them on float32 arrays, their sums in
// float64.  out[b] = a1, a2, a3, abs_rel, rmse, log_10, rmse_log, silog, sq_rel, ratio, valid pixels (metrics NaN when none)
template <bool SCALE>
__global__ __launch_bounds__(NT) void metric_eval_kernel(const float *__restrict__ pred, const float *__restrict__ depth,
                                                         double *__restrict__ out, int H, int W, float lo, float hi, Crop c) {
    __shared__ unsigned hist[2][BINS];
    __shared__ unsigned found[2][3];
    __shared__ unsigned nextkey[2];
    __shared__ int cnt_red[NT / 64];
    __shared__ double red[9][NT / 64];
    const int b = blockIdx.x;
    const float *p = pred + (size_t)b * H * W, *d = depth + (size_t)b * H * W;
    double *o = out + (size_t)b * 11;
    float mg, mp;
    int n;
    if (!joint_medians(p, d, W, lo, hi, c, hist, found, nextkey, cnt_red, mg, mp, n)) {
        if (threadIdx.x < 11) o[threadIdx.x] = threadIdx.x == 10 ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    // SCALE: validate() (median scaling, clamps); else evaluate_metric_depth.py's eval(): the prediction as it is, only inf / nan replaced
    const float ratio = SCALE ? mg / mp : 1.f;
    const int cw = c.x1 - c.x0, npx = (c.y1 - c.y0) * cw;
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float t1 = 1.25f, t2 = 1.25f * 1.25f, t3 = 1.25f * 1.25f * 1.25f;
    for (int q = threadIdx.x; q < npx; q += NT) {
        const size_t at = (size_t)(c.y0 + q / cw) * W + c.x0 + q % cw;
        const float g = d[at];
        if (!(g > lo && g < hi)) continue;
        float v = SCALE ? p[at] * ratio : p[at];           // pred *= ratio; clamps; inf -> max, nan -> min  (:370-374)
        if (SCALE) {
            v = v < lo ? lo : v;
            v = v > hi ? hi : v;
        }
        if (isinf(v)) v = hi;
        if (isnan(v)) v = lo;
        const float th = fmaxf(g / v, v / g), df = g - v;
        const float lg = logf(g) - logf(v), l10 = fabsf(log10f(g) - log10f(v));
        s[0] += th < t1 ? 1.0 : 0.0; s[1] += th < t2 ? 1.0 : 0.0; s[2] += th < t3 ? 1.0 : 0.0;
        s[3] += (double)(fabsf(df) / g);
        s[4] += (double)(df * df);
        s[5] += (double)l10;
        s[6] += (double)(lg * lg);
        s[7] += (double)(-lg);                             // err = log(pred) - log(gt)
        s[8] += (double)(df * df / g);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double v = s[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[9];
        for (int k = 0; k < 9; ++k) {
            r[k] = 0.0;
            for (int w = 0; w < NT / 64; ++w) r[k] += red[k][w];
        }
        const double dn = (double)n, me = r[7] / dn;
        o[0] = r[0] / dn; o[1] = r[1] / dn; o[2] = r[2] / dn; o[3] = r[3] / dn; o[4] = sqrt(r[4] / dn); o[5] = r[5] / dn;
        o[6] = sqrt(r[6] / dn); o[7] = sqrt(fmax(r[6] / dn - me * me, 0.0)) * 100.0; o[8] = r[8] / dn;
        o[9] = (double)ratio; o[10] = dn;
    }
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
__global__ __launch_bounds__(256) void silog_finish_kernel(const double *__restrict__ part, int nblk, float *__restrict__ stats) {
    __shared__ double red[3][256];
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { n += part[3 * i]; s1 += part[3 * i + 1]; s2 += part[3 * i + 2]; }
    red[0][threadIdx.x] = n; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {                 // fixed-order tree: the result does not depend on scheduling
        if ((int)threadIdx.x < o)
            for (int a = 0; a < 3; ++a) red[a][threadIdx.x] += red[a][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    n = red[0][0]; s1 = red[1][0]; s2 = red[2][0];
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
// pred, depth [B,H,W] float32 (the prediction already at the ground truth's size) -> out [B][11] doubles: a1, a2, a3, abs_rel, rmse, log_10,
// rmse_log, silog, sq_rel, median ratio, valid pixels — the per-image body of the reference's validate() (train_ft_SQLdepth.py:347-375,
// utils.py:76-96); crop: 0 none, 1 Garg, 2 Eigen (KITTI), 3 Eigen (NYU: rows 45..470, columns 41..600).  median_scaling = 0: the body of
// evaluate_metric_depth.py:65-141 instead — the prediction unscaled and unclamped (inf -> max_eval, nan -> min_eval), ratio reported as 1
extern "C" int sqd_metric_depth_eval(const float *pred, const float *depth, double *out, int B, int H, int W, float min_eval, float max_eval,
                                     int crop, int median_scaling, void *stream) {
    SQD_CHECK_ARG(pred && depth && out && B > 0 && H > 0 && W > 0 && crop >= 0 && crop <= 3 && max_eval > min_eval, "sqd_metric_depth_eval: bad arguments");
    Crop c = {0, H, 0, W};
    if (crop == 1) c = Crop{(int)(0.40810811 * H), (int)(0.99189189 * H), (int)(0.03594771 * W), (int)(0.96405229 * W)};
    if (crop == 2) c = Crop{(int)(0.3324324 * H), (int)(0.91351351 * H), (int)(0.0359477 * W), (int)(0.96405229 * W)};
    if (crop == 3) c = Crop{45 < H ? 45 : H, 471 < H ? 471 : H, 41 < W ? 41 : W, 601 < W ? 601 : W};
    (void)hipGetLastError();
    if (median_scaling) hipLaunchKernelGGL(metric_eval_kernel<true>, dim3(B), dim3(NT), 0, (hipStream_t)stream, pred, depth, out, H, W, min_eval, max_eval, c);
    else hipLaunchKernelGGL(metric_eval_kernel<false>, dim3(B), dim3(NT), 0, (hipStream_t)stream, pred, depth, out, H, W, min_eval, max_eval, c);
    SQD_CHECK_LAUNCH("sqd_metric_depth_eval");
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
    hipLaunchKernelGGL(silog_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nblk, stats);
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
