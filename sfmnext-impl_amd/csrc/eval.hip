// eval.hip — the depth-evaluation arithmetic of the reference (evaluate_depth_config.py) on the device, in double precision:
//   sqd_disp_post_process : batch_post_process_disparity (evaluate_depth_config.py:50-59) incl. the flip of the second pass
//   sqd_depth_eval        : the per-image body of evaluate() (:225-261) — bilinear resize of the prediction to the ground truth's
//                           size (cv2.resize, INTER_LINEAR), Garg / Eigen crop mask, scale factor, median scaling, clamp,
//                           compute_errors (:30-47) -> abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, ratio, valid pixels
// Nothing here is on the training step; it exists so that a run over the KITTI test split does not ship every prediction to the
// host.  One 1024-thread workgroup per image: the masked prediction is never materialised — every pass
// (count, the radix selects of the two medians, the error sums) re-evaluates the bilinear sample, ~0.5 M cheap evaluations each.
// Medians are exact (11-bit radix select on the bit patterns of the positive doubles; an even count averages the two middle
// elements as numpy.median does); sums are fixed-order, hence deterministic.
#include "sqd_common.h"

namespace {
using namespace sqd;
constexpr int NT = 1024, BINS = 2048;

__global__ __launch_bounds__(256) void post_process_kernel(const float *__restrict__ disp, double *__restrict__ out, int N, int h, int w) {
    const size_t total = (size_t)N * h * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w);
        const size_t row = i / w;                                  // n * h + y
        const double l = (double)disp[i];
        const double r = (double)disp[total + row * w + (w - 1 - x)];          // second pass ran on the flipped image
        const double m = (double)(0.5f * (disp[i] + disp[total + row * w + (w - 1 - x)]));      // 0.5 * (l + r) on the float32 arrays
        // np.linspace(0, 1, w)[x] = x * (1 / (w - 1)) (numpy: start + arange * step, last element set to stop)
        const double step = 1.0 / (double)(w - 1);
        const double lx = x == w - 1 ? 1.0 : (double)x * step;
        const double lxr = x == 0 ? 1.0 : (double)(w - 1 - x) * step;
        const double t0 = 20.0 * (lx - 0.05), t1 = 20.0 * (lxr - 0.05);
        const double lm = 1.0 - fmin(fmax(t0, 0.0), 1.0), rm = 1.0 - fmin(fmax(t1, 0.0), 1.0);
        out[i] = (rm * l + lm * r) + ((1.0 - lm) - rm) * m;
    }
}

struct EvalArgs {
    const double *pred;     // [h][w]
    const float *gt;        // [Hg][Wg]
    int h, w, Hg, Wg;
    int y0, y1, x0, x1;     // crop rows / columns (eigen), or the whole image
    int eigen;
    double min_depth, max_depth, scale;
    int median_scaling;
    double *out;            // [9]
};

struct Axis {
    int i0, i1;
    double a0, a1;
};
__device__ __forceinline__ Axis axis_of(int d, int n_src, double scale) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);            // OpenCV: float coefficient
    int i0 = (int)floorf(f);
    f -= (float)i0;
    if (i0 < 0) { i0 = 0; f = 0.f; }
    if (i0 >= n_src - 1) { i0 = n_src - 1; f = 0.f; }
    Axis a;
    a.i0 = i0;
    a.i1 = min(i0 + 1, n_src - 1);
    a.a0 = (double)(1.0f - f);
    a.a1 = (double)f;
    return a;
}

struct Sampler {
    const EvalArgs &a;
    double sx, sy;
    __device__ Sampler(const EvalArgs &a_) : a(a_), sx((double)a_.w / (double)a_.Wg), sy((double)a_.h / (double)a_.Hg) {}
    __device__ __forceinline__ bool valid(int y, int x, float g) const {
        if (a.eigen) return g > (float)a.min_depth && g < (float)a.max_depth && y >= a.y0 && y < a.y1 && x >= a.x0 && x < a.x1;
        return g > 0.f;
    }
    __device__ __forceinline__ double pred(int y, int x) const {   // horizontal pass first, then vertical (OpenCV's order)
        const Axis ax = axis_of(x, a.w, sx), ay = axis_of(y, a.h, sy);
        const double *r0 = a.pred + (size_t)ay.i0 * a.w, *r1 = a.pred + (size_t)ay.i1 * a.w;
        const double t0 = r0[ax.i0] * ax.a0 + r0[ax.i1] * ax.a1;
        const double t1 = r1[ax.i0] * ax.a0 + r1[ax.i1] * ax.a1;
        return (t0 * ay.a0 + t1 * ay.a1) * a.scale;
    }
};

__device__ double block_sum(double v, double *red) {               // fixed order: lane tree, then the 16 waves in order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}

// element of rank `rank` (0-based, ascending) among the valid pixels; WHICH = 0: ground truth, 1: scaled prediction
template <int WHICH>
__device__ double select_rank(const Sampler &S, long long rank, unsigned *hist, unsigned long long *shared) {
    unsigned long long prefix = 0;                                 // bits decided so far (high part of the key)
    int decided = 0;
    const int shifts[6] = {53, 42, 31, 20, 9, 0};
    const int widths[6] = {11, 11, 11, 11, 11, 9};
    const size_t total = (size_t)S.a.Hg * S.a.Wg;
    for (int lvl = 0; lvl < 6; ++lvl) {
        for (int i = threadIdx.x; i < BINS; i += NT) hist[i] = 0;
        __syncthreads();
        for (size_t p = threadIdx.x; p < total; p += NT) {
            const int y = (int)(p / S.a.Wg), x = (int)(p % S.a.Wg);
            const float g = S.a.gt[p];
            if (!S.valid(y, x, g)) continue;
            const double v = WHICH == 0 ? (double)g : S.pred(y, x);
            const unsigned long long key = (unsigned long long)__double_as_longlong(v);
            if (decided && (key >> (64 - decided)) != prefix) continue;
            atomicAdd(&hist[(unsigned)((key >> shifts[lvl]) & ((1u << widths[lvl]) - 1))], 1u);     // integer atomics: order-free
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
            shared[1] = (unsigned long long)r;
        }
        __syncthreads();
        prefix = (prefix << widths[lvl]) | shared[0];
        rank = (long long)shared[1];
        decided += widths[lvl];
        __syncthreads();
    }
    return __longlong_as_double((long long)prefix);
}

__global__ __launch_bounds__(NT) void depth_eval_kernel(EvalArgs a) {
    __shared__ unsigned hist[BINS];
    __shared__ unsigned long long shared[2];
    __shared__ double red[NT / 64];
    const Sampler S(a);
    const size_t total = (size_t)a.Hg * a.Wg;
    double cnt = 0.0;
    for (size_t p = threadIdx.x; p < total; p += NT)
        if (S.valid((int)(p / a.Wg), (int)(p % a.Wg), a.gt[p])) cnt += 1.0;
    const long long n = (long long)block_sum(cnt, red);
    if (n == 0) {
        if (threadIdx.x < 9) a.out[threadIdx.x] = threadIdx.x == 8 ? 0.0 : __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    double ratio = 1.0;
    if (a.median_scaling) {
        const long long r1 = (n - 1) / 2, r2 = n / 2;
        const double g1 = select_rank<0>(S, r1, hist, shared), g2 = r2 == r1 ? g1 : select_rank<0>(S, r2, hist, shared);
        const double p1 = select_rank<1>(S, r1, hist, shared), p2 = r2 == r1 ? p1 : select_rank<1>(S, r2, hist, shared);
        const double med_gt = (double)(((float)g1 + (float)g2) * 0.5f);       // numpy.median of the float32 ground truth: float mean
        const double med_pred = (p1 + p2) * 0.5;
        ratio = med_gt / med_pred;
    }
    const double t1 = 1.25, t2 = 1.25 * 1.25, t3 = 1.25 * 1.25 * 1.25;
    double s_abs = 0, s_sq = 0, s_rm = 0, s_lg = 0, c1 = 0, c2 = 0, c3 = 0;
    for (size_t p = threadIdx.x; p < total; p += NT) {
        const int y = (int)(p / a.Wg), x = (int)(p % a.Wg);
        const float gf = a.gt[p];
        if (!S.valid(y, x, gf)) continue;
        const double g = (double)gf;
        double v = S.pred(y, x) * ratio;
        v = v < a.min_depth ? a.min_depth : v;
        v = v > a.max_depth ? a.max_depth : v;
        const double th = fmax(g / v, v / g), d = g - v, lg = (double)logf(gf) - log(v);      // np.log of the float32 ground truth is float32
        c1 += th < t1 ? 1.0 : 0.0; c2 += th < t2 ? 1.0 : 0.0; c3 += th < t3 ? 1.0 : 0.0;
        s_abs += fabs(d) / g; s_sq += d * d / g; s_rm += d * d; s_lg += lg * lg;
    }
    const double dn = (double)n;
    const double r_abs = block_sum(s_abs, red), r_sq = block_sum(s_sq, red), r_rm = block_sum(s_rm, red), r_lg = block_sum(s_lg, red);
    const double r1 = block_sum(c1, red), r2 = block_sum(c2, red), r3 = block_sum(c3, red);
    if (threadIdx.x == 0) {
        a.out[0] = r_abs / dn; a.out[1] = r_sq / dn; a.out[2] = sqrt(r_rm / dn); a.out[3] = sqrt(r_lg / dn);
        a.out[4] = r1 / dn; a.out[5] = r2 / dn; a.out[6] = r3 / dn;
        a.out[7] = a.median_scaling ? ratio : __longlong_as_double(0x7ff8000000000000ll);
        a.out[8] = dn;
    }
}
}  // namespace

// disp [2N,h,w] fp32: outputs of the N images followed by the outputs of their horizontally flipped copies (as the reference
// batches them, evaluate_depth_config.py:133-137) -> out [N,h,w] fp64
extern "C" int sqd_disp_post_process(const float *disp, double *out, int N, int h, int w, void *stream) {
    SQD_CHECK_ARG(disp && out && N > 0 && h > 0 && w > 1, "sqd_disp_post_process: bad arguments");
    const size_t total = (size_t)N * h * w;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    (void)hipGetLastError();
    hipLaunchKernelGGL(post_process_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, disp, out, N, h, w);
    SQD_CHECK_LAUNCH("sqd_disp_post_process");
    return SQD_OK;
}

// pred [h,w] fp64 (the network's depth output, post-processed or not), gt [Hg,Wg] fp32 -> out [9] fp64:
// abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, median-scaling ratio (NaN when disabled), number of valid pixels (0: metrics NaN).
// eigen_crop 1: valid = min_depth < gt < max_depth inside the Garg / Eigen crop (evaluate_depth_config.py:233-241); 0: gt > 0.
extern "C" int sqd_depth_eval(const double *pred, int h, int w, const float *gt, int Hg, int Wg, int eigen_crop, double min_depth,
                              double max_depth, double pred_scale, int median_scaling, double *out, void *stream) {
    SQD_CHECK_ARG(pred && gt && out && h > 0 && w > 0 && Hg > 0 && Wg > 0 && min_depth > 0 && max_depth > min_depth && pred_scale > 0,
                  "sqd_depth_eval: bad arguments");
    EvalArgs a;
    a.pred = pred; a.gt = gt; a.h = h; a.w = w; a.Hg = Hg; a.Wg = Wg; a.eigen = eigen_crop ? 1 : 0;
    // crop = np.array([0.40810811 * gt_height, 0.99189189 * gt_height, 0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)
    a.y0 = (int)(0.40810811 * Hg); a.y1 = (int)(0.99189189 * Hg); a.x0 = (int)(0.03594771 * Wg); a.x1 = (int)(0.96405229 * Wg);
    a.min_depth = min_depth; a.max_depth = max_depth; a.scale = pred_scale; a.median_scaling = median_scaling ? 1 : 0;
    a.out = out;
    (void)hipGetLastError();
    hipLaunchKernelGGL(depth_eval_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, a);
    SQD_CHECK_LAUNCH("sqd_depth_eval");
    return SQD_OK;
}
