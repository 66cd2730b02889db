// smooth.hip — edge-aware smoothness of the mean-normalised depth (include/sqd.h section 4).
// replaces: trainer.py:535-536 (disp / (mean + 1e-7)) and get_smooth_loss (reference layers.py:267-280).
//
// Roofline: HBM.  Forward reads depth 4 + colour 12 B/px (neighbours come from L1/L2), writes only
// per-block partial sums; backward re-reads the same 16 B/px and writes 4 B/px: 36 B/px in total.
//
// The loss is homogeneous of degree 1 in the normalised depth n = d/m', so sum_j dL/dn_j * d_j =
// m' * L_b: the gradient through the per-image mean needs no reduction beyond the forward sums.
#include "sqd_common.h"

namespace {
using namespace sqd;
constexpr int SM_PX_PER_BLOCK = 1024;

__device__ __forceinline__ float edge_w(const float *__restrict__ col, size_t HW, size_t a, size_t b) {
    float g = fabsf(col[a] - col[b]) + fabsf(col[HW + a] - col[HW + b]) + fabsf(col[2 * HW + a] - col[2 * HW + b]);
    return __expf(-(g * (1.f / 3.f)));
}

// sum over k < n of src[k * stride], by the whole workgroup of 256 threads in a fixed order (thread t takes k = t, t + 256, ...; lanes, then
// waves): every workgroup of the forward and of the backward kernel gets the same bits.  (Round 1 let thread 0 walk the partials alone
// while 255 threads waited — 240 serial loads per workgroup were 30 of smooth_bwd's 46 us.)
__device__ __forceinline__ float block_sum_strided(const float *__restrict__ src, int n, int stride, float *red4) {
    float s = 0.f;
    for (int k = threadIdx.x; k < n; k += 256) s += src[(size_t)k * stride];
    s = wave_sum(s);
    __syncthreads();                                 // (red4 may still be read from a previous call)
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = s;
    __syncthreads();
    return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}
__device__ __forceinline__ float image_mean(const float *__restrict__ part, int b, int nblk, int HW, float *red4) {
    // the per-block depth sums of sqd_depth_up_fwd
    return block_sum_strided(part + (size_t)b * nblk * 2 + 1, nblk, 2, red4) / (float)HW;
}

__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float *__restrict__ depth, const float *__restrict__ color,
                                                         const float *__restrict__ part, int nblk,
                                                         float *__restrict__ sm_part, int H, int W, int nblk_s) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const size_t HW = (size_t)H * W;
    __shared__ float red4[4];
    const float s_m = part ? image_mean(part, b, nblk, (int)HW, red4) : 0.f;      // (workgroup-uniform branch)
    const float im = part ? 1.f / (s_m + 1e-7f) : 1.f;     // part == NULL: caller already normalised (get_smooth_loss)
    const float *d = depth + (size_t)b * HW, *col = color + (size_t)b * 3 * HW;
    float sx = 0.f, sy = 0.f;
    const int q0 = blk * SM_PX_PER_BLOCK + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int q = q0 + k;
        if (q < (int)HW) {
            int y = q / W, x = q - y * W;
            float n0 = d[q] * im;
            if (x + 1 < W) sx += fabsf(n0 - d[q + 1] * im) * edge_w(col, HW, q, q + 1);
            if (y + 1 < H) sy += fabsf(n0 - d[q + W] * im) * edge_w(col, HW, q, q + W);
        }
    }
    __shared__ float red[2][4];
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = sx;
        red[1][wv] = sy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *o = sm_part + ((size_t)b * nblk_s + blk) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : v < 0.f ? -1.f : 0.f; }

__global__ __launch_bounds__(256) void smooth_bwd_kernel(const float *__restrict__ depth, const float *__restrict__ color,
                                                         const float *__restrict__ part, int nblk,
                                                         const float *__restrict__ sm_part, int nblk_s, float gout,
                                                         float *__restrict__ g_depth, long long gstride, int B, int H,
                                                         int W) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const size_t HW = (size_t)H * W;
    __shared__ float red4[4];
    const float inx = 1.f / ((float)B * (float)H * (float)(W - 1)), iny = 1.f / ((float)B * (float)(H - 1) * (float)W);
    // part == NULL (workgroup-uniform): the stand-alone get_smooth_loss on an already normalised disparity — no mean-normalisation term
    const float s_m = part ? image_mean(part, b, nblk, (int)HW, red4) : 0.f;
    const float ax = part ? block_sum_strided(sm_part + (size_t)b * nblk_s * 2, nblk_s, 2, red4) : 0.f;
    const float ay = part ? block_sum_strided(sm_part + (size_t)b * nblk_s * 2 + 1, nblk_s, 2, red4) : 0.f;
    const float s_Lb = ax * inx + ay * iny;
    const float mp = s_m + 1e-7f, im = part ? 1.f / mp : 1.f;
    const float mean_term = part ? -s_Lb * im / (float)HW : 0.f;
    const float *d = depth + (size_t)b * HW, *col = color + (size_t)b * 3 * HW;
    float *g = g_depth + (size_t)b * gstride;
    const int q0 = blk * SM_PX_PER_BLOCK + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int q = q0 + k;
        if (q < (int)HW) {
            int y = q / W, x = q - y * W;
            float n0 = d[q] * im, gn = 0.f;
            if (x + 1 < W) gn += inx * sgn(n0 - d[q + 1] * im) * edge_w(col, HW, q, q + 1);
            if (x > 0) gn -= inx * sgn(d[q - 1] * im - n0) * edge_w(col, HW, q - 1, q);
            if (y + 1 < H) gn += iny * sgn(n0 - d[q + W] * im) * edge_w(col, HW, q, q + W);
            if (y > 0) gn -= iny * sgn(d[q - W] * im - n0) * edge_w(col, HW, q - W, q);
            g[q] = gout * (gn * im + mean_term);
        }
    }
}
}  // namespace

extern "C" int sqd_smooth_nblk(int H, int W) { return (H * W + SM_PX_PER_BLOCK - 1) / SM_PX_PER_BLOCK; }

namespace {
// the three scalars of compute_losses from the kernels' partial sums, one workgroup, fixed order:
//   photo = sum(loss_part) * w_photo;  smooth = sum(sm_part[.., 0]) * w_x + sum(sm_part[.., 1]) * w_y;  out = (photo + smooth_weight * smooth, photo, smooth)
__global__ __launch_bounds__(256) void chain_loss_kernel(const float *__restrict__ loss_part, int n_loss, const float *__restrict__ sm_part,
                                                         int n_sm, float w_photo, float w_x, float w_y, float smooth_weight,
                                                         float *__restrict__ out) {
    __shared__ float red[3][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float a0 = 0.f, a1 = 0.f, sx = 0.f, sy = 0.f;
    int i = t;
    for (; i + 256 < n_loss; i += 512) {
        a0 += loss_part[i];
        a1 += loss_part[i + 256];
    }
    if (i < n_loss) a0 += loss_part[i];
    for (int j = t; j < n_sm; j += 256) {
        const float2 v = reinterpret_cast<const float2 *>(sm_part)[j];
        sx += v.x;
        sy += v.y;
    }
    float v0 = sqd::wave_sum_to_lane63(a0 + a1), v1 = sqd::wave_sum_to_lane63(sx), v2 = sqd::wave_sum_to_lane63(sy);
    if (lane == 63) { red[0][wave] = v0; red[1][wave] = v1; red[2][wave] = v2; }
    __syncthreads();
    if (t == 0) {
        const float photo = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) * w_photo;
        const float smooth = (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) * w_x + (((red[2][0] + red[2][1]) + red[2][2]) + red[2][3]) * w_y;
        out[0] = photo + smooth_weight * smooth;
        out[1] = photo;
        out[2] = smooth;
    }
}
}  // namespace

extern "C" int sqd_chain_loss(const float *loss_part, int n_loss, const float *sm_part, int n_sm, float w_photo, float w_x, float w_y,
                              float smooth_weight, float *out, void *stream) {
    SQD_CHECK_ARG(loss_part && sm_part && out && n_loss > 0 && n_sm > 0, "sqd_chain_loss: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(chain_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_part, n_loss, sm_part, n_sm, w_photo, w_x, w_y,
                       smooth_weight, out);
    SQD_CHECK_LAUNCH("sqd_chain_loss");
    return SQD_OK;
}

extern "C" int sqd_smooth_fwd(const float *depth, const float *color, const float *part, int nblk, float *sm_part, int B,
                              int H, int W, void *stream) {
    SQD_CHECK_ARG(depth && color && sm_part, "sqd_smooth_fwd: null pointer");
    SQD_CHECK_ARG(B > 0 && H > 1 && W > 1 && (!part || nblk > 0), "sqd_smooth_fwd: bad shape");
    const int nb = sqd_smooth_nblk(H, W);
    (void)hipGetLastError();   // drop any stale error left by other HIP users of this thread
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, depth, color, part, nblk, sm_part,
                       H, W, nb);
    SQD_CHECK_LAUNCH("sqd_smooth_fwd");
    return SQD_OK;
}

extern "C" int sqd_smooth_bwd(const float *depth, const float *color, const float *part, int nblk, const float *sm_part,
                              float gout, float *g_depth, int64_t g_depth_img_stride, int B, int H, int W,
                              void *stream) {
    SQD_CHECK_ARG(depth && color && (sm_part || !part) && g_depth, "sqd_smooth_bwd: null pointer");
    SQD_CHECK_ARG(g_depth_img_stride >= (int64_t)H * W, "sqd_smooth_bwd: g_depth_img_stride too small");
    SQD_CHECK_ARG(B > 0 && H > 1 && W > 1 && (nblk > 0 || !part), "sqd_smooth_bwd: bad shape");
    const int nb = sqd_smooth_nblk(H, W);
    (void)hipGetLastError();   // drop any stale error left by other HIP users of this thread
    hipLaunchKernelGGL(smooth_bwd_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, depth, color, part, nblk, sm_part,
                       nb, gout, g_depth, (long long)g_depth_img_stride, B, H, W);
    SQD_CHECK_LAUNCH("sqd_smooth_bwd");
    return SQD_OK;
}
