// standalone.hip — forward kernels behind the reference's stand-alone layer classes (layers.py surface):
// BackprojectDepth.forward (layers.py:210-215), Project3D.forward (layers.py:247-258), SSIM.forward
// (layers.py:31-46).  The training path never calls these one by one (they are fused inside
// photo_fwd_pk.hip); they exist so that scripts written against the reference's layers keep working.
// Same canonical arithmetic as the fused kernel (oracle/warp_chain.c): FMA chains for the two big
// products, true divisions.
#include "sqd_common.h"

namespace {
using namespace sqd;

__global__ __launch_bounds__(256) void backproject_kernel(const float *__restrict__ depth, const float *__restrict__ inv_K,
                                                          float *__restrict__ pts, int H, int W) {
    const int b = blockIdx.y, HW = H * W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW) return;
    const float *ik = inv_K + (size_t)b * 16;
    const int y = q / W, x = q - y * W;
    const float fx = (float)x, fy = (float)y, d = depth[(size_t)b * HW + q];
    float *o = pts + (size_t)b * 4 * HW + q;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = ik[i * 4 + 0] * fx;
        acc = fmaf(ik[i * 4 + 1], fy, acc);
        acc = fmaf(ik[i * 4 + 2], 1.0f, acc);
        o[(size_t)i * HW] = d * acc;
    }
    o[(size_t)3 * HW] = 1.0f;
}

__global__ __launch_bounds__(256) void project3d_kernel(const float *__restrict__ pts, const float *__restrict__ K,
                                                        const float *__restrict__ T, float *__restrict__ grid, int H, int W,
                                                        float eps) {
    const int b = blockIdx.y, HW = H * W;
    __shared__ float P[12];
    if (threadIdx.x < 12) {
        const int i = threadIdx.x / 4, j = threadIdx.x % 4;
        const float *Kb = K + (size_t)b * 16, *Tb = T + (size_t)b * 16;
        float acc = Kb[i * 4 + 0] * Tb[0 * 4 + j];              // (K @ T)[:3]   layers.py:248
        acc += Kb[i * 4 + 1] * Tb[1 * 4 + j];
        acc += Kb[i * 4 + 2] * Tb[2 * 4 + j];
        acc += Kb[i * 4 + 3] * Tb[3 * 4 + j];
        P[threadIdx.x] = acc;
    }
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW) return;
    const float *p = pts + (size_t)b * 4 * HW + q;
    const float X0 = p[0], X1 = p[HW], X2 = p[2 * (size_t)HW], X3 = p[3 * (size_t)HW];
    float cam[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = P[i * 4 + 0] * X0;
        acc = fmaf(P[i * 4 + 1], X1, acc);
        acc = fmaf(P[i * 4 + 2], X2, acc);
        acc = fmaf(P[i * 4 + 3], X3, acc);
        cam[i] = acc;
    }
    const float z = cam[2] + eps;
    const float u = (cam[0] / z) / (float)(W - 1), v = (cam[1] / z) / (float)(H - 1);
    *reinterpret_cast<float2 *>(grid + ((size_t)b * HW + q) * 2) = make_float2((u - 0.5f) * 2.0f, (v - 0.5f) * 2.0f);
}

__device__ __forceinline__ int refl(int p, int n) {
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

// one thread per output element, 49 taps straight from global memory (L1/L2 serve the overlap)
__global__ __launch_bounds__(256) void ssim_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                   float *__restrict__ out, int H, int W) {
    const int plane = blockIdx.y, HW = H * W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW) return;
    const int py = q / W, px = q - py * W;
    const float *xp = x + (size_t)plane * HW, *yp = y + (size_t)plane * HW;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
    for (int dy = -3; dy <= 3; ++dy) {
        const int ry = refl(py + dy, H) * W;
        for (int dx = -3; dx <= 3; ++dx) {
            const int o = ry + refl(px + dx, W);
            const float a = xp[o], b = yp[o];
            sx += a; sy += b;
            sxx = fmaf(a, a, sxx); syy = fmaf(b, b, syy); sxy = fmaf(a, b, sxy);
        }
    }
    const float k = 1.0f / 49.0f, C1 = 0.0001f, C2 = 0.0009f;
    const float mx = sx * k, my = sy * k;
    const float vx = sxx * k - mx * mx, vy = syy * k - my * my, vxy = sxy * k - mx * my;
    const float n = (2.f * mx * my + C1) * (2.f * vxy + C2), d = (mx * mx + my * my + C1) * (vx + vy + C2);
    out[(size_t)plane * HW + q] = fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
}

// F.grid_sample(img, grid, padding_mode="border", align_corners=True) — reference trainer.py:431-435 (ATen
// grid_sampler_2d).  Same tap arithmetic as the fused warp kernel (unnormalise, clip_coordinates, floor, four weights,
// out-of-range taps skipped): one thread per output pixel, all channels.
__global__ __launch_bounds__(256) void grid_sample_border_kernel(const float *__restrict__ img, const float *__restrict__ grid,
                                                                 float *__restrict__ out, int *__restrict__ x0y0, int C, int H,
                                                                 int W, int Ho, int Wo) {
    const int b = blockIdx.y, HWo = Ho * Wo, HW = H * W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HWo) return;
    const float2 g = *reinterpret_cast<const float2 *>(grid + ((size_t)b * HWo + q) * 2);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float ix = ((g.x + 1.0f) * 0.5f) * wm1, iy = ((g.y + 1.0f) * 0.5f) * hm1;
    ix = fminf(wm1, fmaxf(ix, 0.f));
    iy = fminf(hm1, fmaxf(iy, 0.f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
    const float ax = ix - fx0, ay = iy - fy0, bx = (fx0 + 1.f) - ix, by = (fy0 + 1.f) - iy;
    const float wnw = bx * by, wne = xin ? ax * by : 0.f, wsw = yin ? bx * ay : 0.f, wse = (xin && yin) ? ax * ay : 0.f;
    const int o00 = y0 * W + x0, o01 = o00 + (xin ? 1 : 0), o10 = o00 + (yin ? W : 0), o11 = o10 + (xin ? 1 : 0);
    if (x0y0) *reinterpret_cast<int2 *>(x0y0 + ((size_t)b * HWo + q) * 2) = make_int2(x0, y0);
    for (int c = 0; c < C; ++c) {
        const float *sc = img + ((size_t)b * C + c) * HW;
        float acc = sc[o00] * wnw;
        acc = fmaf(sc[o01], wne, acc);
        acc = fmaf(sc[o10], wsw, acc);
        acc = fmaf(sc[o11], wse, acc);
        out[((size_t)b * C + c) * HWo + q] = acc;
    }
}
}  // namespace

extern "C" int sqd_backproject_fwd(const float *depth, const float *inv_K, float *cam_points, int B, int H, int W, void *stream) {
    SQD_CHECK_ARG(depth && inv_K && cam_points && B > 0 && H > 0 && W > 0, "sqd_backproject_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(backproject_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, depth, inv_K,
                       cam_points, H, W);
    SQD_CHECK_LAUNCH("sqd_backproject_fwd");
    return SQD_OK;
}

extern "C" int sqd_project3d_fwd(const float *points, const float *K, const float *T, float *grid, int B, int H, int W, float eps,
                                 void *stream) {
    SQD_CHECK_ARG(points && K && T && grid && B > 0 && H > 1 && W > 1, "sqd_project3d_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(project3d_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, points, K, T, grid, H, W,
                       eps);
    SQD_CHECK_LAUNCH("sqd_project3d_fwd");
    return SQD_OK;
}

extern "C" int sqd_ssim_fwd(const float *x, const float *y, float *out, int planes, int H, int W, void *stream) {
    SQD_CHECK_ARG(x && y && out && planes > 0 && H >= 4 && W >= 4, "sqd_ssim_fwd: bad arguments (H, W >= 4)");
    (void)hipGetLastError();
    hipLaunchKernelGGL(ssim_kernel, dim3((H * W + 255) / 256, planes), dim3(256), 0, (hipStream_t)stream, x, y, out, H, W);
    SQD_CHECK_LAUNCH("sqd_ssim_fwd");
    return SQD_OK;
}

extern "C" int sqd_grid_sample_border_fwd(const float *img, const float *grid, float *out, int *x0y0, int B, int C, int H, int W,
                                          int Ho, int Wo, void *stream) {
    SQD_CHECK_ARG(img && grid && out && B > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "sqd_grid_sample_border_fwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(grid_sample_border_kernel, dim3((Ho * Wo + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, img, grid, out,
                       x0y0, C, H, W, Ho, Wo);
    SQD_CHECK_LAUNCH("sqd_grid_sample_border_fwd");
    return SQD_OK;
}
