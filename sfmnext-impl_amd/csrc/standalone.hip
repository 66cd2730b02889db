// standalone.hip — kernels behind the reference's stand-alone layer classes (layers.py surface):
// BackprojectDepth.forward (layers.py:210-215), Project3D.forward (layers.py:247-258), SSIM.forward
// (layers.py:31-46) and — round 6 — their adjoints.  The training path never calls these one by one (they are fused inside
// photo_tile.hip); they exist so that scripts written against the reference's layers keep working, gradients included.
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

// ---- adjoints of the three layers above (round 6: the reference's SSIM / BackprojectDepth / Project3D are autograd modules; scripts that
// train through them get their gradients from these kernels — the training path differentiates the fused chain, photo_tile.hip)
// SSIM, pass 1: d out / d (window means of x, y, x^2 (= y^2's), x y) times the upstream gradient, per window -> co [plane][4][H][W]
__global__ __launch_bounds__(256) void ssim_bwd_coef_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ g,
                                                            float *__restrict__ co, int H, int W) {
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
    const float A1 = 2.f * mx * my + C1, A2 = 2.f * vxy + C2, B1 = mx * mx + my * my + C1, B2 = vx + vy + C2;
    const float iB1 = 1.f / B1, iB2 = 1.f / B2, iB = iB1 * iB2, S = A1 * A2 * iB;
    const float r = (1.f - S) * 0.5f;
    // out = clamp(r, 0, 1): torch.clamp passes the gradient at the bounds; window mean -> sum: 1 / 49
    const float kk = (r >= 0.f && r <= 1.f) ? -0.5f * k * g[(size_t)plane * HW + q] : 0.f;
    const float common = (A2 - A1) * iB, diff = S * (iB2 - iB1);
    float *o = co + (size_t)plane * 4 * HW + q;
    o[0] = kk * 2.f * (my * common + mx * diff);            // d / d mean x
    o[HW] = kk * 2.f * (mx * common + my * diff);           // d / d mean y
    o[2 * (size_t)HW] = kk * (-S * iB2);                    // d / d mean x^2 = d / d mean y^2
    o[3 * (size_t)HW] = kk * (2.f * A1 * iB);               // d / d mean x y
}
// number of offsets d in [-3, 3] with refl(r + d) == q (|r - q| <= 3): the adjoint of ReflectionPad2d(3) along one axis
__device__ __forceinline__ int refl_count(int r, int q, int n) {
    return 1 + (int)(q >= 1 && q + r <= 3) + (int)(q <= n - 2 && (n - 1 - q) + (n - 1 - r) <= 3);
}
// pass 2: every element collects the coefficients of the windows that contain it
__global__ __launch_bounds__(256) void ssim_bwd_adjoint_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ co,
                                                               float *__restrict__ gx, float *__restrict__ gy, int H, int W) {
    const int plane = blockIdx.y, HW = H * W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW) return;
    const int qy = q / W, qx = q - qy * W;
    const float *c = co + (size_t)plane * 4 * HW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int dy = -3; dy <= 3; ++dy) {
        const int py = qy + dy;
        if (py < 0 || py >= H) continue;
        const int my = refl_count(py, qy, H);
        for (int dx = -3; dx <= 3; ++dx) {
            const int px = qx + dx;
            if (px < 0 || px >= W) continue;
            const float m = (float)(my * refl_count(px, qx, W));
            const int o = py * W + px;
            a0 = fmaf(m, c[o], a0); a1 = fmaf(m, c[HW + o], a1); a2 = fmaf(m, c[2 * (size_t)HW + o], a2); a3 = fmaf(m, c[3 * (size_t)HW + o], a3);
        }
    }
    const float xv = x[(size_t)plane * HW + q], yv = y[(size_t)plane * HW + q];
    if (gx) gx[(size_t)plane * HW + q] = a0 + 2.f * xv * a2 + yv * a3;
    if (gy) gy[(size_t)plane * HW + q] = a1 + 2.f * yv * a2 + xv * a3;
}

// BackprojectDepth w.r.t. the depth: pts_i = depth * ray_i
__global__ __launch_bounds__(256) void backproject_bwd_kernel(const float *__restrict__ g_pts, const float *__restrict__ inv_K, float *__restrict__ g_depth,
                                                              int H, int W) {
    const int b = blockIdx.y, HW = H * W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW) return;
    const float *ik = inv_K + (size_t)b * 16;
    const int y = q / W, x = q - y * W;
    const float fx = (float)x, fy = (float)y;
    const float *g = g_pts + (size_t)b * 4 * HW + q;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float ray = ik[i * 4 + 0] * fx;
        ray = fmaf(ik[i * 4 + 1], fy, ray);
        ray = fmaf(ik[i * 4 + 2], 1.0f, ray);
        acc = fmaf(g[(size_t)i * HW], ray, acc);
    }
    g_depth[(size_t)b * HW + q] = acc;
}

// Project3D w.r.t. the points and (through per-workgroup partials of g_P = d / d (K T)[:3]) w.r.t. T
__global__ __launch_bounds__(256) void project3d_bwd_kernel(const float *__restrict__ pts, const float *__restrict__ K, const float *__restrict__ T,
                                                            const float *__restrict__ g_grid, float *__restrict__ g_pts, float *__restrict__ gP_part,
                                                            int H, int W, float eps) {
    const int b = blockIdx.y, HW = H * W;
    __shared__ float P[12];
    __shared__ float red[4][12];
    if (threadIdx.x < 12) {
        const int i = threadIdx.x / 4, j = threadIdx.x % 4;
        const float *Kb = K + (size_t)b * 16, *Tb = T + (size_t)b * 16;
        float acc = Kb[i * 4 + 0] * Tb[0 * 4 + j];
        acc += Kb[i * 4 + 1] * Tb[1 * 4 + j];
        acc += Kb[i * 4 + 2] * Tb[2 * 4 + j];
        acc += Kb[i * 4 + 3] * Tb[3 * 4 + j];
        P[threadIdx.x] = acc;
    }
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;
    float gp[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) gp[j] = 0.f;
    if (q < HW) {
        const float *p = pts + (size_t)b * 4 * HW + q;
        const float X[4] = {p[0], p[HW], p[2 * (size_t)HW], p[3 * (size_t)HW]};
        float cam[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = P[i * 4 + 0] * X[0];
            acc = fmaf(P[i * 4 + 1], X[1], acc);
            acc = fmaf(P[i * 4 + 2], X[2], acc);
            acc = fmaf(P[i * 4 + 3], X[3], acc);
            cam[i] = acc;
        }
        const float z = cam[2] + eps, iz = 1.f / z;
        const float2 gg = *reinterpret_cast<const float2 *>(g_grid + ((size_t)b * HW + q) * 2);
        const float gu = 2.f * gg.x / (float)(W - 1), gv = 2.f * gg.y / (float)(H - 1);        // grid = (pix / (n - 1) - 0.5) * 2
        const float gc[3] = {gu * iz, gv * iz, -(gu * cam[0] + gv * cam[1]) * iz * iz};
        float *go = g_pts + (size_t)b * 4 * HW + q;
#pragma unroll
        for (int j = 0; j < 4; ++j) go[(size_t)j * HW] = P[j] * gc[0] + P[4 + j] * gc[1] + P[8 + j] * gc[2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) gp[i * 4 + j] = gc[i] * X[j];
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const float v = wave_sum(gp[j]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) gP_part[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// g_T[b] = K[b]^T (rows 0..2) g_P[b], g_P[b] = the fixed-order sum of the workgroup partials
__global__ __launch_bounds__(64) void project3d_gT_kernel(const float *__restrict__ gP_part, const float *__restrict__ K, float *__restrict__ g_T, int nblk) {
    const int b = blockIdx.x, t = threadIdx.x;
    __shared__ float gP[12];
    if (t < 12) {
        float s = 0.f;
        for (int k = 0; k < nblk; ++k) s += gP_part[((size_t)b * nblk + k) * 12 + t];
        gP[t] = s;
    }
    __syncthreads();
    if (t < 16) {
        const int r = t / 4, j = t % 4;
        const float *Kb = K + (size_t)b * 16;
        g_T[(size_t)b * 16 + t] = Kb[0 * 4 + r] * gP[0 * 4 + j] + Kb[1 * 4 + r] * gP[1 * 4 + j] + Kb[2 * 4 + r] * gP[2 * 4 + j];
    }
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

extern "C" int sqd_ssim_bwd(const float *x, const float *y, const float *g, float *coef_ws, float *g_x, float *g_y, int planes, int H, int W,
                            void *stream) {
    SQD_CHECK_ARG(x && y && g && coef_ws && (g_x || g_y) && planes > 0 && H >= 4 && W >= 4, "sqd_ssim_bwd: bad arguments (H, W >= 4)");
    (void)hipGetLastError();
    const dim3 grid((H * W + 255) / 256, planes);
    hipLaunchKernelGGL(ssim_bwd_coef_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, g, coef_ws, H, W);
    hipLaunchKernelGGL(ssim_bwd_adjoint_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, coef_ws, g_x, g_y, H, W);
    SQD_CHECK_LAUNCH("sqd_ssim_bwd");
    return SQD_OK;
}

extern "C" int sqd_backproject_bwd(const float *g_points, const float *inv_K, float *g_depth, int B, int H, int W, void *stream) {
    SQD_CHECK_ARG(g_points && inv_K && g_depth && B > 0 && H > 0 && W > 0, "sqd_backproject_bwd: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(backproject_bwd_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, g_points, inv_K, g_depth, H, W);
    SQD_CHECK_LAUNCH("sqd_backproject_bwd");
    return SQD_OK;
}

extern "C" int sqd_project3d_bwd_nblk(int H, int W) { return (H * W + 255) / 256; }
extern "C" int sqd_project3d_bwd(const float *points, const float *K, const float *T, const float *g_grid, float *g_points, float *gP_part,
                                 float *g_T, int B, int H, int W, float eps, void *stream) {
    SQD_CHECK_ARG(points && K && T && g_grid && g_points && gP_part && g_T && B > 0 && H > 1 && W > 1, "sqd_project3d_bwd: bad arguments");
    (void)hipGetLastError();
    const int nblk = sqd_project3d_bwd_nblk(H, W);
    hipLaunchKernelGGL(project3d_bwd_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, points, K, T, g_grid, g_points, gP_part, H, W, eps);
    hipLaunchKernelGGL(project3d_gT_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, gP_part, K, g_T, nblk);
    SQD_CHECK_LAUNCH("sqd_project3d_bwd");
    return SQD_OK;
}
