// depth_pose.hip — (1) bilinear depth upsample + per-image reductions, (2) pose -> projection matrices,
// and their adjoints.  See include/sqd.h for the reference functions each entry point replaces.
//
// Roofline: (1) is HBM-bound: reads h*w*4 B, writes H*W*4 B per image (5 B/px at x2); (2) is
// launch-latency-bound (B*S threads of work) — it exists to replace ~80 tiny ATen launches.
#include <stdarg.h>

#include "sqd_common.h"

namespace sqd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sqd

extern "C" int sqd_abi_version(void) { return SQD_ABI_VERSION; }
extern "C" const char *sqd_last_error(void) { return sqd::g_err; }

namespace {
using namespace sqd;

constexpr int UP_PX_PER_BLOCK = 1024;  // 256 threads x 4 consecutive pixels

// Source index / weights of F.interpolate(bilinear, align_corners=False), the arithmetic of
// oracle/warp_chain.c::sqo_depth_up (bit-identical to ATen's CPU kernel).
struct Tap {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Tap make_tap(int dst, float scale, int in_size) {
    float f = scale * ((float)dst + 0.5f) - 0.5f;
    f = f < 0.f ? 0.f : f;
    Tap t;
    t.i0 = (int)f;
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = f - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

__global__ __launch_bounds__(256) void depth_up_fwd_kernel(const float *__restrict__ disp, float *__restrict__ depth,
                                                           float *__restrict__ part, int h, int w, int H, int W,
                                                           int nblk) {
    const int b = blockIdx.y, blk = blockIdx.x;
    const int HW = H * W;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const float *p = disp + (size_t)b * h * w;
    float s_inv = 0.f, s_d = 0.f;
    const int q0 = blk * UP_PX_PER_BLOCK + threadIdx.x * 4;
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int q = q0 + k;
        if (q < HW) {
            int y = q / W, x = q - y * W;
            Tap ty = make_tap(y, sy, h), tx = make_tap(x, sx, w);
            float w00 = ty.l0 * tx.l0, w01 = ty.l0 * tx.l1, w10 = ty.l1 * tx.l0, w11 = ty.l1 * tx.l1;
            float acc = w01 * p[ty.i0 * w + tx.i1];
            acc = fmaf(w00, p[ty.i0 * w + tx.i0], acc);
            acc = fmaf(w10, p[ty.i1 * w + tx.i0], acc);
            acc = fmaf(w11, p[ty.i1 * w + tx.i1], acc);
            out[k] = acc;
            s_inv += 1.0f / acc;
            s_d += acc;
        } else {
            out[k] = 0.f;
        }
    }
    float *dst = depth + (size_t)b * HW + q0;
    if (q0 + 3 < HW && (HW & 3) == 0) {
        *reinterpret_cast<float4 *>(dst) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        for (int k = 0; k < 4; ++k)
            if (q0 + k < HW) dst[k] = out[k];
    }
    __shared__ float red[2][4];
    s_inv = wave_sum(s_inv);
    s_d = wave_sum(s_d);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = s_inv;
        red[1][wv] = s_d;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *o = part + ((size_t)b * nblk + blk) * 2;
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// adjoint of the upsample (gather form: one thread per low-res pixel) with the mean-inverse-depth
// term folded in: g_full(q) = g_depth(q) - g_mid[b] / (depth(q)^2 * HW)
__global__ __launch_bounds__(256) void depth_up_bwd_kernel(const float *__restrict__ g_depth, int ng,
                                                           const float *__restrict__ depth,
                                                           const float *__restrict__ g_mid,
                                                           float *__restrict__ g_lr, int h, int w, int H, int W) {
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= h * w) return;
    const int i = q / w, j = q - i * w;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const float isy = (float)H / (float)h, isx = (float)W / (float)w;
    const int HW = H * W;
    const float gm = g_mid ? -g_mid[b] / (float)HW : 0.f;
    // candidate destination rows/cols whose taps can touch source index i / j (conservative bounds)
    int ylo = max(0, (int)floorf(((float)i - 1.f) * isy) - 1), yhi = min(H - 1, (int)ceilf(((float)i + 2.f) * isy) + 1);
    int xlo = max(0, (int)floorf(((float)j - 1.f) * isx) - 1), xhi = min(W - 1, (int)ceilf(((float)j + 2.f) * isx) + 1);
    const float *gd = g_depth + (size_t)b * ng * HW;
    const float *dp = depth + (size_t)b * HW;
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
        Tap ty = make_tap(y, sy, h);
        float wy0 = ty.i0 == i ? ty.l0 : 0.f, wy1 = ty.i1 == i ? ty.l1 : 0.f;
        if (wy0 == 0.f && wy1 == 0.f) continue;
        for (int x = xlo; x <= xhi; ++x) {
            Tap tx = make_tap(x, sx, w);
            float wx0 = tx.i0 == j ? tx.l0 : 0.f, wx1 = tx.i1 == j ? tx.l1 : 0.f;
            if (wx0 == 0.f && wx1 == 0.f) continue;
            float g = gd[y * W + x];
            for (int n = 1; n < ng; ++n) g += gd[(size_t)n * HW + y * W + x];
            if (g_mid) {
                float d = dp[y * W + x];
                g += gm / (d * d);
            }
            // the four forward weights are the rounded products l_y * l_x
            acc += g * (wy0 * wx0 + wy0 * wx1 + wy1 * wx0 + wy1 * wx1);
        }
    }
    g_lr[(size_t)b * h * w + q] = acc;
}

// ---- pose ----------------------------------------------------------------------------------------
struct Rod {
    float x, y, z, ca, sa, C, ang, inv;
};
__device__ __forceinline__ void rodrigues(const float v[3], float R[9], Rod &r) {
    float ang = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);  // torch.norm(vec,2,2)   layers.py:116
    float inv = 1.f / (ang + 1e-7f);
    float x = v[0] * inv, y = v[1] * inv, z = v[2] * inv;         // axis = vec/(angle+1e-7) :117
    float ca = cosf(ang), sa = sinf(ang), C = 1.f - ca;
    float xs = x * sa, ys = y * sa, zs = z * sa, xC = x * C, yC = y * C, zC = z * C;
    float xyC = x * yC, yzC = y * zC, zxC = z * xC;
    R[0] = x * xC + ca; R[1] = xyC - zs;   R[2] = zxC + ys;
    R[3] = xyC + zs;    R[4] = y * yC + ca; R[5] = yzC - xs;
    R[6] = zxC - ys;    R[7] = yzC + xs;   R[8] = z * zC + ca;
    r = Rod{x, y, z, ca, sa, C, ang, inv};
}

// one 64-lane block per image; lane s < S builds T and P for source s
__global__ __launch_bounds__(64) void pose_mats_fwd_kernel(const float *__restrict__ aa, const float *__restrict__ tr,
                                                           unsigned invert_mask, const float *__restrict__ K,
                                                           const float *__restrict__ part, int nblk, int HW,
                                                           float *__restrict__ mid_out, float *__restrict__ Tm,
                                                           float *__restrict__ Pm, int S) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float mid = 1.f;
    if (part) {
        float s = 0.f;
        for (int k = lane; k < nblk; k += 64) s += part[((size_t)b * nblk + k) * 2];
        s = wave_sum(s);
        mid = s / (float)HW;
        if (lane == 0 && mid_out) mid_out[b] = mid;
    }
    if (lane >= S) return;
    const int s = lane;
    const float *v = aa + ((size_t)b * S + s) * 3, *t0 = tr + ((size_t)b * S + s) * 3;
    float vv[3] = {v[0], v[1], v[2]}, R[9];
    Rod rd;
    rodrigues(vv, R, rd);
    float t[3] = {t0[0] * mid, t0[1] * mid, t0[2] * mid};           // trainer.py:420-421
    float M[16];
    if ((invert_mask >> s) & 1u) {                                   // M = R^T . T(-t)   layers.py:82-90
        float nt[3] = {t[0] * -1.f, t[1] * -1.f, t[2] * -1.f};
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[j * 3 + i];
            M[i * 4 + 3] = (R[0 * 3 + i] * nt[0] + R[1 * 3 + i] * nt[1]) + R[2 * 3 + i] * nt[2];
        }
    } else {                                                         // M = T(t) . R
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j];
            M[i * 4 + 3] = t[i];
        }
    }
    M[12] = M[13] = M[14] = 0.f;
    M[15] = 1.f;
    float *To = Tm + ((size_t)b * S + s) * 16;
    for (int k = 0; k < 16; ++k) To[k] = M[k];
    const float *Kb = K + (size_t)b * 16;
    float *Po = Pm + ((size_t)b * S + s) * 12;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {                                // P = (K @ T)[:3]   layers.py:248
            float acc = Kb[i * 4 + 0] * M[0 * 4 + j];
            acc += Kb[i * 4 + 1] * M[1 * 4 + j];
            acc += Kb[i * 4 + 2] * M[2 * 4 + j];
            acc += Kb[i * 4 + 3] * M[3 * 4 + j];
            Po[i * 4 + j] = acc;
        }
}

__global__ __launch_bounds__(64) void pose_mats_bwd_kernel(const float *__restrict__ aa, const float *__restrict__ tr,
                                                           unsigned invert_mask, const float *__restrict__ K,
                                                           const float *__restrict__ mid_in,
                                                           const float *__restrict__ gP, float *__restrict__ g_aa,
                                                           float *__restrict__ g_tr, float *__restrict__ g_mid,
                                                           int S) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float gmid = 0.f;
    if (lane < S) {
        const int s = lane;
        const float mid = mid_in ? mid_in[b] : 1.f;
        const float *v = aa + ((size_t)b * S + s) * 3, *t0 = tr + ((size_t)b * S + s) * 3;
        float vv[3] = {v[0], v[1], v[2]}, R[9];
        Rod r;
        rodrigues(vv, R, r);
        const float *Kb = K + (size_t)b * 16, *g = gP + ((size_t)b * S + s) * 12;
        float gM[12];                                                // rows 0..2 of g_M (row 3 of M is constant)
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 4; ++j)
                gM[k * 4 + j] = Kb[0 * 4 + k] * g[0 * 4 + j] + Kb[1 * 4 + k] * g[1 * 4 + j] + Kb[2 * 4 + k] * g[2 * 4 + j];
        float gR[9], gt[3];
        if ((invert_mask >> s) & 1u) {
            float nt[3] = {-t0[0] * mid, -t0[1] * mid, -t0[2] * mid};
            float gnt[3] = {0.f, 0.f, 0.f};
            for (int i = 0; i < 3; ++i)
                for (int k = 0; k < 3; ++k) {
                    // M[i][k] = R[k][i];  M[i][3] = sum_k R[k][i] * nt[k]
                    gR[k * 3 + i] = gM[i * 4 + k] + gM[i * 4 + 3] * nt[k];
                    gnt[k] += R[k * 3 + i] * gM[i * 4 + 3];
                }
            for (int k = 0; k < 3; ++k) gt[k] = -gnt[k];
        } else {
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) gR[i * 3 + j] = gM[i * 4 + j];
                gt[i] = gM[i * 4 + 3];
            }
        }
        for (int k = 0; k < 3; ++k) {
            g_tr[((size_t)b * S + s) * 3 + k] = gt[k] * mid;
            gmid += gt[k] * t0[k];
        }
        // ---- Rodrigues adjoint ----
        const float x = r.x, y = r.y, z = r.z, ca = r.ca, sa = r.sa, C = r.C;
        const float xC = x * C, yC = y * C, zC = z * C;
        float gx = 0, gy = 0, gz = 0, gca = 0, gsa = 0, gC = 0, gxC = 0, gyC = 0, gzC = 0;
        gx += gR[0] * xC; gxC += gR[0] * x; gca += gR[0];
        gy += gR[4] * yC; gyC += gR[4] * y; gca += gR[4];
        gz += gR[8] * zC; gzC += gR[8] * z; gca += gR[8];
        float gxyC = gR[1] + gR[3], gzs = gR[3] - gR[1];
        float gzxC = gR[2] + gR[6], gys = gR[2] - gR[6];
        float gyzC = gR[5] + gR[7], gxs = gR[7] - gR[5];
        gx += gxyC * yC; gyC += gxyC * x;
        gy += gyzC * zC; gzC += gyzC * y;
        gz += gzxC * xC; gxC += gzxC * z;
        gx += gxC * C; gC += gxC * x;
        gy += gyC * C; gC += gyC * y;
        gz += gzC * C; gC += gzC * z;
        gx += gxs * sa; gsa += gxs * x;
        gy += gys * sa; gsa += gys * y;
        gz += gzs * sa; gsa += gzs * z;
        gca -= gC;
        float gang = -sa * gca + ca * gsa;
        float gv[3] = {gx * r.inv, gy * r.inv, gz * r.inv};
        gang += -(gx * vv[0] + gy * vv[1] + gz * vv[2]) * r.inv * r.inv;
        if (r.ang > 0.f)
            for (int k = 0; k < 3; ++k) gv[k] += gang * vv[k] / r.ang;   // d||v||/dv, 0 at the origin
        for (int k = 0; k < 3; ++k) g_aa[((size_t)b * S + s) * 3 + k] = gv[k];
    }
    gmid = wave_sum(gmid);
    if (lane == 0 && g_mid) g_mid[b] = gmid;
}
}  // namespace

extern "C" int sqd_depth_up_nblk(int H, int W) { return (H * W + UP_PX_PER_BLOCK - 1) / UP_PX_PER_BLOCK; }

extern "C" int sqd_depth_up_fwd(const float *disp_lr, float *depth, float *part, int B, int h, int w, int H, int W,
                                void *stream) {
    SQD_CHECK_ARG(disp_lr && depth && part, "sqd_depth_up_fwd: null pointer");
    SQD_CHECK_ARG(B > 0 && h > 0 && w > 0 && H >= h && W >= w, "sqd_depth_up_fwd: bad shape B=%d h=%d w=%d H=%d W=%d", B, h, w, H, W);
    int nblk = sqd_depth_up_nblk(H, W);
    (void)hipGetLastError();   // drop any stale error left by other HIP users of this thread
    hipLaunchKernelGGL(depth_up_fwd_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, disp_lr, depth, part, h, w,
                       H, W, nblk);
    SQD_CHECK_LAUNCH("sqd_depth_up_fwd");
    return SQD_OK;
}

extern "C" int sqd_depth_up_bwd(const float *g_depth, int ng, const float *depth, const float *g_mid, float *g_disp_lr,
                                int B, int h, int w, int H, int W, void *stream) {
    SQD_CHECK_ARG(g_depth && depth && g_disp_lr && ng >= 1, "sqd_depth_up_bwd: null pointer / ng < 1");
    SQD_CHECK_ARG(B > 0 && h > 0 && w > 0 && H >= h && W >= w, "sqd_depth_up_bwd: bad shape");
    (void)hipGetLastError();   // drop any stale error left by other HIP users of this thread
    hipLaunchKernelGGL(depth_up_bwd_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, g_depth,
                       ng, depth, g_mid, g_disp_lr, h, w, H, W);
    SQD_CHECK_LAUNCH("sqd_depth_up_bwd");
    return SQD_OK;
}

static unsigned invert_mask_of(const int32_t *inv, int S) {
    unsigned m = 0;
    for (int s = 0; s < S; ++s)
        if (inv[s]) m |= 1u << s;
    return m;
}

extern "C" int sqd_pose_mats_fwd(const float *axisangle, const float *translation, const int32_t *invert_host,
                                 const float *K, const float *part, int nblk, int HW, float *mid, float *T, float *P,
                                 int B, int S, void *stream) {
    SQD_CHECK_ARG(axisangle && translation && invert_host && K && T && P, "sqd_pose_mats_fwd: null pointer");
    SQD_CHECK_ARG(B > 0 && S > 0 && S <= SQD_MAX_SOURCES, "sqd_pose_mats_fwd: bad B=%d S=%d", B, S);
    SQD_CHECK_ARG(!part || (nblk > 0 && HW > 0), "sqd_pose_mats_fwd: part given but nblk/HW invalid");
    (void)hipGetLastError();   // drop any stale error left by other HIP users of this thread
    hipLaunchKernelGGL(pose_mats_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, axisangle, translation,
                       invert_mask_of(invert_host, S), K, part, nblk, HW, mid, T, P, S);
    SQD_CHECK_LAUNCH("sqd_pose_mats_fwd");
    return SQD_OK;
}

extern "C" int sqd_pose_mats_bwd(const float *axisangle, const float *translation, const int32_t *invert_host,
                                 const float *K, const float *mid, const float *g_P, float *g_axisangle,
                                 float *g_translation, float *g_mid, int B, int S, void *stream) {
    SQD_CHECK_ARG(axisangle && translation && invert_host && K && g_P && g_axisangle && g_translation,
                  "sqd_pose_mats_bwd: null pointer");
    SQD_CHECK_ARG(B > 0 && S > 0 && S <= SQD_MAX_SOURCES, "sqd_pose_mats_bwd: bad B=%d S=%d", B, S);
    (void)hipGetLastError();   // drop any stale error left by other HIP users of this thread
    hipLaunchKernelGGL(pose_mats_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, axisangle, translation,
                       invert_mask_of(invert_host, S), K, mid, g_P, g_axisangle, g_translation, g_mid, S);
    SQD_CHECK_LAUNCH("sqd_pose_mats_bwd");
    return SQD_OK;
}
