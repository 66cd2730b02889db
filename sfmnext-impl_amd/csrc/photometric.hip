// photometric.hip — entry points of the photometric chain (include/sqd.h section 3) and its backward kernel.
//
// The forward kernels live in photo_tile.hip (fused warp + SSIM, identity maps, coefficient planes for the backward).
// Backward ("column march"): one wavefront owns a strip of 58 output columns (64 lanes = 58 + a 3-column halo on each side)
// and marches down TH output rows; the adjoint of ReflectionPad2d(3)+AvgPool2d(7,1) is a 6-instruction DPP chain along x
// (box7) and a scatter ring of 7 accumulator rows along y — no LDS, no barriers.
//
// Roofline: HBM.  Algorithmic bytes per target pixel (S = 2): backward 84 B + the 36 B/px coefficient
// planes photo_coef writes and this kernel reads.
#include "sqd_common.h"

namespace {
using namespace sqd;

typedef float f2ua __attribute__((ext_vector_type(2), aligned(4)));      // 8-byte load at 4-byte alignment
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
constexpr float INV49 = 1.0f / 49.0f;
constexpr int OWN0 = 3, OWN1 = 61;   // owned lanes [3, 61): 58 output columns per wavefront

// ---- projection chain: the fp32 order of oracle/warp_chain.c (bit-exact integer taps) ------------
__device__ __forceinline__ void cam_ray(const float *ik, float fx, float fy, float c[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = ik[i * 3 + 0] * fx;                 // layers.py:211  (FMA chain k = 0..2)
        acc = fmaf(ik[i * 3 + 1], fy, acc);
        acc = fmaf(ik[i * 3 + 2], 1.0f, acc);
        c[i] = acc;
    }
}

struct Strip {
    int b, y_begin, cx, xr;    // image, first output row, lane column (may be outside), reflected column
    bool own_col;              // lane owns an in-image output column
};

__device__ __forceinline__ Strip strip_of(int task, int nsx, int nsy, int TH, int W, int lane) {
    Strip s;
    int sx = task % nsx;
    int t2 = task / nsx;
    int sy = t2 % nsy;
    s.b = t2 / nsy;
    s.y_begin = sy * TH;
    s.cx = sx * SQD_STRIP_COLS - 3 + lane;
    s.xr = reflect_idx(s.cx, W);
    s.own_col = lane >= OWN0 && lane < OWN1 && s.cx >= 0 && s.cx < W;
    return s;
}

// ===================================================================================================
// backward: one wavefront per (image, source, strip).  Reads the coefficient maps written by the
// forward, box-filters them with the adjoint of ReflectionPad2d(3)+AvgPool(7) (scatter ring of 7
// accumulator rows, border multiplicities as scalars), then runs the adjoint of grid_sample /
// Project3D / BackprojectDepth per pixel.
// ===================================================================================================
__device__ __forceinline__ int refl_mult(int r, int q, int n) {
    // number of offsets d in [-3,3] with reflect(r+d) == q, for r,q in [0,n), |r-q| <= 3
    int m = 1;
    m += (q >= 1 && q + r <= 3) ? 1 : 0;
    m += (q <= n - 2 && (n - 1 - q) + (n - 1 - r) <= 3) ? 1 : 0;
    return m;
}

__global__ __launch_bounds__(256) void photo_bwd_kernel(sqd_photo_bwd_args a, int TH, int nsx, int nsy, int ntasks) {
    const int S = a.S;
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (task >= ntasks) return;
    const int H = a.H, W = a.W;
    const int HW = H * W;
    // task -> (b, s, sy, sx)
    const int per_img = nsx * nsy;
    const int bs = task / per_img;
    const int b = bs / S, s = bs - b * S;
    const Strip st = strip_of((b * nsy * nsx) + (task - bs * per_img), nsx, nsy, TH, W, lane);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const bool in_img_col = st.cx >= 0 && st.cx < W;

    float ik[9], P[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) ik[i * 3 + k] = a.inv_K[(size_t)b * 16 + i * 4 + k];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = a.P[((size_t)b * S + s) * 12 + k];

    const float *__restrict__ coef = a.coef + (size_t)b * 9 * HW;
    const uint8_t *__restrict__ idx = a.idx + (size_t)b * HW;
    const float *__restrict__ tgt = a.target + (size_t)b * 3 * HW;
    const float *__restrict__ src = a.sources[s] + (size_t)b * 3 * HW;
    const float *__restrict__ smp = a.sample[s] + (size_t)b * HW * 2;
    const float *__restrict__ dep = a.depth + (size_t)b * HW;
    float *__restrict__ gdep = a.g_depth + (size_t)b * a.g_depth_img_stride + (size_t)s * HW;
    const __amdgpu_buffer_rsrc_t coef_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(coef), 0, (unsigned)(9 * HW) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t idx_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(idx), 0, (unsigned)HW, 0x00020000);

    // border-column bookkeeping (wave-uniform): lanes of columns 0 and W-1 in this strip
    const int x_first = st.cx - lane;              // column of lane 0
    const int lane_c0 = -x_first, lane_cl = (W - 1) - x_first;
    const bool left_border = lane_c0 >= 0 && lane_c0 + 1 < 64 && lane_c0 <= OWN1;       // strip sees columns 0..3
    const bool right_border = lane_cl <= 63 && lane_cl >= OWN0;                        // strip sees columns W-4..W-1

    float acc[7][9];      // acc[i] <-> output row q = r - 3 + i
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[k][c] = 0.f;
    float gP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gP[k] = 0.f;

    const int nrows = TH + 6;
#pragma nounroll
    for (int j = 0; j < nrows; ++j) {
        const int r = st.y_begin - 3 + j;            // coefficient row (zero outside the image)
        float g[9];
        const bool have = r >= 0 && r < H && in_img_col;
        // raw buffer loads: rows / columns outside the image and pixels another source won read zeros through an offset beyond
        // the descriptor's extent — no branch around the 10 loads of a row
        const unsigned pofs = have ? (unsigned)(r * W + st.cx) : 0x80000000u;
        const bool win = ((int)__builtin_amdgcn_raw_buffer_load_b8(idx_r, pofs, 0, 0) & 0xff) == (S + s);
        const unsigned cofs = win ? pofs * 4u : 0x80000000u;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const float v = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(coef_r, cofs + (unsigned)(c * HW) * 4u, 0, 0));
            float bx = box7(v);
            // reflection adjoint along x: columns 1..3 / W-4..W-2 also receive the mirrored window sums
            if (left_border) {
                const float c0 = __shfl(v, lane_c0 & 63, 64), c1 = __shfl(v, (lane_c0 + 1) & 63, 64),
                            c2 = __shfl(v, (lane_c0 + 2) & 63, 64);
                bx += st.cx == 1 ? (c0 + c1) + c2 : st.cx == 2 ? c0 + c1 : st.cx == 3 ? c0 : 0.f;
            }
            if (right_border) {
                const float c0 = __shfl(v, lane_cl & 63, 64), c1 = __shfl(v, (lane_cl - 1) & 63, 64),
                            c2 = __shfl(v, (lane_cl - 2) & 63, 64);
                bx += st.cx == W - 2 ? (c0 + c1) + c2 : st.cx == W - 3 ? c0 + c1 : st.cx == W - 4 ? c0 : 0.f;
            }
            g[c] = bx;
        }
        // shift the accumulator ring (row r-4 was consumed last iteration), open row q = r+3 in slot 6
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int c = 0; c < 9; ++c) acc[i][c] = acc[i + 1][c];
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[6][c] = 0.f;
        // scatter row r into the accumulator rows q = r-3+i with the reflection multiplicity
        const bool r_in = r >= 0 && r < H;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int q = r - 3 + i;
            float m = 0.f;
            if (r_in && q >= 0 && q < H) m = (float)refl_mult(r, q, H);
#pragma unroll
            for (int c = 0; c < 9; ++c) acc[i][c] = fmaf(m, g[c], acc[i][c]);
        }
        // row q = r-3 (slot 0) is complete
        const int qy = r - 3;
        if (j >= 6 && qy >= st.y_begin && qy < H && qy < st.y_begin + TH && st.own_col) {
            const int qo = qy * W + st.cx;
            const float2 gs = *reinterpret_cast<const float2 *>(smp + (size_t)qo * 2);
            // recompute taps from the stored grid exactly as the forward did
            float ix = ((gs.x + 1.0f) * 0.5f) * wm1, iy = ((gs.y + 1.0f) * 0.5f) * hm1;
            const bool mx = ix > 0.f && ix < wm1, my = iy > 0.f && iy < hm1;   // clip_coordinates_set_grad
            ix = fminf(wm1, fmaxf(ix, 0.f));
            iy = fminf(hm1, fmaxf(iy, 0.f));
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
            const float ax = ix - fx0, ay = iy - fy0, bxw = (fx0 + 1.f) - ix, byw = (fy0 + 1.f) - iy;
            // the two taps of a row are 8 contiguous bytes: one (4-byte aligned) load; at the last column the pair starts one
            // pixel earlier and the tap is its second element
            const int o00 = y0 * W + x0 - (xin ? 0 : 1), o10 = o00 + (yin ? W : 0);
            const bool l1on = idx[qo] == (uint8_t)(S + s);
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float *sc = src + c * HW;
                const f2ua pn = *reinterpret_cast<const f2ua *>(sc + o00), ps = *reinterpret_cast<const f2ua *>(sc + o10);
                const float vnw = xin ? pn.x : pn.y, vne = xin ? pn.y : 0.f, vsw = yin ? (xin ? ps.x : ps.y) : 0.f,
                            vse = (xin && yin) ? ps.y : 0.f;
                float wv = vnw * (bxw * byw);
                wv = fmaf(vne, ax * byw, wv);
                wv = fmaf(vsw, bxw * ay, wv);
                wv = fmaf(vse, ax * ay, wv);
                const float t = tgt[c * HW + qo];
                // d to_optimise / d w_c  (window terms + L1 term), trainer.py:441-453
                float gw = acc[0][c] + 2.f * wv * acc[0][3 + c] + t * acc[0][6 + c];
                if (l1on) {
                    const float df = wv - t;
                    gw += (0.15f / 3.f) * (df > 0.f ? 1.f : df < 0.f ? -1.f : 0.f);
                }
                gix += gw * ((vne - vnw) * byw + (vse - vsw) * ay);
                giy += gw * ((vsw - vnw) * bxw + (vse - vne) * ax);
            }
            // unnormalise + clamp adjoints, then (x - 0.5)*2, / (W-1)
            const float ggx = mx ? gix * (wm1 * 0.5f) : 0.f, ggy = my ? giy * (hm1 * 0.5f) : 0.f;
            const float gu = (ggx * 2.f) / wm1, gv = (ggy * 2.f) / hm1;
            // recompute the camera point
            float cr[3], X[3];
            cam_ray(ik, (float)st.cx, (float)qy, cr);
            const float d = dep[qo];
#pragma unroll
            for (int i = 0; i < 3; ++i) X[i] = d * cr[i];
            const float camz = fmaf(P[11], 1.0f, fmaf(P[10], X[2], fmaf(P[9], X[1], P[8] * X[0])));
            const float z = camz + 1e-7f;
            const float camx = fmaf(P[3], 1.0f, fmaf(P[2], X[2], fmaf(P[1], X[1], P[0] * X[0])));
            const float camy = fmaf(P[7], 1.0f, fmaf(P[6], X[2], fmaf(P[5], X[1], P[4] * X[0])));
            const float iz = 1.f / z;
            float gpx = gu * iz, gpy = gv * iz;
            float gpz = -(gu * camx + gv * camy) * iz * iz;
            gpx *= a.gscale; gpy *= a.gscale; gpz *= a.gscale;
            float gX[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) gX[k] = P[k] * gpx + P[4 + k] * gpy + P[8 + k] * gpz;
            gdep[qo] = cr[0] * gX[0] + cr[1] * gX[1] + cr[2] * gX[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gP[k] = fmaf(gpx, X[k], gP[k]);
                gP[4 + k] = fmaf(gpy, X[k], gP[4 + k]);
                gP[8 + k] = fmaf(gpz, X[k], gP[8 + k]);
            }
            gP[3] += gpx; gP[7] += gpy; gP[11] += gpz;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float v = wave_sum(gP[k]);
        if (lane == 0) a.g_P_part[(size_t)task * 12 + k] = v;
    }
}

// g_P_part [B, S, tasks_per_image, 12] -> g_P [B,S,12]   (one 64-lane block per (b,s))
__global__ __launch_bounds__(64) void gP_reduce_kernel(const float *__restrict__ part, float *__restrict__ gP, int tpi) {
    const int bs = blockIdx.x, lane = threadIdx.x;
    for (int k = 0; k < 12; ++k) {
        float s = 0.f;
        for (int t = lane; t < tpi; t += 64) s += part[((size_t)bs * tpi + t) * 12 + k];
        s = wave_sum(s);
        if (lane == 0) gP[(size_t)bs * 12 + k] = s;
    }
}

// backward: one wavefront per (image, source, strip); it runs 3 waves per SIMD, i.e. 3072 resident wavefronts on 256 CUs.
// The default strip height is the smallest one (>= 8 rows) whose task count fits in a single such round — measured at
// config B: TH = 8 (6912 tasks) 225 us, 12: 208, 16: 246, 20 (2880 tasks): 181, 24: 201, 32: 248.
int pick_th_bwd(int th, int B, int S, int H, int W) {
    if (th > 0) return th;
    const int nsx = (W + SQD_STRIP_COLS - 1) / SQD_STRIP_COLS;
    for (int t = 8; t <= 32; ++t)
        if ((long long)B * S * nsx * ((H + t - 1) / t) <= 3072) return t;
    return 8;
}

int check_shape(const char *who, int B, int S, int H, int W, int TH) {
    SQD_CHECK_ARG(S >= 1 && S <= SQD_MAX_SOURCES, "%s: S=%d source frames unsupported (1..%d)", who, S, SQD_MAX_SOURCES);
    SQD_CHECK_ARG(B > 0 && H >= 8 && W >= 8, "%s: bad shape B=%d H=%d W=%d", who, B, H, W);
    SQD_CHECK_ARG((long long)B * 9 * H * W < (1ll << 31), "%s: tensor too large for 32-bit pixel offsets", who);
    SQD_CHECK_ARG(TH >= 1 && TH <= 4096, "%s: rows_per_task=%d out of range", who, TH);
    return SQD_OK;
}
}  // namespace

extern "C" int sqd_photo_ntasks(int B, int H, int W, int rows_per_task) {
    return 4 * sqd::photo_tile_count(B, H, W, rows_per_task);       // one loss partial per wavefront of a tile
}
extern "C" int sqd_photo_bwd_ntasks(int B, int S, int H, int W, int rows_per_task) {
    const int TH = pick_th_bwd(rows_per_task, B, S, H, W);
    const int nsx = (W + SQD_STRIP_COLS - 1) / SQD_STRIP_COLS, nsy = (H + TH - 1) / TH;
    return S * B * nsx * nsy;
}

extern "C" int sqd_photo_fwd(const sqd_photo_args *a) {
    SQD_CHECK_ARG(a && a->depth && a->inv_K && a->P && a->target && a->identity, "sqd_photo_fwd: null input");
    if (check_shape("sqd_photo_fwd", a->B, a->S, a->H, a->W, 8)) return SQD_EINVAL;
    for (int s = 0; s < a->S; ++s) SQD_CHECK_ARG(a->sources[s], "sqd_photo_fwd: null source %d", s);
    SQD_CHECK_ARG(a->S <= 2 || (a->sel && a->idx), "sqd_photo_fwd: more than 2 source frames need the sel and idx outputs (running minimum)");
    (void)hipGetLastError();
    sqd::launch_photo_tile(*a, nullptr, 1, (hipStream_t)a->stream);
    SQD_CHECK_LAUNCH("sqd_photo_fwd");
    return SQD_OK;
}

extern "C" int sqd_identity_fwd(const float *target, const float *const *sources, const float *noise, float *identity,
                                int B, int S, int H, int W, int rows_per_task, void *stream) {
    SQD_CHECK_ARG(target && sources && identity, "sqd_identity_fwd: null pointer");
    if (check_shape("sqd_identity_fwd", B, S, H, W, 8)) return SQD_EINVAL;
    sqd_photo_args a = {};
    a.target = target;
    for (int s = 0; s < S; ++s) {
        SQD_CHECK_ARG(sources[s], "sqd_identity_fwd: null source %d", s);
        a.sources[s] = sources[s];
    }
    a.sel = identity;   // MODE 0 writes its [B,S,H,W] output through `sel`
    a.B = B; a.S = S; a.H = H; a.W = W; a.rows_per_task = rows_per_task;
    (void)hipGetLastError();
    sqd::launch_photo_tile(a, noise, 0, (hipStream_t)stream);
    SQD_CHECK_LAUNCH("sqd_identity_fwd");
    return SQD_OK;
}

extern "C" int sqd_photo_coef(const float *target, const float *const *warped, const uint8_t *idx, float *coef, int B, int S,
                              int H, int W, int rows_per_task, void *stream) {
    SQD_CHECK_ARG(target && warped && idx && coef, "sqd_photo_coef: null pointer");
    if (check_shape("sqd_photo_coef", B, S, H, W, 8)) return SQD_EINVAL;
    sqd_photo_args a = {};
    a.target = target;
    for (int s = 0; s < S; ++s) {
        SQD_CHECK_ARG(warped[s], "sqd_photo_coef: null warped image %d", s);
        a.warped[s] = const_cast<float *>(warped[s]);
    }
    a.idx = const_cast<uint8_t *>(idx);
    a.coef = coef;
    a.B = B; a.S = S; a.H = H; a.W = W; a.rows_per_task = rows_per_task;
    (void)hipGetLastError();
    sqd::launch_photo_tile(a, nullptr, 2, (hipStream_t)stream);
    SQD_CHECK_LAUNCH("sqd_photo_coef");
    return SQD_OK;
}

extern "C" int sqd_photo_bwd(const sqd_photo_bwd_args *a) {
    SQD_CHECK_ARG(a && a->depth && a->inv_K && a->P && a->target && a->coef && a->idx && a->g_depth && a->g_P_part, "sqd_photo_bwd: null pointer");
    for (int s = 0; a->S <= SQD_MAX_SOURCES && s < a->S; ++s) SQD_CHECK_ARG(a->sources[s] && a->sample[s], "sqd_photo_bwd: null source / sample %d", s);
    const int TH = pick_th_bwd(a->rows_per_task, a->B, a->S, a->H, a->W);
    if (check_shape("sqd_photo_bwd", a->B, a->S, a->H, a->W, TH)) return SQD_EINVAL;
    SQD_CHECK_ARG(a->g_depth_img_stride >= (int64_t)a->S * a->H * a->W, "sqd_photo_bwd: g_depth_img_stride too small");
    const int nsx = (a->W + SQD_STRIP_COLS - 1) / SQD_STRIP_COLS, nsy = (a->H + TH - 1) / TH;
    const int ntasks = a->B * a->S * nsx * nsy;
    (void)hipGetLastError();
    hipLaunchKernelGGL(photo_bwd_kernel, dim3((ntasks + 3) / 4), dim3(256), 0, (hipStream_t)a->stream, *a, TH, nsx,
                       nsy, ntasks);
    SQD_CHECK_LAUNCH("sqd_photo_bwd");
    return SQD_OK;
}

extern "C" int sqd_photo_bwd_reduce(const float *g_P_part, float *g_P, int ntasks, int tasks_per_image, int B, int S,
                                    void *stream) {
    SQD_CHECK_ARG(g_P_part && g_P && ntasks == B * S * tasks_per_image, "sqd_photo_bwd_reduce: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(gP_reduce_kernel, dim3(B * S), dim3(64), 0, (hipStream_t)stream, g_P_part, g_P, tasks_per_image);
    SQD_CHECK_LAUNCH("sqd_photo_bwd_reduce");
    return SQD_OK;
}
