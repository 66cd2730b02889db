// photometric.hip — entry points of the photometric chain (include/sqd.h section 3) and its backward kernel.
//
// The kernels live in photo_tile.hip (fused warp + SSIM forward, identity maps, coefficient planes, the tile backward);
// here: argument checks, the launches and the small reduction of the g_P partials.
//
// Roofline: HBM.  Algorithmic bytes per target pixel (S = 2): backward 84 B + the 36 B/px coefficient
// planes photo_coef writes and the backward kernel reads.
#include "photo_launch.h"

namespace {
using namespace sqd;

typedef float f2ua __attribute__((ext_vector_type(2), aligned(4)));      // 8-byte load at 4-byte alignment
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
constexpr float INV49 = 1.0f / 49.0f;
constexpr int OWN0 = 3, OWN1 = 61;   // owned lanes [3, 61): 58 output columns per wavefront

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

int check_shape(const char *who, int B, int S, int H, int W, int TH) {
    SQD_CHECK_ARG(S >= 1 && S <= SQD_MAX_SOURCES, "%s: S=%d source frames unsupported (1..%d)", who, S, SQD_MAX_SOURCES);
    SQD_CHECK_ARG(B > 0 && H >= 8 && W >= 8, "%s: bad shape B=%d H=%d W=%d", who, B, H, W);
    SQD_CHECK_ARG((long long)B * 9 * H * W < (1ll << 31), "%s: tensor too large for 32-bit pixel offsets", who);
    SQD_CHECK_ARG(TH >= 1 && TH <= 4096, "%s: rows_per_task=%d out of range", who, TH);
    return SQD_OK;
}
}  // namespace

extern "C" int sqd_photo_ntasks(int B, int H, int W, int rows_per_task) {
    return sqd::photo_fwd_waves(B, H, W, rows_per_task) * sqd::photo_tile_count(B, H, W, rows_per_task, 1);       // one loss partial per wavefront of a tile
}
extern "C" int sqd_photo_bwd_ntasks(int B, int S, int H, int W, int rows_per_task) {
    return S * 4 * sqd::photo_tile_count(B, H, W, rows_per_task, 2);      // one g_P partial per wavefront of a tile and source
}

static int check_loss_flags(const char *who, int flags, int S) {
    SQD_CHECK_ARG((flags & ~(7 | SQD_SOURCES_HWC)) == 0, "%s: unknown loss_flags %d", who, flags);
    SQD_CHECK_ARG(!(flags & SQD_SOURCES_HWC) || ((flags & 7) == 0 && S == 2), "%s: SQD_SOURCES_HWC goes with the default loss options and two source frames (loss_flags %d, S=%d)", who, flags, S);
    SQD_CHECK_ARG(!(flags & SQD_LOSS_AVG_REPROJECTION) || S >= 2, "%s: avg_reprojection needs at least two source frames (the mean over one is that frame: drop the flag; S=%d)", who, S);
    return SQD_OK;
}

namespace sqd { void photo_set_fwd_variant(int v); }
extern "C" int sqd_photo_set_fwd_variant(int variant) {
    SQD_CHECK_ARG((variant & 0x3f) <= 6 && variant >= 0, "sqd_photo_set_fwd_variant: bits 0-5 take 0 .. 6 (include/sqd.h), got %d", variant);
    sqd::photo_set_fwd_variant(variant);
    return SQD_OK;
}

extern "C" int sqd_photo_fwd(const sqd_photo_args *a) {
    SQD_CHECK_ARG(a && a->depth && a->inv_K && a->P && a->target && (a->identity || (a->loss_flags & SQD_LOSS_NO_AUTOMASK)), "sqd_photo_fwd: null input");
    if (check_loss_flags("sqd_photo_fwd", a->loss_flags, a->S)) return SQD_EINVAL;
    if (check_shape("sqd_photo_fwd", a->B, a->S, a->H, a->W, 8)) return SQD_EINVAL;
    for (int s = 0; s < a->S; ++s) SQD_CHECK_ARG(a->sources[s], "sqd_photo_fwd: null source %d", s);
    SQD_CHECK_ARG(a->S <= 2 || (a->sel && a->idx), "sqd_photo_fwd: more than 2 source frames need the sel and idx outputs (running minimum)");
    (void)hipGetLastError();
    if (sqd::launch_photo_tile(*a, nullptr, 1, (hipStream_t)a->stream)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_photo_fwd");
    return SQD_OK;
}

extern "C" int sqd_identity_fwd(const float *target, const float *const *sources, const float *noise, float *identity,
                                int B, int S, int H, int W, int rows_per_task, void *stream) {
    return sqd_identity_fwd_ex(target, sources, noise, identity, B, S, H, W, rows_per_task, 0, stream);
}

extern "C" int sqd_identity_fwd_ex(const float *target, const float *const *sources, const float *noise, float *identity,
                                   int B, int S, int H, int W, int rows_per_task, int loss_flags, void *stream) {
    SQD_CHECK_ARG(target && sources && identity, "sqd_identity_fwd: null pointer");
    if (check_loss_flags("sqd_identity_fwd", loss_flags, S)) return SQD_EINVAL;
    if (check_shape("sqd_identity_fwd", B, S, H, W, 8)) return SQD_EINVAL;
    sqd_photo_args a = {};
    a.target = target;
    for (int s = 0; s < S; ++s) {
        SQD_CHECK_ARG(sources[s], "sqd_identity_fwd: null source %d", s);
        a.sources[s] = sources[s];
    }
    a.sel = identity;   // MODE 0 writes its [B,S,H,W] output through `sel`
    a.B = B; a.S = S; a.H = H; a.W = W; a.rows_per_task = rows_per_task; a.loss_flags = loss_flags;
    (void)hipGetLastError();
    if (sqd::launch_photo_tile(a, noise, 0, (hipStream_t)stream)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_identity_fwd");
    return SQD_OK;
}

// ---- pixel-interleaved copies of the source frames (SQD_SOURCES_HWC) -------------------------------------------------------------------
// [B,3,H,W] -> [B,H,W,3]: a thread takes four pixels — one 16-byte load per colour plane, three 16-byte stores (48 consecutive bytes).  HBM:
// 24 bytes per pixel and frame.  n frames per launch (blockIdx.y).
namespace {
struct PackPtrs {
    const float *in[SQD_MAX_SOURCES];
    float *out[SQD_MAX_SOURCES];
};
__global__ __launch_bounds__(256) void pack_pixels_kernel(PackPtrs p, int HW4, int B) {
    const float *__restrict__ in = p.in[blockIdx.y];
    float *__restrict__ out = p.out[blockIdx.y];
    const size_t n = (size_t)B * HW4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / HW4, q = i - b * HW4;
        const float4 *src = reinterpret_cast<const float4 *>(in + b * 3 * 4 * (size_t)HW4) + q;
        const float4 r = src[0], g = src[HW4], bl = src[2 * (size_t)HW4];
        float4 *dst = reinterpret_cast<float4 *>(out) + i * 3;
        dst[0] = make_float4(r.x, g.x, bl.x, r.y);
        dst[1] = make_float4(g.y, bl.y, r.z, g.z);
        dst[2] = make_float4(bl.z, r.w, g.w, bl.w);
    }
}
}  // namespace

extern "C" int sqd_photo_sources_hwc_ok(int B, int S, int H, int W, int rows_per_task, int loss_flags) {
    return B > 0 && H > 0 && W > 0 && sqd::photo_sources_hwc_ok(B, S, H, W, rows_per_task, loss_flags) ? 1 : 0;
}

extern "C" int sqd_pack_pixels(const float *const *planar, float *const *px, int n, int B, int H, int W, void *stream) {
    SQD_CHECK_ARG(planar && px, "sqd_pack_pixels: null pointer");
    SQD_CHECK_ARG(n >= 1 && n <= SQD_MAX_SOURCES, "sqd_pack_pixels: %d frames (1..%d)", n, SQD_MAX_SOURCES);
    SQD_CHECK_ARG(B > 0 && H > 0 && W > 0 && (size_t)H * W % 4 == 0 && (size_t)B * 3 * H * W < (1ull << 31), "sqd_pack_pixels: bad shape %dx3x%dx%d (H*W a multiple of 4)", B, H, W);
    PackPtrs p = {};
    for (int i = 0; i < n; ++i) {
        SQD_CHECK_ARG(planar[i] && px[i], "sqd_pack_pixels: null frame %d", i);
        p.in[i] = planar[i];
        p.out[i] = px[i];
    }
    const int HW4 = H * W / 4;
    const size_t total = (size_t)B * HW4;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    (void)hipGetLastError();
    hipLaunchKernelGGL(pack_pixels_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, p, HW4, B);
    SQD_CHECK_LAUNCH("sqd_pack_pixels");
    return SQD_OK;
}

extern "C" int sqd_photo_coef(const float *target, const float *const *warped, const uint8_t *idx, float *coef, int B, int S,
                              int H, int W, int rows_per_task, void *stream) {
    return sqd_photo_coef_ex(target, warped, idx, coef, B, S, H, W, rows_per_task, 0, stream);
}

extern "C" int sqd_photo_coef_ex(const float *target, const float *const *warped, const uint8_t *idx, float *coef, int B, int S,
                                 int H, int W, int rows_per_task, int loss_flags, void *stream) {
    SQD_CHECK_ARG(target && warped && idx && coef, "sqd_photo_coef: null pointer");
    if (check_loss_flags("sqd_photo_coef", loss_flags, S)) return SQD_EINVAL;
    if (check_shape("sqd_photo_coef", B, S, H, W, 8)) return SQD_EINVAL;
    sqd_photo_args a = {};
    a.target = target;
    for (int s = 0; s < S; ++s) {
        SQD_CHECK_ARG(warped[s], "sqd_photo_coef: null warped image %d", s);
        a.warped[s] = const_cast<float *>(warped[s]);
    }
    a.idx = const_cast<uint8_t *>(idx);
    a.coef = coef;
    a.B = B; a.S = S; a.H = H; a.W = W; a.rows_per_task = rows_per_task; a.loss_flags = loss_flags;
    (void)hipGetLastError();
    if (sqd::launch_photo_tile(a, nullptr, 2, (hipStream_t)stream)) return SQD_EINVAL;
    SQD_CHECK_LAUNCH("sqd_photo_coef");
    return SQD_OK;
}

extern "C" int sqd_photo_bwd(const sqd_photo_bwd_args *a) {
    SQD_CHECK_ARG(a && a->depth && a->inv_K && a->P && a->target && a->coef && a->idx && a->g_depth && a->g_P_part, "sqd_photo_bwd: null pointer");
    SQD_CHECK_ARG(a->S >= 1 && a->S <= SQD_MAX_SOURCES, "sqd_photo_bwd: S=%d source frames unsupported (1..%d)", a->S, SQD_MAX_SOURCES);
    for (int s = 0; s < a->S; ++s) SQD_CHECK_ARG(a->sources[s] && a->sample[s], "sqd_photo_bwd: null source / sample %d", s);
    if (check_shape("sqd_photo_bwd", a->B, a->S, a->H, a->W, 8)) return SQD_EINVAL;
    if (check_loss_flags("sqd_photo_bwd", a->loss_flags, a->S)) return SQD_EINVAL;
    SQD_CHECK_ARG(a->rows_per_task >= 0, "sqd_photo_bwd: rows_per_task=%d", a->rows_per_task);
    SQD_CHECK_ARG(!(a->loss_flags & SQD_SOURCES_HWC) || a->W >= 64, "sqd_photo_bwd: SQD_SOURCES_HWC needs W >= 64 (W=%d)", a->W);
    SQD_CHECK_ARG(a->g_depth_img_stride >= (int64_t)((a->S + 1) / 2) * a->H * a->W, "sqd_photo_bwd: g_depth_img_stride too small");
    (void)hipGetLastError();
    sqd::launch_photo_bwd_tile(*a, (hipStream_t)a->stream);
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
