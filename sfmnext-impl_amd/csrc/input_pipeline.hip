// input_pipeline.hip — the per-frame preprocessing of the reference's MonoDataset (datasets/mono_dataset.py:90-201) on the device,
// byte-exact with what PIL / torchvision compute on the host there:
//   sqd_resample_h_u8 / _v_u8 : Image.resize((W, H), Image.ANTIALIAS) — Pillow's two-pass Lanczos resample with 22-bit fixed-point
//                               coefficients and an 8-bit intermediate image (libImaging/Resample.c); the horizontal pass can mirror
//                               the frame first (the `do_flip` branch, mono_dataset.py:163)
//   sqd_luma_sum_u8           : per-frame sum of the ITU-R 601 luma (ImageEnhance.Contrast's degenerate grey level)
//   sqd_color_jitter_step_u8  : one step of torchvision's ColorJitter on PIL images — brightness / contrast / saturation as
//                               ImageEnhance blends (libImaging/Blend.c arithmetic), hue through Pillow's RGB <-> HSV (Convert.c)
//   sqd_u8_to_chw_f32         : ToTensor ([H,W,3] bytes -> [3,H,W] float = v / 255)
// Bytes in, bytes out, integer / float32 arithmetic in the order of the C sources: HBM-bound element-wise work, no matrix cores.
// Not on the training step proper: it keeps a real KITTI feed off the host's cores.
#include "sqd_common.h"

namespace {
using namespace sqd;
constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int acc) {
    const int v = acc >> PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// out[f][y][xx][c] = clip8(sum_k in[f][y][x0 + k][c] * coef[xx][k] + half); flip[f]: source column W0 - 1 - x
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                                         const int *__restrict__ bounds, const int *__restrict__ coef, int ksize, int n,
                                                         int H0, int W0, int W, const unsigned char *__restrict__ flip) {
    const size_t total = (size_t)n * H0 * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xx = (int)(i % W);
        const size_t row = i / W;                                  // f * H0 + y
        const int f = (int)(row / H0);
        const int x0 = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const bool fl = flip && flip[f];
        const unsigned char *src = in + row * (size_t)W0 * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int k = 0; k < cnt; ++k) {
            const int x = fl ? W0 - 1 - (x0 + k) : x0 + k;
            const int w = coef[(size_t)xx * ksize + k];
            a0 += src[x * 3] * w; a1 += src[x * 3 + 1] * w; a2 += src[x * 3 + 2] * w;
        }
        unsigned char *o = out + i * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                                         const int *__restrict__ bounds, const int *__restrict__ coef, int ksize, int n,
                                                         int H0, int H, int W) {
    const size_t total = (size_t)n * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W);
        const size_t t2 = i / W;
        const int yy = (int)(t2 % H), f = (int)(t2 / H);
        const int y0 = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const unsigned char *src = in + ((size_t)f * H0 * W + x) * 3;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int k = 0; k < cnt; ++k) {
            const unsigned char *p = src + (size_t)(y0 + k) * W * 3;
            const int w = coef[(size_t)yy * ksize + k];
            a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
        }
        unsigned char *o = out + i * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }     // Convert.c L24

// sums [f] += luma over the frame (integer: order-free, exact); one workgroup per (chunk, frame)
__global__ __launch_bounds__(256) void luma_sum_kernel(const unsigned char *__restrict__ img, unsigned long long *__restrict__ sums, int HW) {
    const int f = blockIdx.y;
    const unsigned char *p = img + (size_t)f * HW * 3;
    unsigned long long s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) s += (unsigned long long)luma(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[f], s);
}

// Blend.c ImagingBlend(in1 = degenerate, in2 = image, alpha): float arithmetic, truncation; outside [0, 1] clipped first
__device__ __forceinline__ unsigned char blend(int deg, int v, float alpha, bool inside) {
    const float t = (float)deg + alpha * (float)(v - deg);
    if (inside) return (unsigned char)(int)t;
    return t <= 0.f ? 0 : t >= 255.f ? 255 : (unsigned char)(int)t;
}

// Convert.c rgb2hsv_row / hsv2rgb (float variables, double literals — the promotions are spelled out)
__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int &uh, int &us, int &uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) { uh = 0; us = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    const int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
    uh = ih < 0 ? 0 : ih > 255 ? 255 : ih;
    us = is < 0 ? 0 : is > 255 ? 255 : is;
}
__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int &r, int &g, int &b) {
    if (s == 0) { r = g = b = v; return; }
    const double hf = (double)(float)h * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)s / 255.0);
    const double vf = (double)(float)v;
    auto rnd = [](double x) { const int q = (int)floor(x + 0.5); return q < 0 ? 0 : q > 255 ? 255 : q; };       // round() of a value >= 0
    const int p = rnd(vf * (1.0 - (double)fs)), q = rnd(vf * (1.0 - (double)fs * (double)f)),
              t = rnd(vf * (1.0 - (double)fs * (1.0 - (double)f)));
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// one step of ColorJitter for every frame: op[f] in {0 brightness, 1 contrast, 2 saturation, 3 hue, anything else: copy};
// factor[f]: the blend factor (ops 0-2); hshift[f]: np.uint8(hue_factor * 255) (op 3); lsum[f]: the frame's luma sum (op 1)
__global__ __launch_bounds__(256) void jitter_step_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                                          const int *__restrict__ op, const float *__restrict__ factor,
                                                          const int *__restrict__ hshift, const unsigned long long *__restrict__ lsum,
                                                          int n, int HW) {
    const size_t total = (size_t)n * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int f = (int)(i / HW);
        const unsigned char *p = in + i * 3;
        int r = p[0], g = p[1], b = p[2];
        const int o = op[f];
        const float a = factor[f];
        const bool inside = a >= 0.f && a <= 1.f;
        if (o == 0) {
            r = blend(0, r, a, inside); g = blend(0, g, a, inside); b = blend(0, b, a, inside);
        } else if (o == 1) {
            const int m = (int)((double)lsum[f] / (double)HW + 0.5);       // int(ImageStat.Stat(L).mean[0] + 0.5)
            r = blend(m, r, a, inside); g = blend(m, g, a, inside); b = blend(m, b, a, inside);
        } else if (o == 2) {
            const int l = luma(r, g, b);
            r = blend(l, r, a, inside); g = blend(l, g, a, inside); b = blend(l, b, a, inside);
        } else if (o == 3) {
            int h, s, v;
            rgb2hsv(r, g, b, h, s, v);
            h = (h + hshift[f]) & 255;
            hsv2rgb(h, s, v, r, g, b);
        }
        unsigned char *q = out + i * 3;
        q[0] = (unsigned char)r; q[1] = (unsigned char)g; q[2] = (unsigned char)b;
    }
}

__global__ __launch_bounds__(256) void to_tensor_kernel(const unsigned char *__restrict__ in, float *__restrict__ out, int n, int HW) {
    const size_t total = (size_t)n * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t f = i / HW, px = i % HW;
        const unsigned char *p = in + i * 3;
        float *o = out + f * 3 * HW + px;
        o[0] = (float)p[0] / 255.0f; o[HW] = (float)p[1] / 255.0f; o[2 * (size_t)HW] = (float)p[2] / 255.0f;
    }
}

int grid_for(size_t n) {
    const size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 16384 ? 16384 : b);
}
}  // namespace

// in [n,H0,W0,3] -> out [n,H0,W,3]; bounds [W,2] (first source column, taps), coef [W,ksize] int32 (22-bit fixed point) — the
// tables of Pillow's precompute_coeffs / normalize_coeffs_8bpc for (W0 -> W), computed by the caller; flip [n] bytes or NULL
extern "C" int sqd_resample_h_u8(const unsigned char *in, unsigned char *out, const int *bounds, const int *coef, int ksize, int n, int H0,
                                 int W0, int W, const unsigned char *flip, void *stream) {
    SQD_CHECK_ARG(in && out && bounds && coef && ksize > 0 && n > 0 && H0 > 0 && W0 > 0 && W > 0, "sqd_resample_h_u8: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((size_t)n * H0 * W)), dim3(256), 0, (hipStream_t)stream, in, out, bounds, coef, ksize, n,
                       H0, W0, W, flip);
    SQD_CHECK_LAUNCH("sqd_resample_h_u8");
    return SQD_OK;
}
// in [n,H0,W,3] -> out [n,H,W,3]; bounds [H,2], coef [H,ksize] for (H0 -> H)
extern "C" int sqd_resample_v_u8(const unsigned char *in, unsigned char *out, const int *bounds, const int *coef, int ksize, int n, int H0,
                                 int H, int W, void *stream) {
    SQD_CHECK_ARG(in && out && bounds && coef && ksize > 0 && n > 0 && H0 > 0 && H > 0 && W > 0, "sqd_resample_v_u8: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for((size_t)n * H * W)), dim3(256), 0, (hipStream_t)stream, in, out, bounds, coef, ksize, n,
                       H0, H, W);
    SQD_CHECK_LAUNCH("sqd_resample_v_u8");
    return SQD_OK;
}
// img [n,H,W,3] -> sums [n] uint64 += luma (the caller zeroes sums)
extern "C" int sqd_luma_sum_u8(const unsigned char *img, unsigned long long *sums, int n, int HW, void *stream) {
    SQD_CHECK_ARG(img && sums && n > 0 && HW > 0, "sqd_luma_sum_u8: bad arguments");
    const int chunks = HW / 4096 < 1 ? 1 : HW / 4096 > 64 ? 64 : HW / 4096;
    (void)hipGetLastError();
    hipLaunchKernelGGL(luma_sum_kernel, dim3(chunks, n), dim3(256), 0, (hipStream_t)stream, img, sums, HW);
    SQD_CHECK_LAUNCH("sqd_luma_sum_u8");
    return SQD_OK;
}
// one ColorJitter step: in, out [n,H,W,3]; op, hshift [n] int32, factor [n] float, lsum [n] uint64 (read for op 1 only)
extern "C" int sqd_color_jitter_step_u8(const unsigned char *in, unsigned char *out, const int *op, const float *factor, const int *hshift,
                                        const unsigned long long *lsum, int n, int HW, void *stream) {
    SQD_CHECK_ARG(in && out && op && factor && hshift && lsum && n > 0 && HW > 0, "sqd_color_jitter_step_u8: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(jitter_step_kernel, dim3(grid_for((size_t)n * HW)), dim3(256), 0, (hipStream_t)stream, in, out, op, factor, hshift, lsum,
                       n, HW);
    SQD_CHECK_LAUNCH("sqd_color_jitter_step_u8");
    return SQD_OK;
}
// in [n,H,W,3] bytes -> out [n,3,H,W] float = v / 255
extern "C" int sqd_u8_to_chw_f32(const unsigned char *in, float *out, int n, int HW, void *stream) {
    SQD_CHECK_ARG(in && out && n > 0 && HW > 0, "sqd_u8_to_chw_f32: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(to_tensor_kernel, dim3(grid_for((size_t)n * HW)), dim3(256), 0, (hipStream_t)stream, in, out, n, HW);
    SQD_CHECK_LAUNCH("sqd_u8_to_chw_f32");
    return SQD_OK;
}
