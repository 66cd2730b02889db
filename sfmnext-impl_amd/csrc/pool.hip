// pool.hip — MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem, channels-last, forward + backward.
// replaces: self.encoder.maxpool (torchvision ResNet, called from reference networks/resnet_encoder.py:96).
// forward : y[n,ho,wo,c] = max over the 3x3 window at (2ho-1, 2wo-1) (positions outside the image are skipped); the window
//           position of the maximum (0..8, first maximum in row-major scan order — ATen's tie rule) goes to idx (1 byte).
// backward: gather form — every input pixel looks at the (at most 2x2) windows that contain it and takes dy where idx
//           names it.  No atomics: deterministic, unlike ATen's scatter (max_pool_backward_nhwc).
// Roofline: HBM — forward 4 B read + 1.25 B written per input element, backward 1.25 B read + 4 B written.
#include "sqd_common.h"

namespace {
using namespace sqd;

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                          unsigned char *__restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
    const int V = C / 4;
    const size_t total = (size_t)N * Ho * Wo * V;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % V);
        size_t t = i / V;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho), n = (int)(t / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int4 k = make_int4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if (hi < 0 || hi >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if (wi < 0 || wi >= W) continue;
                const float4 v = *reinterpret_cast<const float4 *>(x + (((size_t)n * H + hi) * W + wi) * C + cg * 4);
                const int p = r * 3 + s;
                // ATen: (val > maxval) || isnan(val)
                if (v.x > m.x || v.x != v.x) { m.x = v.x; k.x = p; }
                if (v.y > m.y || v.y != v.y) { m.y = v.y; k.y = p; }
                if (v.z > m.z || v.z != v.z) { m.z = v.z; k.z = p; }
                if (v.w > m.w || v.w != v.w) { m.w = v.w; k.w = p; }
            }
        }
        reinterpret_cast<float4 *>(y)[i] = m;
        reinterpret_cast<uchar4 *>(idx)[i] = make_uchar4((unsigned char)k.x, (unsigned char)k.y, (unsigned char)k.z, (unsigned char)k.w);
    }
}

// xb != NULL (launched with 256 % (C / 4) == 0: a thread keeps its channel group): dx is the whole gradient of the BatchNorm + activation that
// produced the pooled tensor (the ResNet stem's bn1 + ReLU; its other consumer's gradient arrives as `addend`) — that node's two backward sums,
// (sum dz, sum dz * xhat) with dz = dx * act'(.), are taken here, one partial row [C][2] per workgroup, instead of by a reduction pass of their
// own over dx and xb: xb [N,H,W,C] the BatchNorm's input, maskb the sign bits of its pre-activation (1 byte per 4 elements; NULL: no activation),
// slope = the activation's derivative on the negative side (ReLU 0, LeakyReLU 0.01)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ dy, const unsigned char *__restrict__ idx,
                                                          const float *__restrict__ addend, float *__restrict__ dx, int N, int H, int W,
                                                          int C, int Ho, int Wo, const float *__restrict__ xb = nullptr,
                                                          const unsigned char *__restrict__ maskb = nullptr, const float *__restrict__ meanb = nullptr,
                                                          const float *__restrict__ rstdb = nullptr, float slope = 0.f, float *__restrict__ partb = nullptr) {
    __shared__ float4 red[2][256];
    const int V = C / 4;
    const size_t total = (size_t)N * H * W * V;
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), qb = sb, mub = sb, rsb = sb;
    if (xb) {
        mub = reinterpret_cast<const float4 *>(meanb)[threadIdx.x % V];
        rsb = reinterpret_cast<const float4 *>(rstdb)[threadIdx.x % V];
    }
    // a thread takes a 2 x 2 quad of input pixels (rows 2a, 2a+1, columns 2b, 2b+1) of one channel group: the four windows that can hold them —
    // (a, b) all four, (a, b+1) the odd column, (a+1, b) the odd row, (a+1, b+1) the odd-odd pixel — are fetched ONCE (until round 6: per pixel, nine
    // window fetches per quad: 40.5 / 44.3 / 68.3 us without / with the skip gradient / with the BatchNorm sums at 12 x 64 x 96 x 320 against 24.0 /
    // 37.1 / 58.0 now, tools/bench_pool.py); per pixel the same terms in the same order as before (the other consumer's gradient, then the windows
    // by row, column)
    const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    const size_t quads = (size_t)N * H2 * W2 * V;
    (void)total;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
        const int cg = (int)(q % V);
        size_t t = q / V;
        const int b2 = (int)(t % W2);
        t /= W2;
        const int a2 = (int)(t % H2), n = (int)(t / H2);
        uchar4 k[2][2];
        float4 v[2][2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int ho = a2 + dh, wo = b2 + dw;
                if (ho < Ho && wo < Wo) {
                    const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * V + cg;
                    k[dh][dw] = reinterpret_cast<const uchar4 *>(idx)[o];
                    v[dh][dw] = reinterpret_cast<const float4 *>(dy)[o];
                } else {
                    k[dh][dw] = make_uchar4(255, 255, 255, 255);        // (no window: matches no position)
                    v[dh][dw] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int pw = 0; pw < 2; ++pw) {
                const int hi = 2 * a2 + ph, wi = 2 * b2 + pw;
                if (hi >= H || wi >= W) continue;
                const size_t i = (((size_t)n * H + hi) * W + wi) * V + cg;
                float4 g = addend ? reinterpret_cast<const float4 *>(addend)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                // windows holding row hi: ho = a (position row 1 for an even row, 2 for an odd one) and, for an odd row, ho = a + 1 (position row 0)
#pragma unroll
                for (int dh = 0; dh <= ph; ++dh)
#pragma unroll
                    for (int dw = 0; dw <= pw; ++dw) {
                        const int p = (ph + 1 - 2 * dh) * 3 + (pw + 1 - 2 * dw);
                        const uchar4 kk = k[dh][dw];
                        const float4 vv = v[dh][dw];
                        if (kk.x == p) g.x += vv.x;
                        if (kk.y == p) g.y += vv.y;
                        if (kk.z == p) g.z += vv.z;
                        if (kk.w == p) g.w += vv.w;
                    }
                reinterpret_cast<float4 *>(dx)[i] = g;
                if (xb) {
                    const unsigned bits = maskb ? maskb[i] : 0xfu;
                    const float4 xv = reinterpret_cast<const float4 *>(xb)[i];
                    const float d0 = g.x * ((bits & 1u) ? 1.f : slope), d1 = g.y * ((bits & 2u) ? 1.f : slope), d2 = g.z * ((bits & 4u) ? 1.f : slope),
                                d3 = g.w * ((bits & 8u) ? 1.f : slope);
                    sb.x += d0; sb.y += d1; sb.z += d2; sb.w += d3;
                    qb.x = fmaf(d0, (xv.x - mub.x) * rsb.x, qb.x); qb.y = fmaf(d1, (xv.y - mub.y) * rsb.y, qb.y);
                    qb.z = fmaf(d2, (xv.z - mub.z) * rsb.z, qb.z); qb.w = fmaf(d3, (xv.w - mub.w) * rsb.w, qb.w);
                }
            }
    }
    if (xb) {      // (uniform) the workgroup's partial row: the threads of a channel group added in a fixed order
        red[0][threadIdx.x] = sb;
        red[1][threadIdx.x] = qb;
        __syncthreads();
        if ((int)threadIdx.x < V) {
            for (int k = threadIdx.x + V; k < 256; k += V) {
                const float4 u = red[0][k], v = red[1][k];
                sb.x += u.x; sb.y += u.y; sb.z += u.z; sb.w += u.w;
                qb.x += v.x; qb.y += v.y; qb.z += v.z; qb.w += v.w;
            }
            float *o = partb + ((size_t)blockIdx.x * C + threadIdx.x * 4) * 2;
            reinterpret_cast<float4 *>(o)[0] = make_float4(sb.x, qb.x, sb.y, qb.y);
            reinterpret_cast<float4 *>(o)[1] = make_float4(sb.z, qb.z, sb.w, qb.w);
        }
    }
}

int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : b);
}
}  // namespace

// x [N,H,W,C] -> y [N,Ho,Wo,C], idx [N,Ho,Wo,C] bytes; Ho = (H - 1) / 2 + 1, Wo likewise; C multiple of 4
extern "C" int sqd_maxpool3x3s2_fwd(const float *x, float *y, unsigned char *idx, int N, int H, int W, int C, void *stream) {
    SQD_CHECK_ARG(x && y && idx && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "sqd_maxpool3x3s2_fwd: bad arguments (C=%d)", C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((size_t)N * Ho * Wo * C / 4)), dim3(256), 0, (hipStream_t)stream, x, y, idx, N, H, W,
                       C, Ho, Wo);
    SQD_CHECK_LAUNCH("sqd_maxpool3x3s2_fwd");
    return SQD_OK;
}

// dy [N,Ho,Wo,C], idx from the forward -> dx [N,H,W,C] (fully overwritten) = gather(dy) [+ addend [N,H,W,C], may be NULL: the
// gradient of the input's other consumer, folded in here instead of a separate accumulation pass]
extern "C" int sqd_maxpool3x3s2_bwd(const float *dy, const unsigned char *idx, const float *addend, float *dx, int N, int H, int W, int C,
                                    void *stream) {
    SQD_CHECK_ARG(dy && idx && dx && N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0, "sqd_maxpool3x3s2_bwd: bad arguments (C=%d)", C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C / 4)), dim3(256), 0, (hipStream_t)stream, dy, idx, addend,
                       dx, N, H, W, C, Ho, Wo);                 // (a thread per 2 x 2 quad of input pixels and channel group)
    SQD_CHECK_LAUNCH("sqd_maxpool3x3s2_bwd");
    return SQD_OK;
}

// the partial rows sqd_maxpool3x3s2_bwd_bn writes at this shape (0: not served — the channel groups must divide a workgroup's 256 threads)
extern "C" int sqd_maxpool3x3s2_bwd_bn_rows(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4 || 256 % (C / 4)) return 0;
    const int g = grid_for((size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C / 4);
    return g < 2048 ? g : 2048;
}

// sqd_maxpool3x3s2_bwd that also takes the two BatchNorm-backward sums of the node that produced the pooled tensor x = act(BatchNorm(xb)): dx is
// that node's whole incoming gradient.  xb [N,H,W,C], maskb (may be NULL with act 0) the sign bits its forward stored, meanb / rstdb [C] its saved
// statistics, act 0 none / 1 ReLU / 2 LeakyReLU(0.01) -> partb [sqd_maxpool3x3s2_bwd_bn_rows(N,H,W,C)][C][2] = (sum dz, sum dz * xhat), the
// precomputed partials of its sqd_bn_train_bwd_pre
extern "C" int sqd_maxpool3x3s2_bwd_bn(const float *dy, const unsigned char *idx, const float *addend, float *dx, int N, int H, int W, int C,
                                       const float *xb, const unsigned char *maskb, const float *meanb, const float *rstdb, int act, float *partb,
                                       void *stream) {
    SQD_CHECK_ARG(dy && idx && dx && xb && meanb && rstdb && partb, "sqd_maxpool3x3s2_bwd_bn: null pointer");
    SQD_CHECK_ARG(act == 0 || ((act == 1 || act == 2) && maskb), "sqd_maxpool3x3s2_bwd_bn: act %d (0 none, 1 ReLU, 2 LeakyReLU: these with the sign mask)", act);
    const int rows = sqd_maxpool3x3s2_bwd_bn_rows(N, H, W, C);
    SQD_CHECK_ARG(rows > 0, "sqd_maxpool3x3s2_bwd_bn: shape %dx%dx%dx%d not served (C / 4 must divide 256)", N, H, W, C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dy, idx, addend, dx, N, H, W, C, Ho, Wo, xb, act ? maskb : nullptr,
                       meanb, rstdb, act == 2 ? 0.01f : 0.f, partb);
    SQD_CHECK_LAUNCH("sqd_maxpool3x3s2_bwd_bn");
    return SQD_OK;
}

// ---------------------------------------------------------------------------------------------------
// space-to-depth(2) of a channels-last image, channel count padded with zeros:
//   y[n, h2, w2, c*4 + dy*2 + dx] = x[n, 2*h2 + dy, 2*w2 + dx, c]   (torch.nn.functional.pixel_unshuffle's channel order)
// feeds the 7x7/2 stems, which run as 4x4/1 convolutions on this layout (nnkernels.conv2d_stem_s2d).
// ---------------------------------------------------------------------------------------------------
namespace {
// one thread = 4 consecutive output channels (one float4 store); reads stay inside two 2C-float runs of the image rows
__global__ __launch_bounds__(256) void s2d_kernel(const float *__restrict__ x, float *__restrict__ y, int N, int H, int W, int C, int Cp) {
    const int H2 = H / 2, W2 = W / 2, Q = Cp / 4;
    const size_t total = (size_t)N * H2 * W2 * Q;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Q);                       // output channels 4c..4c+3 = input channel c at (dy,dx) = (0,0),(0,1),(1,0),(1,1)
        size_t t = i / Q;
        const int w2 = (int)(t % W2);
        t /= W2;
        const int h2 = (int)(t % H2), n = (int)(t / H2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
            const float *r0 = x + (((size_t)n * H + 2 * h2) * W + 2 * w2) * C + c, *r1 = r0 + (size_t)W * C;
            v = make_float4(r0[0], r0[C], r1[0], r1[C]);
        }
        reinterpret_cast<float4 *>(y)[i] = v;
    }
}

// the same from planar (NCHW) sources, with the input normalisation and the channel concatenation of two frames folded in:
//   channel c < C0 from x0[n, c], c >= C0 from x1[n, c - C0];  value (v - sub) * (1 / div) — a tensor divided by a scalar is a
//   multiplication by the scalar's float reciprocal in ATen's CUDA/HIP kernels, which is what the reference runs;  sample n lands at
//   y + n * y_stride
//   One thread per OUTPUT pixel: its Cp floats are Cp/4 consecutive float4 stores (a wave writes 64 x Cp x 4 contiguous bytes) and
//   every plane is read as two runs of 128 consecutive floats per wave.  (Round 2's version gave a thread one (pixel, channel): a
//   wave's stores were 16 bytes out of every Cp x 4 — 40 us per call for 41 MB at config B.)
template <int QMAX>                                      // Cp / 4 <= QMAX float4 per pixel
__global__ __launch_bounds__(256) void s2d_planar_kernel(const float *__restrict__ x0, const float *__restrict__ x1, float *__restrict__ y,
                                                         int N, int H, int W, int C0, int C1, int Cp, long long y_stride, float sub, float inv_div,
                                                         unsigned *__restrict__ amax_y) {
    unsigned am = 0u;                                    // max |y| over this thread's values (amax_y == NULL: not recorded)
    const int H2 = H / 2, W2 = W / 2, Q = Cp / 4, C = C0 + C1;
    const size_t per = (size_t)H2 * W2, total = (size_t)N * per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int w2 = (int)(i % W2);
        const size_t t = i / W2;
        const int h2 = (int)(t % H2), n = (int)(t / H2);
        const size_t o = (size_t)(2 * h2) * W + 2 * w2;
        float4 v[QMAX];
#pragma unroll
        for (int c = 0; c < QMAX; ++c) {
            v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < C) {
                const float *pl = c < C0 ? x0 + ((size_t)n * C0 + c) * H * W : x1 + ((size_t)n * C1 + (c - C0)) * H * W;
                const float2 a = *reinterpret_cast<const float2 *>(pl + o);
                const float2 b = *reinterpret_cast<const float2 *>(pl + o + W);
                v[c] = make_float4((a.x - sub) * inv_div, (a.y - sub) * inv_div, (b.x - sub) * inv_div, (b.y - sub) * inv_div);
                am = max(max(am, abs_bits(v[c].x)), max(abs_bits(v[c].y), max(abs_bits(v[c].z), abs_bits(v[c].w))));
            }
        }
        float4 *dst = reinterpret_cast<float4 *>(y + (size_t)n * y_stride) + ((size_t)h2 * W2 + w2) * Q;
#pragma unroll
        for (int c = 0; c < QMAX; ++c)
            if (c < Q) dst[c] = v[c];
    }
    amax_commit(am, amax_y);
}

// filter regrouping of the space-to-depth stems and its adjoint:
//   w [K,C,7,7] (KCRS) <-> ws [K,4,4,Cp] (KRSC', channel c*4 + dy*2 + dx), tap u = 2r' + dy - 1, v = 2s' + dx - 1
// (wcl: the 7x7 filter / its gradient lives in channels-last memory [K,7,7,C] — what torch keeps behind a channels_last parameter —
//  instead of [K,C,7,7]: read / written in place, no layout copy on either side)
template <bool ADJOINT>
__global__ __launch_bounds__(256) void stem_regroup_kernel(const float *__restrict__ src, float *__restrict__ dst, int K, int C, int Cp, int wcl) {
    const int total = K * 16 * Cp;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ch = i % Cp, rs = (i / Cp) % 16, k = i / (16 * Cp);
        const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1, rp = rs >> 2, sp = rs & 3;
        const int u = 2 * rp + dy - 1, v = 2 * sp + dx - 1;
        const bool ok = c < C && u >= 0 && u < 7 && v >= 0 && v < 7;
        const int wi = wcl ? ((k * 7 + u) * 7 + v) * C + c : ((k * C + c) * 7 + u) * 7 + v;
        if (!ADJOINT) dst[i] = ok ? src[wi] : 0.f;
        else if (ok) dst[wi] = src[i];                                  // every (k,c,u,v) has exactly one (r',dy,s',dx)
    }
}
}  // namespace

// x [N,H,W,C] (H, W even) -> y [N,H/2,W/2,Cp], Cp >= 4*C, channels 4*C..Cp-1 zero
extern "C" int sqd_space_to_depth2(const float *x, float *y, int N, int H, int W, int C, int Cp, void *stream) {
    SQD_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && Cp >= 4 * C,
                  "sqd_space_to_depth2: bad arguments (H=%d W=%d C=%d Cp=%d)", H, W, C, Cp);
    (void)hipGetLastError();
    SQD_CHECK_ARG(Cp % 4 == 0, "sqd_space_to_depth2: Cp=%d must be a multiple of 4", Cp);
    hipLaunchKernelGGL(s2d_kernel, dim3(grid_for((size_t)N * (H / 2) * (W / 2) * Cp / 4)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Cp);
    SQD_CHECK_LAUNCH("sqd_space_to_depth2");
    return SQD_OK;
}

// x0 [N,C0,H,W], x1 [N,C1,H,W] or NULL (planar, dense) -> y[n] = y + n * y_stride floats, each [H/2,W/2,Cp] as sqd_space_to_depth2 of the
// channel concatenation (x0, x1), every value (v - sub) / div: the frame staging of the two stems in one pass — the encoder's
// (x - 0.45) / 0.225 (reference networks/resnet_encoder.py:93) and the pose network's torch.cat of a frame pair
// (reference trainer.py:319-326); y_stride lets the pairs of one sample interleave along the batch
extern "C" int sqd_space_to_depth2_planar(const float *x0, const float *x1, float *y, int N, int H, int W, int C0, int C1, int Cp,
                                          int64_t y_stride, float sub, float div, void *stream) {
    return sqd_space_to_depth2_planar_amax(x0, x1, y, N, H, W, C0, C1, Cp, y_stride, sub, div, nullptr, stream);
}
// ... and amax_y (may be NULL; cleared by the caller — once for all the launches that fill one batch): the bit pattern of max |y| (sqd.h 10b)
extern "C" int sqd_space_to_depth2_planar_amax(const float *x0, const float *x1, float *y, int N, int H, int W, int C0, int C1, int Cp,
                                               int64_t y_stride, float sub, float div, float *amax_y, void *stream) {
    SQD_CHECK_ARG(x0 && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C0 > 0 && C1 >= 0 && (x1 || C1 == 0) &&
                      Cp >= 4 * (C0 + C1) && Cp % 4 == 0 && div != 0.f && y_stride >= (int64_t)(H / 2) * (W / 2) * Cp && y_stride % 4 == 0,
                  "sqd_space_to_depth2_planar: bad arguments (H=%d W=%d C0=%d C1=%d Cp=%d)", H, W, C0, C1, Cp);
    SQD_CHECK_ARG(((uintptr_t)x0 & 7) == 0 && ((uintptr_t)x1 & 7) == 0 && ((uintptr_t)y & 15) == 0, "sqd_space_to_depth2_planar: alignment");
    (void)hipGetLastError();
    SQD_CHECK_ARG(Cp <= 64, "sqd_space_to_depth2_planar: Cp=%d (at most 64: frames of up to 16 channels)", Cp);
    const dim3 grid(grid_for((size_t)N * (H / 2) * (W / 2)));
    if (Cp <= 16)
        hipLaunchKernelGGL(s2d_planar_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x0, x1, y, N, H, W, C0, C1, Cp, (long long)y_stride, sub, 1.0f / div, (unsigned *)amax_y);
    else if (Cp <= 32)
        hipLaunchKernelGGL(s2d_planar_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x0, x1, y, N, H, W, C0, C1, Cp, (long long)y_stride, sub, 1.0f / div, (unsigned *)amax_y);
    else
        hipLaunchKernelGGL(s2d_planar_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, x0, x1, y, N, H, W, C0, C1, Cp, (long long)y_stride, sub, 1.0f / div, (unsigned *)amax_y);
    SQD_CHECK_LAUNCH("sqd_space_to_depth2_planar");
    return SQD_OK;
}

// w [K,C,7,7] contiguous -> ws [K,4,4,Cp] (adjoint = 0) or g_ws [K,4,4,Cp] -> g_w [K,C,7,7] fully overwritten (adjoint = 1)
extern "C" int sqd_stem_regroup_ex(const float *src, float *dst, int K, int C, int Cp, int adjoint, int w_channels_last, void *stream) {
    SQD_CHECK_ARG(src && dst && K > 0 && C > 0 && Cp >= 4 * C, "sqd_stem_regroup: bad arguments");
    (void)hipGetLastError();
    const int nb = (K * 16 * Cp + 255) / 256;
    if (adjoint) hipLaunchKernelGGL((stem_regroup_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, src, dst, K, C, Cp, w_channels_last);
    else hipLaunchKernelGGL((stem_regroup_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, src, dst, K, C, Cp, w_channels_last);
    SQD_CHECK_LAUNCH("sqd_stem_regroup");
    return SQD_OK;
}
extern "C" int sqd_stem_regroup(const float *src, float *dst, int K, int C, int Cp, int adjoint, void *stream) {
    return sqd_stem_regroup_ex(src, dst, K, C, Cp, adjoint, 0, stream);
}
