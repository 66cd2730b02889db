// upcat.hip — bilinear resize (align_corners=True) of x to the skip tensor's size, fused with the channel
// concat [up(x), skip], channels-last activations; and its adjoint.
// replaces: F.interpolate(x, size=skip.shape[2:], mode='bilinear', align_corners=True) + torch.cat
//           in UpSampleBN.forward (reference networks/resnet_encoder.py:114-117).
// Roofline: HBM — forward writes (Cx+Cs)*4 B and reads ~Cx*4/s^2 + Cs*4 B per output pixel; ATen's
// upsample_bilinear2d took 417 us per call on this shape class, i.e. ~5 % of the whole train step.
#include "sqd_common.h"

namespace {
using namespace sqd;

struct Tap {
    int i0, i1;
    float l0, l1;
};
// ATen area_pixel_compute_source_index(align_corners=True): src = dst * (in-1)/(out-1)
__device__ __forceinline__ Tap tap_ac(int dst, float scale, int in_size) {
    const float f = scale * (float)dst;
    Tap t;
    t.i0 = (int)f;
    t.i0 = t.i0 > in_size - 1 ? in_size - 1 : t.i0;
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = f - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

__global__ __launch_bounds__(256) void upcat_fwd_kernel(const float *__restrict__ x, const float *__restrict__ skip,
                                                        float *__restrict__ out, int N, int Hi, int Wi, int Cx, int Ho, int Wo,
                                                        int Cs, float sy, float sx, unsigned *__restrict__ amax_out) {
    unsigned am = 0u;                                       // max |out| over this thread's elements (amax_out == NULL: not recorded)
    // 32-bit index arithmetic (the host checks that every tensor has fewer than 2^31 float4 groups): the six 64-bit
    // divisions per element of the size_t version were most of this kernel's instructions
    const unsigned Ct = Cx + Cs, Vt = Ct / 4, Vx = Cx / 4;
    const unsigned total = (unsigned)N * Ho * Wo * Vt;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cv = i % Vt;
        const unsigned pix = i / Vt;
        if (cv >= Vx) {
            const float4 v = reinterpret_cast<const float4 *>(skip)[pix * (Cs / 4) + (cv - Vx)];
            reinterpret_cast<float4 *>(out)[i] = v;
            am = max(max(am, abs_bits(v.x)), max(abs_bits(v.y), max(abs_bits(v.z), abs_bits(v.w))));
            continue;
        }
        const int xo = (int)(pix % (unsigned)Wo);
        const unsigned t2 = pix / (unsigned)Wo;
        const int yo = (int)(t2 % (unsigned)Ho), n = (int)(t2 / (unsigned)Ho);
        const Tap ty = tap_ac(yo, sy, Hi), tx = tap_ac(xo, sx, Wi);
        const float4 *xb = reinterpret_cast<const float4 *>(x) + (size_t)n * Hi * Wi * Vx + cv;
        const float4 v00 = xb[((size_t)ty.i0 * Wi + tx.i0) * Vx], v01 = xb[((size_t)ty.i0 * Wi + tx.i1) * Vx];
        const float4 v10 = xb[((size_t)ty.i1 * Wi + tx.i0) * Vx], v11 = xb[((size_t)ty.i1 * Wi + tx.i1) * Vx];
        float4 o;
        o.x = ty.l0 * (tx.l0 * v00.x + tx.l1 * v01.x) + ty.l1 * (tx.l0 * v10.x + tx.l1 * v11.x);
        o.y = ty.l0 * (tx.l0 * v00.y + tx.l1 * v01.y) + ty.l1 * (tx.l0 * v10.y + tx.l1 * v11.y);
        o.z = ty.l0 * (tx.l0 * v00.z + tx.l1 * v01.z) + ty.l1 * (tx.l0 * v10.z + tx.l1 * v11.z);
        o.w = ty.l0 * (tx.l0 * v00.w + tx.l1 * v01.w) + ty.l1 * (tx.l0 * v10.w + tx.l1 * v11.w);
        reinterpret_cast<float4 *>(out)[i] = o;
        am = max(max(am, abs_bits(o.x)), max(abs_bits(o.y), max(abs_bits(o.z), abs_bits(o.w))));
    }
    amax_commit(am, amax_out);
}

// adjoint: first N*Hi*Wi*Cx/4 work items gather g_x, the remaining N*Ho*Wo*Cs/4 copy g_skip
// xb != NULL (launched with 256 % (Cx / 4) == 0: a thread keeps its channel group over the g_x items): g_x is the whole gradient of the
// BatchNorm + activation that produced x (a decoder stage's output) — that node's two backward sums (sum dz, sum dz * xhat), dz = g_x * act'(.),
// are taken here, one partial row [Cx][2] per workgroup (pool.hip's maxpool_bwd_kernel does the same for the stem): xb [N,Hi,Wi,Cx] the
// BatchNorm's input, maskb its sign bits (NULL: no activation), slope the activation's derivative on the negative side
__global__ __launch_bounds__(256) void upcat_bwd_kernel(const float *__restrict__ g_out, float *__restrict__ g_x,
                                                        float *__restrict__ g_skip, int N, int Hi, int Wi, int Cx, int Ho,
                                                        int Wo, int Cs, float sy, float sx, float isy, float isx, const float *__restrict__ xb = nullptr,
                                                        const unsigned char *__restrict__ maskb = nullptr, const float *__restrict__ meanb = nullptr,
                                                        const float *__restrict__ rstdb = nullptr, float slope = 0.f, float *__restrict__ partb = nullptr,
                                                        unsigned *__restrict__ amax_gx = nullptr) {
    // amax_gx (may be NULL; cleared by the caller): the bit pattern of max |g_x| — the operand scale of a convolution backward that reads g_x on
    // two-term fp16 operands (the decoder's first 1x1 layer has no BatchNorm behind it: g_x is its output gradient)
    __shared__ float4 red[2][256];
    unsigned am = 0u;
    const unsigned Ct = Cx + Cs, Vt = Ct / 4, Vx = Cx / 4, Vs = Cs / 4;
    const unsigned nx = (unsigned)N * Hi * Wi * Vx, ns = (unsigned)N * Ho * Wo * Vs;
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), qb = sb, mub = sb, rsb = sb;
    if (xb) {
        mub = reinterpret_cast<const float4 *>(meanb)[threadIdx.x % Vx];
        rsb = reinterpret_cast<const float4 *>(rstdb)[threadIdx.x % Vx];
    }
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nx + ns; i += gridDim.x * 256u) {
        if (i >= nx) {
            const unsigned k = i - nx;
            const unsigned pix = k / Vs;
            reinterpret_cast<float4 *>(g_skip)[k] = reinterpret_cast<const float4 *>(g_out)[(size_t)pix * Vt + Vx + (k % Vs)];
            continue;
        }
        const unsigned cv = i % Vx;
        const unsigned pix = i / Vx;
        const int xi = (int)(pix % (unsigned)Wi);
        const unsigned t2 = pix / (unsigned)Wi;
        const int yi = (int)(t2 % (unsigned)Hi), n = (int)(t2 / (unsigned)Hi);
        // destination rows / columns whose taps can touch (yi, xi) — conservative, membership re-tested
        const int ylo = max(0, (int)floorf(((float)yi - 1.f) * isy) - 1), yhi = min(Ho - 1, (int)ceilf(((float)yi + 1.f) * isy) + 1);
        const int xlo = max(0, (int)floorf(((float)xi - 1.f) * isx) - 1), xhi = min(Wo - 1, (int)ceilf(((float)xi + 1.f) * isx) + 1);
        const float4 *gb = reinterpret_cast<const float4 *>(g_out) + (size_t)n * Ho * Wo * Vt + cv;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int y = ylo; y <= yhi; ++y) {
            const Tap ty = tap_ac(y, sy, Hi);
            const float wy = (ty.i0 == yi ? ty.l0 : 0.f) + (ty.i1 == yi ? ty.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int xq = xlo; xq <= xhi; ++xq) {
                const Tap tx = tap_ac(xq, sx, Wi);
                const float w = wy * ((tx.i0 == xi ? tx.l0 : 0.f) + (tx.i1 == xi ? tx.l1 : 0.f));
                if (w == 0.f) continue;
                const float4 g = gb[((size_t)y * Wo + xq) * Vt];
                acc.x = fmaf(w, g.x, acc.x); acc.y = fmaf(w, g.y, acc.y); acc.z = fmaf(w, g.z, acc.z); acc.w = fmaf(w, g.w, acc.w);
            }
        }
        reinterpret_cast<float4 *>(g_x)[i] = acc;
        am = max(max(am, abs_bits(acc.x)), max(abs_bits(acc.y), max(abs_bits(acc.z), abs_bits(acc.w))));
        if (xb) {
            const unsigned bits = maskb ? maskb[i] : 0xfu;
            const float4 xv = reinterpret_cast<const float4 *>(xb)[i];
            const float d0 = acc.x * ((bits & 1u) ? 1.f : slope), d1 = acc.y * ((bits & 2u) ? 1.f : slope), d2 = acc.z * ((bits & 4u) ? 1.f : slope),
                        d3 = acc.w * ((bits & 8u) ? 1.f : slope);
            sb.x += d0; sb.y += d1; sb.z += d2; sb.w += d3;
            qb.x = fmaf(d0, (xv.x - mub.x) * rsb.x, qb.x); qb.y = fmaf(d1, (xv.y - mub.y) * rsb.y, qb.y);
            qb.z = fmaf(d2, (xv.z - mub.z) * rsb.z, qb.z); qb.w = fmaf(d3, (xv.w - mub.w) * rsb.w, qb.w);
        }
    }
    if (xb) {      // (uniform) the workgroup's partial row: the threads of a channel group added in a fixed order
        red[0][threadIdx.x] = sb;
        red[1][threadIdx.x] = qb;
        __syncthreads();
        if (threadIdx.x < Vx) {
            for (unsigned k = threadIdx.x + Vx; k < 256u; k += Vx) {
                const float4 u = red[0][k], v = red[1][k];
                sb.x += u.x; sb.y += u.y; sb.z += u.z; sb.w += u.w;
                qb.x += v.x; qb.y += v.y; qb.z += v.z; qb.w += v.w;
            }
            float *o = partb + ((size_t)blockIdx.x * Cx + threadIdx.x * 4) * 2;
            reinterpret_cast<float4 *>(o)[0] = make_float4(sb.x, qb.x, sb.y, qb.y);
            reinterpret_cast<float4 *>(o)[1] = make_float4(sb.z, qb.z, sb.w, qb.w);
        }
    }
    amax_commit(am, amax_gx);
}

int grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
}  // namespace

extern "C" int sqd_upcat_fwd(const float *x, const float *skip, float *out, int N, int Hi, int Wi, int Cx, int Ho, int Wo,
                             int Cs, void *stream) {
    return sqd_upcat_fwd_amax(x, skip, out, N, Hi, Wi, Cx, Ho, Wo, Cs, nullptr, stream);
}
// ... and amax_out (may be NULL; cleared by the caller): the bit pattern of max |out|, for the convolution that reads `out` on two-term
// fp16 operands (sqd.h section 10b)
extern "C" int sqd_upcat_fwd_amax(const float *x, const float *skip, float *out, int N, int Hi, int Wi, int Cx, int Ho, int Wo,
                                  int Cs, float *amax_out, void *stream) {
    SQD_CHECK_ARG(x && skip && out, "sqd_upcat_fwd: null pointer");
    SQD_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Cx % 4 == 0 && Cs % 4 == 0 && Cx > 0 && Cs > 0,
                  "sqd_upcat_fwd: bad shape (channels must be multiples of 4)");
    SQD_CHECK_ARG((long long)N * Ho * Wo * (Cx + Cs) / 4 < (1ll << 31), "sqd_upcat_fwd: output of 2^31 float4 groups or more");
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    (void)hipGetLastError();
    hipLaunchKernelGGL(upcat_fwd_kernel, dim3(grid_for((size_t)N * Ho * Wo * (Cx + Cs) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       skip, out, N, Hi, Wi, Cx, Ho, Wo, Cs, sy, sx, (unsigned *)amax_out);
    SQD_CHECK_LAUNCH("sqd_upcat_fwd");
    return SQD_OK;
}

extern "C" int sqd_upcat_bwd(const float *g_out, float *g_x, float *g_skip, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs,
                             void *stream) {
    return sqd_upcat_bwd_bn(g_out, g_x, g_skip, N, Hi, Wi, Cx, Ho, Wo, Cs, nullptr, nullptr, nullptr, nullptr, 0, nullptr, stream);
}
// the partial rows sqd_upcat_bwd_bn writes at this shape (0: not served — the channel groups of x must divide a workgroup's 256 threads)
extern "C" int sqd_upcat_bwd_bn_rows(int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs) {
    if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || Cx < 4 || Cx % 4 || Cs < 4 || Cs % 4 || 256 % (Cx / 4)) return 0;
    const int g = grid_for((size_t)N * Hi * Wi * Cx / 4 + (size_t)N * Ho * Wo * Cs / 4);
    return g < 2048 ? g : 2048;
}
// sqd_upcat_bwd that also takes the two BatchNorm-backward sums of the node that produced x = act(BatchNorm(xb)) (g_x is its whole incoming
// gradient): xb [N,Hi,Wi,Cx], maskb the sign bits its forward stored (NULL with act 0), meanb / rstdb [Cx], act 0 none / 1 ReLU / 2 LeakyReLU(0.01)
// -> partb [sqd_upcat_bwd_bn_rows(...)][Cx][2] for its sqd_bn_train_bwd_pre.  xb = NULL: sqd_upcat_bwd.
extern "C" int sqd_upcat_bwd_bn(const float *g_out, float *g_x, float *g_skip, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs, const float *xb,
                                const unsigned char *maskb, const float *meanb, const float *rstdb, int act, float *partb, void *stream) {
    return sqd_upcat_bwd_bn_amax(g_out, g_x, g_skip, N, Hi, Wi, Cx, Ho, Wo, Cs, xb, maskb, meanb, rstdb, act, partb, nullptr, stream);
}
// ... and amax_gx (may be NULL; cleared by the caller): the bit pattern of max |g_x| (sqd.h section 10b)
extern "C" int sqd_upcat_bwd_bn_amax(const float *g_out, float *g_x, float *g_skip, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs, const float *xb,
                                     const unsigned char *maskb, const float *meanb, const float *rstdb, int act, float *partb, float *amax_gx,
                                     void *stream) {
    SQD_CHECK_ARG(!xb || (meanb && rstdb && partb && (act == 0 || ((act == 1 || act == 2) && maskb)) && sqd_upcat_bwd_bn_rows(N, Hi, Wi, Cx, Ho, Wo, Cs) > 0),
                  "sqd_upcat_bwd_bn: the BatchNorm sums need meanb, rstdb, partb, the sign mask with ReLU / LeakyReLU, and Cx / 4 dividing 256");
    SQD_CHECK_ARG(g_out && g_x && g_skip, "sqd_upcat_bwd: null pointer");
    SQD_CHECK_ARG(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && Cx % 4 == 0 && Cs % 4 == 0 && Cx > 0 && Cs > 0,
                  "sqd_upcat_bwd: bad shape (channels must be multiples of 4)");
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const float isy = sy > 0.f ? 1.f / sy : (float)Ho, isx = sx > 0.f ? 1.f / sx : (float)Wo;
    const size_t total = (size_t)N * Hi * Wi * Cx / 4 + (size_t)N * Ho * Wo * Cs / 4;
    SQD_CHECK_ARG(total < (1ull << 31) && (long long)N * Ho * Wo * (Cx + Cs) / 4 < (1ll << 31), "sqd_upcat_bwd: tensors of 2^31 float4 groups or more");
    (void)hipGetLastError();
    hipLaunchKernelGGL(upcat_bwd_kernel, dim3(xb ? sqd_upcat_bwd_bn_rows(N, Hi, Wi, Cx, Ho, Wo, Cs) : grid_for(total)), dim3(256), 0, (hipStream_t)stream, g_out,
                       g_x, g_skip, N, Hi, Wi, Cx, Ho, Wo, Cs, sy, sx, isy, isx, xb, act ? maskb : nullptr, meanb, rstdb, act == 2 ? 0.01f : 0.f, partb,
                       (unsigned *)amax_gx);
    SQD_CHECK_LAUNCH("sqd_upcat_bwd");
    return SQD_OK;
}
