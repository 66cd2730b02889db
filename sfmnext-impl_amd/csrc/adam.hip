// adam.hip — one-launch multi-tensor Adam step (reference trainer.py:128-135,244: torch.optim.Adam,
// default betas/eps, no weight decay, no amsgrad).
// replaces: the ~30 multi_tensor_apply / foreach launches of torch.optim.Adam.step() per optimiser step.
// Roofline: HBM — 16 B read + 12 B written per parameter (p, g, m, v -> p, m, v); 33 M parameters at
// config B = 0.92 GB per step.
#include "sqd_common.h"

namespace {
using namespace sqd;
constexpr int CHUNK = 4096;          // elements per block: 256 threads x 4 float4

struct TensorRec {                   // one parameter tensor (device table, rebuilt only when shapes change)
    float *p;
    float *m;
    float *v;
    long long n;
};

// amax_recs != NULL: [ntensors] device pointers to max |x| records (sqd_common.h: amax_commit; cleared on the stream by the caller) or NULL per
// tensor — the kernel leaves max |p_new| of every listed tensor behind, so that the convolutions that read those filters on two-term fp16
// operands in the next step need no pass of their own over the weights (amax.hip: sqd_amax_multi, 131 MB per step at configs[1])
__global__ __launch_bounds__(256) void adam_kernel(const TensorRec *__restrict__ recs, const float *const *__restrict__ grads,
                                                   const int2 *__restrict__ chunks, float step_size, float omb1, float beta2,
                                                   float omb2, float eps, float inv_bc2_sqrt, const float *__restrict__ hyper,
                                                   unsigned *const *__restrict__ amax_recs = nullptr) {
    if (hyper) {                                     // step-dependent scalars from device memory (graph replay: the launch
        step_size = hyper[0];                        // arguments are frozen at capture time)
        inv_bc2_sqrt = hyper[1];
    }
    const int2 ch = chunks[blockIdx.x];              // (tensor index, chunk index inside the tensor)
    const TensorRec r = recs[ch.x];
    const float *__restrict__ g = grads[ch.x];
    const long long base = (long long)ch.y * CHUNK;
    unsigned am = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
        if (i >= r.n) break;
        if (i + 3 < r.n && ((((size_t)r.p | (size_t)g | (size_t)r.m | (size_t)r.v) & 15) == 0)) {
            float4 pv = *reinterpret_cast<float4 *>(r.p + i), mv = *reinterpret_cast<float4 *>(r.m + i);
            float4 vv = *reinterpret_cast<float4 *>(r.v + i);
            const float4 gv = *reinterpret_cast<const float4 *>(g + i);
            float *pp = &pv.x, *mm = &mv.x, *vq = &vv.x;
            const float *gg = &gv.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mm[e] = fmaf(gg[e] - mm[e], omb1, mm[e]);                     // exp_avg.lerp_(grad, 1 - beta1)
                vq[e] = fmaf(omb2 * gg[e], gg[e], vq[e] * beta2);             // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
                const float denom = sqrtf(vq[e]) * inv_bc2_sqrt + eps;
                pp[e] -= step_size * (mm[e] / denom);                         // param.addcdiv_(exp_avg, denom, -step_size)
                am = max(am, abs_bits(pp[e]));
            }
            *reinterpret_cast<float4 *>(r.p + i) = pv;
            *reinterpret_cast<float4 *>(r.m + i) = mv;
            *reinterpret_cast<float4 *>(r.v + i) = vv;
        } else {
            for (long long e = i; e < i + 4 && e < r.n; ++e) {
                const float ge = g[e];
                float me = r.m[e], ve = r.v[e];
                me = fmaf(ge - me, omb1, me);
                ve = fmaf(omb2 * ge, ge, ve * beta2);
                r.m[e] = me;
                r.v[e] = ve;
                const float pn = r.p[e] - step_size * (me / (sqrtf(ve) * inv_bc2_sqrt + eps));
                r.p[e] = pn;
                am = max(am, abs_bits(pn));
            }
        }
    }
    if (amax_recs) amax_commit(am, amax_recs[ch.x]);      // (workgroup-uniform: one tensor per workgroup; NULL entry: nothing recorded)
}

// AdamW (torch.optim.AdamW: param.mul_(1 - lr * weight_decay) first, then the Adam update) with the gradient optionally scaled by a
// device scalar — the clip coefficient of clip_grad_norm_, so that clipping costs no pass over the gradients
__global__ __launch_bounds__(256) void adamw_kernel(const TensorRec *__restrict__ recs, const float *const *__restrict__ grads,
                                                    const int2 *__restrict__ chunks, float step_size, float omb1, float beta2,
                                                    float omb2, float eps, float inv_bc2_sqrt, float keep, const float *__restrict__ gscale) {
    const float gs = gscale ? gscale[0] : 1.0f;
    const int2 ch = chunks[blockIdx.x];
    const TensorRec r = recs[ch.x];
    const float *__restrict__ g = grads[ch.x];
    const long long base = (long long)ch.y * CHUNK;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
        if (i >= r.n) break;
        if (i + 3 < r.n && ((((size_t)r.p | (size_t)g | (size_t)r.m | (size_t)r.v) & 15) == 0)) {
            float4 pv = *reinterpret_cast<float4 *>(r.p + i), mv = *reinterpret_cast<float4 *>(r.m + i);
            float4 vv = *reinterpret_cast<float4 *>(r.v + i);
            const float4 gv = *reinterpret_cast<const float4 *>(g + i);
            float *pp = &pv.x, *mm = &mv.x, *vq = &vv.x;
            const float *gg = &gv.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = gg[e] * gs;
                mm[e] = fmaf(ge - mm[e], omb1, mm[e]);
                vq[e] = fmaf(omb2 * ge, ge, vq[e] * beta2);
                pp[e] = pp[e] * keep - step_size * (mm[e] / (sqrtf(vq[e]) * inv_bc2_sqrt + eps));
            }
            *reinterpret_cast<float4 *>(r.p + i) = pv;
            *reinterpret_cast<float4 *>(r.m + i) = mv;
            *reinterpret_cast<float4 *>(r.v + i) = vv;
        } else {
            for (long long e = i; e < i + 4 && e < r.n; ++e) {
                const float ge = g[e] * gs;
                float me = r.m[e], ve = r.v[e];
                me = fmaf(ge - me, omb1, me);
                ve = fmaf(omb2 * ge, ge, ve * beta2);
                r.m[e] = me;
                r.v[e] = ve;
                r.p[e] = r.p[e] * keep - step_size * (me / (sqrtf(ve) * inv_bc2_sqrt + eps));
            }
        }
    }
}

// per-chunk sums of squares of the gradients (the chunks of the Adam table), then their fixed-order total
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const TensorRec *__restrict__ recs, const float *const *__restrict__ grads,
                                                         const int2 *__restrict__ chunks, float *__restrict__ part) {
    __shared__ float red[4];
    const int2 ch = chunks[blockIdx.x];
    const long long n = recs[ch.x].n;
    const float *__restrict__ g = grads[ch.x];
    const long long base = (long long)ch.y * CHUNK;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i0 = base + ((long long)k * 256 + threadIdx.x) * 4;
        if (i0 >= n) break;
        if (i0 + 3 < n && ((size_t)g & 15) == 0) {
            const float4 gv = *reinterpret_cast<const float4 *>(g + i0);
            s = fmaf(gv.x, gv.x, s); s = fmaf(gv.y, gv.y, s); s = fmaf(gv.z, gv.z, s); s = fmaf(gv.w, gv.w, s);
        } else {
            for (long long e = i0; e < i0 + 4 && e < n; ++e) s = fmaf(g[e], g[e], s);
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float *__restrict__ part, int n, float max_norm, float *__restrict__ out) {
    __shared__ double red[1024];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;       // four independent chains per thread; fixed order
    int i = threadIdx.x;
    for (; i + 3072 < n; i += 4096) {
        s0 += (double)part[i]; s1 += (double)part[i + 1024]; s2 += (double)part[i + 2048]; s3 += (double)part[i + 3072];
    }
    for (; i < n; i += 1024) s0 += (double)part[i];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        const float coef = max_norm / (norm + 1e-6f);          // torch.nn.utils.clip_grad_norm_
        out[0] = coef < 1.0f ? coef : 1.0f;
        out[1] = norm;
    }
}
}  // namespace

extern "C" int sqd_adam_chunk_elems(void) { return CHUNK; }

// torch.optim.AdamW step on the tables of sqd_adam_step; gscale: device float multiplying every gradient (NULL: 1)
extern "C" int sqd_adamw_step(const void *recs, const void *grads, const void *chunks, int nchunks, double lr, double beta1,
                              double beta2, double eps, double weight_decay, int step, const float *gscale, void *stream) {
    SQD_CHECK_ARG(recs && grads && chunks && nchunks > 0 && step >= 1 && weight_decay >= 0, "sqd_adamw_step: bad arguments");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    (void)hipGetLastError();
    hipLaunchKernelGGL(adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const TensorRec *)recs,
                       (const float *const *)grads, (const int2 *)chunks, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)(1.0 / sqrt(bc2)), (float)(1.0 - lr * weight_decay), gscale);
    SQD_CHECK_LAUNCH("sqd_adamw_step");
    return SQD_OK;
}
// part [nchunks] floats: per-chunk sums of squares of the gradients of one Adam table (several tables may fill one array)
extern "C" int sqd_grad_sumsq(const void *recs, const void *grads, const void *chunks, int nchunks, float *part, void *stream) {
    SQD_CHECK_ARG(recs && grads && chunks && nchunks > 0 && part, "sqd_grad_sumsq: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const TensorRec *)recs,
                       (const float *const *)grads, (const int2 *)chunks, part);
    SQD_CHECK_LAUNCH("sqd_grad_sumsq");
    return SQD_OK;
}
// part [n] -> coef_norm [2] device floats: min(1, max_norm / (sqrt(sum part) + 1e-6)) and the norm (torch.nn.utils.clip_grad_norm_
// over all tensors that contributed); fixed-order sum
extern "C" int sqd_clip_coef(const float *part, int n, double max_norm, float *coef_norm, void *stream) {
    SQD_CHECK_ARG(part && n > 0 && coef_norm && max_norm > 0, "sqd_clip_coef: bad arguments");
    (void)hipGetLastError();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, n, (float)max_norm, coef_norm);
    SQD_CHECK_LAUNCH("sqd_clip_coef");
    return SQD_OK;
}

// recs [ntensors] of {p, m, v, n} (device), grads [ntensors] device pointers (device array), chunks [nchunks] int2 (device)
extern "C" int sqd_adam_step(const void *recs, const void *grads, const void *chunks, int nchunks, double lr, double beta1,
                             double beta2, double eps, int step, void *stream) {
    return sqd_adam_step_amax(recs, grads, chunks, nchunks, lr, beta1, beta2, eps, step, nullptr, stream);
}
// ... and amax_recs (may be NULL): [ntensors] device pointers (device array) to SQD_AMAX_RECORD_FLOATS-float records, or NULL per tensor; the
// records must have been cleared on the stream: the step leaves the bit pattern of max |p| of each listed tensor in its record
extern "C" int sqd_adam_step_amax(const void *recs, const void *grads, const void *chunks, int nchunks, double lr, double beta1,
                                  double beta2, double eps, int step, const void *amax_recs, void *stream) {
    SQD_CHECK_ARG(recs && grads && chunks && nchunks > 0 && step >= 1, "sqd_adam_step: bad arguments");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const float step_size = (float)(lr / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    // hyper-parameters arrive as doubles: torch.optim does this scalar arithmetic in Python floats
    // (1.f - 0.999f would be 1.3e-5 off the 1 - beta2 it uses)
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    (void)hipGetLastError();
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const TensorRec *)recs,
                       (const float *const *)grads, (const int2 *)chunks, step_size, omb1, (float)beta2, omb2, (float)eps, inv_bc2_sqrt,
                       (const float *)nullptr, (unsigned *const *)amax_recs);
    SQD_CHECK_LAUNCH("sqd_adam_step");
    return SQD_OK;
}

// host side of the step-dependent scalars: hyper[0] = lr / (1 - beta1^step), hyper[1] = 1 / sqrt(1 - beta2^step)
extern "C" int sqd_adam_hyper(double lr, double beta1, double beta2, int step, float *hyper_host) {
    SQD_CHECK_ARG(hyper_host && step >= 1, "sqd_adam_hyper: bad arguments");
    hyper_host[0] = (float)(lr / (1.0 - pow(beta1, step)));
    hyper_host[1] = (float)(1.0 / sqrt(1.0 - pow(beta2, step)));
    return SQD_OK;
}

// the same step with hyper (2 device floats, see sqd_adam_hyper) read by the kernel: capturable in a hipGraph whose
// replays only need the two floats refreshed
extern "C" int sqd_adam_step_dev(const void *recs, const void *grads, const void *chunks, int nchunks, const float *hyper_dev,
                                 double beta1, double beta2, double eps, void *stream) {
    return sqd_adam_step_dev_amax(recs, grads, chunks, nchunks, hyper_dev, beta1, beta2, eps, nullptr, stream);
}
extern "C" int sqd_adam_step_dev_amax(const void *recs, const void *grads, const void *chunks, int nchunks, const float *hyper_dev,
                                      double beta1, double beta2, double eps, const void *amax_recs, void *stream) {
    SQD_CHECK_ARG(recs && grads && chunks && nchunks > 0 && hyper_dev, "sqd_adam_step_dev: bad arguments");
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    (void)hipGetLastError();
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const TensorRec *)recs,
                       (const float *const *)grads, (const int2 *)chunks, 0.f, omb1, (float)beta2, omb2, (float)eps, 0.f, hyper_dev,
                       (unsigned *const *)amax_recs);
    SQD_CHECK_LAUNCH("sqd_adam_step_dev");
    return SQD_OK;
}
