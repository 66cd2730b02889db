// amax.hip — max |x| of a tensor as a device scalar: the power-of-two operand scale of the two-term fp16 convolution
// arithmetic (conv.hip, "f16x2") is derived from it.  No reference counterpart: the reference multiplies in fp32 on cuDNN
// (networks/resnet_encoder.py:89-147); this is bookkeeping of the build's own operand format.
// A record (SQD_AMAX_RECORD_FLOATS floats, sqd_common.h) holds BIT PATTERNS of non-negative floats in 16 of its words, the tensor's max |x|
// being their maximum (an unsigned max of |x|'s bits is the max of |x|; a NaN anywhere yields a NaN pattern, which the consumer turns
// into NaN results like fp32 arithmetic would).  Producers that already touch every
// element (BatchNorm / activation / up-sampling kernels, convolution epilogues, the optimiser) record it on the way — the
// `amax` arguments of their entry points, amax_commit() of sqd_common.h; these kernels serve tensors that have no such producer.
// Roofline: HBM, 4 B read per element.
#include "sqd_common.h"

namespace {
using namespace sqd;

__global__ __launch_bounds__(256) void amax_kernel(const float *__restrict__ x, size_t n, unsigned *__restrict__ amax) {
    unsigned m = 0u;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        m = max(max(m, abs_bits(v.x)), max(abs_bits(v.y), max(abs_bits(v.z), abs_bits(v.w))));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) m = max(m, abs_bits(x[i]));
    amax_commit(m, amax);
}

struct TensorRec {                   // the optimiser's parameter table (adam.hip)
    float *p;
    float *m;
    float *v;
    long long n;
};
constexpr int CHUNK = 4096;          // sqd_adam_chunk_elems()

// one entry of amax[] per parameter tensor of the table
__global__ __launch_bounds__(256) void amax_multi_kernel(const TensorRec *__restrict__ recs, const int2 *__restrict__ chunks,
                                                         unsigned *__restrict__ amax) {
    const int2 ch = chunks[blockIdx.x];
    const TensorRec r = recs[ch.x];
    const long long base = (long long)ch.y * CHUNK;
    unsigned m = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + ((long long)k * 256 + threadIdx.x) * 4;
        if (i >= r.n) break;
        if (i + 3 < r.n && ((size_t)r.p & 15) == 0) {
            const float4 v = *reinterpret_cast<const float4 *>(r.p + i);
            m = max(max(m, abs_bits(v.x)), max(abs_bits(v.y), max(abs_bits(v.z), abs_bits(v.w))));
        } else {
            for (long long e = i; e < i + 4 && e < r.n; ++e) m = max(m, abs_bits(r.p[e]));
        }
    }
    amax_commit(m, amax + (size_t)ch.x * SQD_AMAX_RECORD_FLOATS);
}
}  // namespace

// record `amax` (SQD_AMAX_RECORD_FLOATS floats) = max |x[i]|, i < n (the record is cleared on the stream first: two graph nodes)
extern "C" int sqd_amax(const float *x, int64_t n, float *amax, void *stream) {
    SQD_CHECK_ARG(x && amax && n > 0 && ((uintptr_t)x & 15) == 0, "sqd_amax: bad arguments (x 16-byte aligned, n > 0)");
    (void)hipGetLastError();
    if (hipMemsetAsync(amax, 0, SQD_AMAX_RECORD_FLOATS * 4, (hipStream_t)stream) != hipSuccess) {
        sqd::set_error("sqd_amax: hipMemsetAsync failed");
        return SQD_ELAUNCH;
    }
    const size_t nb = ((size_t)n / 4 + 255) / 256;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)(nb < 1 ? 1 : nb > 2048 ? 2048 : nb)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, (unsigned *)amax);
    SQD_CHECK_LAUNCH("sqd_amax");
    return SQD_OK;
}

// record amax + t * SQD_AMAX_RECORD_FLOATS = max |p_t| for every parameter tensor t of an optimiser table (recs / chunks as for
// sqd_adam_step; the ntensors records are cleared first).  One launch for all filters of the networks.
extern "C" int sqd_amax_multi(const void *recs, const void *chunks, int nchunks, int ntensors, float *amax, void *stream) {
    SQD_CHECK_ARG(recs && chunks && nchunks > 0 && ntensors > 0 && amax, "sqd_amax_multi: bad arguments");
    (void)hipGetLastError();
    if (hipMemsetAsync(amax, 0, (size_t)ntensors * SQD_AMAX_RECORD_FLOATS * 4, (hipStream_t)stream) != hipSuccess) {
        sqd::set_error("sqd_amax_multi: hipMemsetAsync failed");
        return SQD_ELAUNCH;
    }
    hipLaunchKernelGGL(amax_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const TensorRec *)recs, (const int2 *)chunks, (unsigned *)amax);
    SQD_CHECK_LAUNCH("sqd_amax_multi");
    return SQD_OK;
}
