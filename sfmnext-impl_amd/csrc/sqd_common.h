// sqd_common.h — shared helpers for the gfx950 kernels of libsqd.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>
#include <utility>

#include "../../include/sqd.h"

namespace sqd {

void set_error(const char *fmt, ...);

#define SQD_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            sqd::set_error(__VA_ARGS__);    \
            return SQD_EINVAL;              \
        }                                   \
    } while (0)

#define SQD_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            sqd::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return SQD_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

constexpr int WAVE = 64;

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(<N-1>) — register arrays indexed by
// the loop counter stay in VGPRs (a runtime index would send them to scratch).
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- wave64 cross-lane primitives (DPP, no LDS) -------------------------------------------------
// lane i receives lane i-1 (lane 0 receives 0)
__device__ __forceinline__ float wave_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
// lane i receives lane i+1 (lane 63 receives 0)
__device__ __forceinline__ float wave_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
// centred 7-tap box sum across lanes: out[i] = v[i-3] + ... + v[i+3] (zero beyond the wave edges).
// Six v_add_f32_dpp, no LDS traffic: the SSIM window reduction along x.
__device__ __forceinline__ float box7(float v) {
    float r = v + wave_shr1(v);          // v[i] + v[i-1]
    r = v + wave_shr1(r);                // .. + v[i-2]
    r = v + wave_shr1(r);                // .. + v[i-3]
    float u = v + wave_shl1(v);          // v[i] + v[i+1]
    u = v + wave_shl1(u);                // .. + v[i+2]
    return r + wave_shl1(u);             // + v[i+1] + v[i+2] + v[i+3]
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// total of v over the wavefront, delivered in lane 63: quad / half-row / row butterflies, then the two row broadcasts (six DPP adds)
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define SQD_DPP_ADD(ctrl, rmask, bc) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, bc))
    SQD_DPP_ADD(0xB1, 0xf, true);      // quad_perm [1,0,3,2]
    SQD_DPP_ADD(0x4E, 0xf, true);      // quad_perm [2,3,0,1]
    SQD_DPP_ADD(0x141, 0xf, true);     // row_half_mirror
    SQD_DPP_ADD(0x140, 0xf, true);     // row_mirror
    SQD_DPP_ADD(0x142, 0xa, false);    // row_bcast:15 into rows 1, 3
    SQD_DPP_ADD(0x143, 0xc, false);    // row_bcast:31 into rows 2, 3
#undef SQD_DPP_ADD
    return v;
}

// ---- max |x| bookkeeping for the two-term fp16 convolution operands (amax.hip, conv.hip "f16x2") ----------------------
// bit pattern of |x|: unsigned order == order of the magnitudes (NaN patterns sort above everything: a NaN poisons the scale like it
// would poison an fp32 product)
__device__ __forceinline__ unsigned abs_bits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
// A max |x| RECORD is SQD_AMAX_WAYS words, one per 64-byte line (SQD_AMAX_RECORD_FLOATS floats = 4 KB): device-scope atomics serialise per
// cache line at ~8 ns each (tools/ubench_atomic_max.hip: 4096 workgroups ending with one atomicMax on one word — or on 16 words of one
// line — turn a 23 us element-wise pass into 55 us; on 16 words in 16 lines into 23.5 us, on 64 into 23.1; a pre-check load costs more than
// it saves; in the training step 16 ways still cost the BatchNorm passes 2 us each, profiles/r05f).  A workgroup combines its waves
// through LDS and issues ONE atomic to way (blockIdx mod 64); the reader takes the max over the ways (amax_record_bits: one vector load).  EVERY thread of the workgroup must call amax_commit (it contains a barrier); amax == NULL:
// nothing recorded (uniform: no barrier either).  The record must have been cleared on the stream before the launch.  `which` (0 / 1): a
// kernel that records two maxima back to back gives them different staging rows (no barrier between the first one's read and the second
// one's write).
constexpr int AMAX_WAYS = SQD_AMAX_WAYS, AMAX_WAY_STRIDE = SQD_AMAX_RECORD_FLOATS / SQD_AMAX_WAYS;      // words
__device__ __forceinline__ void amax_commit(unsigned m, unsigned *__restrict__ amax, int which = 0) {
    if (!amax) return;
    __shared__ unsigned amax_stage[2][16];
    unsigned *amax_wave_max = amax_stage[which];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    const int nw = (int)((blockDim.x * blockDim.y * blockDim.z + 63u) >> 6);
    const int tid = (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z));
    if (nw > 1) {
        if ((tid & 63) == 0) amax_wave_max[tid >> 6] = m;
        __syncthreads();
        if (tid == 0)
            for (int w = 1; w < nw; ++w) m = max(m, amax_wave_max[w]);
    }
    if (tid == 0 && m != 0u) atomicMax(amax + ((blockIdx.x + 5u * blockIdx.y + 3u * blockIdx.z) & (AMAX_WAYS - 1)) * AMAX_WAY_STRIDE, m);
}
// the maximum a record holds: ONE vector load (lane e reads way e; 64 scalar loads from 64 cache lines per operand cost a two-term
// kernel ~10 us at its start — round 5's first in-step measurement, where those plans lost to the three-term ones they beat by 20 % in
// isolation) and a wave-wide maximum; the result is wave-uniform (SGPR)
__device__ __forceinline__ unsigned amax_record_bits(const float *__restrict__ rec) {
    static_assert(AMAX_WAYS == 64, "one way per lane");
    unsigned m = __float_as_uint(rec[(__lane_id() & (AMAX_WAYS - 1)) * AMAX_WAY_STRIDE]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}

// reflection index of ReflectionPad2d (pad < n), clamped for lanes far outside the image
__device__ __forceinline__ int reflect_idx(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * (n - 1) - p : p;
    return min(max(p, 0), n - 1);
}

// out[i] = sum over splits of part[s][i] for the 16 float4 columns of block `block` (256 threads): its 16 thread groups each add the
// splits s = g, g + 16, ... in order, then a fixed-order tree over the groups — deterministic for a given split count.  Shared by
// conv.hip's split_reduce_kernel and by the kernels that carry a pending reduction along (bn_finalize_bwd_kernel).
__device__ __forceinline__ void split_reduce_block(const float *__restrict__ part, float *__restrict__ out, size_t n, int splits, int block,
                                                   float4 (*red)[16]) {
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t i = (size_t)block * 16 + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n / 4)
        for (int s = grp; s < splits; s += 16) {
            const float4 b = reinterpret_cast<const float4 *>(part + (size_t)s * n)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    red[grp][col] = a;
    __syncthreads();
    for (int w = 8; w >= 1; w >>= 1) {
        if (grp < w) {
            const float4 b = red[grp + w][col];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            red[grp][col] = a;
        }
        __syncthreads();
    }
    if (grp == 0 && i < n / 4) reinterpret_cast<float4 *>(out)[i] = a;
}
}  // namespace sqd
