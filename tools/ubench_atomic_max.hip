// ubench_atomic_max.hip — what does "every workgroup ends with one atomicMax" cost a short kernel on gfx950, and how far apart must the
// target words be for the cost to vanish? (dev tool, round 5: layout of the max |x| records of amax_commit, sqd_common.h)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_atomic_max.hip -o tools/bin/ubench_atomic_max && tools/bin/ubench_atomic_max
// Kernel: NB workgroups of 256 threads stream 16 KB each (a ~10 us element-wise pass), then thread 0 issues ONE atomicMax to
// word (blockIdx % NA) * STRIDE of a cleared buffer — agent scope (what a cross-XCD maximum needs) or workgroup scope (L2 of the XCD:
// NOT coherent across XCDs, shown for the price only).  Reported: kernel time per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: no atomic, 1: agent scope, 2: workgroup scope, 3: agent scope behind an agent-scope pre-check load
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ src, float4 *__restrict__ dst, unsigned *__restrict__ rec, int na, int stride_words) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned m = 0;
    for (int i = 0; i < 4; ++i) {
        const float4 v = src[((size_t)blockIdx.x * 4 + i) * 256 + threadIdx.x];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        m = max(m, __float_as_uint(v.x) & 0x7fffffffu);
        dst[((size_t)blockIdx.x * 4 + i) * 256 + threadIdx.x] = acc;
    }
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    __shared__ unsigned wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        unsigned *p = rec + (size_t)(blockIdx.x % na) * stride_words;
        if (MODE == 1) __hip_atomic_fetch_max(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) __hip_atomic_fetch_max(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3 && m > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __hip_atomic_fetch_max(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const int NB = 4096;
    float4 *src, *dst;
    unsigned *rec;
    CHECK(hipMalloc(&src, (size_t)NB * 4 * 256 * 16));
    CHECK(hipMalloc(&dst, (size_t)NB * 4 * 256 * 16));
    CHECK(hipMalloc(&rec, 1 << 22));
    float *h = (float *)malloc((size_t)NB * 4 * 256 * 16);
    srand(1);
    for (size_t i = 0; i < (size_t)NB * 4 * 256 * 4; ++i) h[i] = (float)rand() / RAND_MAX;
    CHECK(hipMemcpy(src, h, (size_t)NB * 4 * 256 * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[4] = {"no atomic", "agent scope", "workgroup scope (L2)", "agent scope + pre-check"};
    const int nas[5] = {1, 4, 16, 64, 256};
    const int strides[4] = {1, 16, 64, 1024};            // words: 4 B, 64 B, 256 B, 4 KB
    for (int mode = 0; mode < 4; ++mode) {
        for (int ai = 0; ai < (mode == 0 ? 1 : 5); ++ai) {
            for (int si = 0; si < (mode == 0 || nas[ai] == 1 ? 1 : 4); ++si) {
                const int na = nas[ai], st = strides[si];
                if ((size_t)na * st * 4 > (1u << 22)) continue;
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CHECK(hipMemsetAsync(rec, 0, 1 << 22, 0));
                    for (int it = 0; it < 3; ++it) {            // (a warm launch first: rec holds the maximum already from the 2nd on for MODE 3)
                        if (it == 1) CHECK(hipMemsetAsync(rec, 0, 1 << 22, 0));
                        if (it == 1) CHECK(hipEventRecord(e0));
                        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(256), 0, 0, src, dst, rec, na, st);
                        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(NB), dim3(256), 0, 0, src, dst, rec, na, st);
                        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(NB), dim3(256), 0, 0, src, dst, rec, na, st);
                        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(NB), dim3(256), 0, 0, src, dst, rec, na, st);
                        if (it == 1) CHECK(hipEventRecord(e1));
                    }
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipDeviceSynchronize());
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                printf("%-26s %3d words, stride %5d B: %7.1f us per launch\n", names[mode], na, st * 4, best * 1e3);
            }
        }
    }
    return 0;
}
