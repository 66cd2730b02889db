// ubench_tr.hip — hardware facts the f16x2 bins / Self Query Layer kernels rest on (dev tool, round 5)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_tr.hip -o /tmp/ubench_tr && /tmp/ubench_tr
// 1. ds_read_b64_tr_b16: which LDS element does lane l receive in slot j, given per-lane addresses?  Expectation (guide T10): a
//    16-lane group reads a [4][16] block — lane t addresses 4 contiguous b16 = block row t / 4, columns 4 (t % 4) .. + 3 — and lane c
//    receives column c: {block[0][c], block[1][c], block[2][c], block[3][c]}.
// 2. v_mfma_f32_32x32x16_f16 operand layout: lane l holds A[l % 32][8 (l / 32) + j] and B[8 (l / 32) + j][l % 32], j = 0..7; the
//    accumulator register r of lane l is C[(r & 3) + 8 (r >> 2) + 4 (l / 32)][l % 32].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// LDS image: 64 rows x 64 columns of b16, element (r, c) holds the value 64 r + c.  Lane l (group g = l / 16, t = l % 16) addresses
// row 4 g + t / 4, columns 4 (t % 4)..+3.
__global__ void k_tr(unsigned short *out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, t = l & 15;
    const unsigned addr = (unsigned)(size_t)(img) + ((4 * g + (t >> 2)) * 64 + 4 * (t & 3)) * 2;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[l * 4 + 1] = (unsigned short)(v.x >> 16);
    out[l * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[l * 4 + 3] = (unsigned short)(v.y >> 16);
}

__global__ void k_mfma(const float *A, const float *B, float *C) {          // A [32][16], B [16][32] row-major, small integers
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[i * 16 + 8 * h + j];
        b[j] = (_Float16)B[(8 * h + j) * 32 + i];
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

int main() {
    unsigned short *d_out, h_out[256];
    CHECK(hipMalloc(&d_out, sizeof(h_out)));
    k_tr<<<1, 64>>>(d_out);
    CHECK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, c = l & 15;
        for (int j = 0; j < 4; ++j) {
            const int want = (4 * g + j) * 64 + c;               // block[j][c] of group g's [4][16] block
            if (h_out[l * 4 + j] != want) ++bad;
        }
    }
    printf("ds_read_b64_tr_b16: %s (%d mismatches against 'lane c of a 16-lane group receives column c of its [4][16] block')\n", bad ? "DIFFERENT" : "as expected", bad);
    if (bad)
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h_out[l * 4] / 64, h_out[l * 4] % 64, h_out[l * 4 + 1] / 64, h_out[l * 4 + 1] % 64,
                   h_out[l * 4 + 2] / 64, h_out[l * 4 + 2] % 64, h_out[l * 4 + 3] / 64, h_out[l * 4 + 3] % 64);
    float hA[32 * 16], hB[16 * 32], hC[32 * 32], *dA, *dB, *dC;
    for (int i = 0; i < 32 * 16; ++i) { hA[i] = (float)((i * 7 + 3) % 11 - 5); hB[i] = (float)((i * 5 + 1) % 13 - 6); }
    CHECK(hipMalloc(&dA, sizeof(hA))); CHECK(hipMalloc(&dB, sizeof(hB))); CHECK(hipMalloc(&dC, sizeof(hC)));
    CHECK(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
    k_mfma<<<1, 64>>>(dA, dB, dC);
    CHECK(hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost));
    bad = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            float s = 0;
            for (int k = 0; k < 16; ++k) s += hA[m * 16 + k] * hB[k * 32 + n];
            if (s != hC[m * 32 + n]) ++bad;
        }
    printf("v_mfma_f32_32x32x16_f16 operand / accumulator layout: %s (%d of 1024 entries differ)\n", bad ? "DIFFERENT" : "as expected", bad);
    return 0;
}
