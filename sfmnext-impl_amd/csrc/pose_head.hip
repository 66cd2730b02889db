// pose_head.hip — the tail of PoseCNN (reference networks/pose_cnn.py:40-45): pose_conv (1x1, 256 -> 6(F-1) channels) followed
// by .mean(3).mean(2) and the 0.01 scale, as one launch each way.  The 1x1 convolution and the spatial mean commute
// (mean_p(W x_p + b) = W mean_p(x_p) + b), so the kernel averages the [h*w, C] feature rows first and applies the J x C matrix
// once per image; a J = 6 output-channel convolution is no shape for the matrix cores (and was the last vendor-library
// convolution of the step).  Latency-bound: B workgroups, C threads.
#include "sqd_common.h"

namespace {
using namespace sqd;
constexpr int MAXJ = 16;

// Output layout: out2 == NULL: out [B][J].  out2 != NULL (J = 6 F): the reference's split of out.view(-1, F, 1, 6) into
// axisangle = [..., :3] and translation = [..., 3:] (pose_cnn.py:44-45) as two dense tensors, out = axisangle [B][F][3],
// out2 = translation [B][F][3] — what sqd_pose_mats_fwd reads, without the slice / cat / contiguous copies in between.
__device__ __forceinline__ size_t pose_slot(int b, int j, int J, bool planar) {
    return planar ? ((size_t)b * (J / 6) + j / 6) * 3 + (j % 6) % 3 : (size_t)b * J + j;
}

// x [B][h][w][C] (channels-last), W [J][C], bias [J] -> out = scale * (W . mean + bias), mean [B][C] (saved for the backward)
__global__ __launch_bounds__(256) void pose_head_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ out,
                                                            float *__restrict__ out2, float *__restrict__ mean, int h, int w, int C,
                                                            int J, float scale) {
    __shared__ float red[4][MAXJ];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) acc[j] = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        // .mean(3) then .mean(2): row means first (reference order)
        float m = 0.f;
        for (int y = 0; y < h; ++y) {
            float s = 0.f;
            for (int xx = 0; xx < w; ++xx) s += x[(((size_t)b * h + y) * w + xx) * C + c];
            m += s / (float)w;
        }
        m /= (float)h;
        mean[(size_t)b * C + c] = m;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
            if (j < J) acc[j] = fmaf(W[(size_t)j * C + c], m, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const float v = wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < J) {
        const int j = threadIdx.x;
        float *dst = (out2 && j % 6 >= 3) ? out2 : out;
        dst[pose_slot(b, j, J, out2 != nullptr)] = scale * ((((red[0][j] + red[1][j]) + red[2][j]) + red[3][j]) + bias[j]);
    }
}

// g (and g2: the forward's layouts) -> dx [B][h][w][C] (every pixel of an image gets the same row), dWpart [B][J][C],
// dbpart [B][J rounded up to 4]
__global__ __launch_bounds__(256) void pose_head_bwd_kernel(const float *__restrict__ g, const float *__restrict__ g2,
                                                            const float *__restrict__ W,
                                                            const float *__restrict__ mean, float *__restrict__ dx,
                                                            float *__restrict__ dWpart, float *__restrict__ dbpart, int P, int C, int J,
                                                            float scale) {
    const int b = blockIdx.x;
    float gj[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) gj[j] = j < J ? scale * ((g2 && j % 6 >= 3) ? g2 : g)[pose_slot(b, j, J, g2 != nullptr)] : 0.f;
    const int JP = (J + 3) & ~3;                    // rows of 4-float groups for sqd_colsum_multi; the padding columns are zero
    if (threadIdx.x < JP) {
        const int j = threadIdx.x;
        dbpart[(size_t)b * JP + j] = j < J ? scale * ((g2 && j % 6 >= 3) ? g2 : g)[pose_slot(b, j, J, g2 != nullptr)] : 0.f;
    }
    const float ip = 1.f / (float)P;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = mean[(size_t)b * C + c];
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
            if (j < J) {
                d = fmaf(gj[j], W[(size_t)j * C + c], d);
                dWpart[((size_t)b * J + j) * C + c] = gj[j] * m;
            }
        d *= ip;
        for (int p = 0; p < P; ++p) dx[((size_t)b * P + p) * C + c] = d;
    }
}
}  // namespace

extern "C" int sqd_pose_head_fwd(const float *x, const float *W, const float *bias, float *out, float *out2, float *mean, int B, int h,
                                 int w, int C, int J, float scale, void *stream) {
    SQD_CHECK_ARG(x && W && bias && out && mean && B > 0 && h > 0 && w > 0 && C > 0 && J >= 1 && J <= MAXJ, "sqd_pose_head_fwd: bad arguments (J <= 16)");
    SQD_CHECK_ARG(!out2 || J % 6 == 0, "sqd_pose_head_fwd: the axisangle / translation split needs J = 6 F");
    (void)hipGetLastError();
    hipLaunchKernelGGL(pose_head_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, W, bias, out, out2, mean, h, w, C, J, scale);
    SQD_CHECK_LAUNCH("sqd_pose_head_fwd");
    return SQD_OK;
}

extern "C" int sqd_pose_head_bwd(const float *g, const float *g2, const float *W, const float *mean, float *dx, float *dWpart,
                                 float *dbpart, int B, int P, int C, int J, float scale, void *stream) {
    SQD_CHECK_ARG(g && W && mean && dx && dWpart && dbpart && B > 0 && P > 0 && C > 0 && J >= 1 && J <= MAXJ, "sqd_pose_head_bwd: bad arguments (J <= 16)");
    SQD_CHECK_ARG(!g2 || J % 6 == 0, "sqd_pose_head_bwd: the axisangle / translation split needs J = 6 F");
    (void)hipGetLastError();
    hipLaunchKernelGGL(pose_head_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, g, g2, W, mean, dx, dWpart, dbpart, P, C, J, scale);
    SQD_CHECK_LAUNCH("sqd_pose_head_bwd");
    return SQD_OK;
}
