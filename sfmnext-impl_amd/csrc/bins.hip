// bins.hip — the adaptive-bins depth head of Depth_Decoder_QueryTr as one kernel pair.
// replaces: reference networks/depth_decoder_QTR.py:61-70
//     out  = softmax_d( conv1x1(energy_maps) )            convert_to_prob: [B,Q,h,w] -> [B,D,h,w]
//     pred = sum_d out[d] * centers[b, d]                  centers = mid-points of the cumulated bin widths
// (conv 1x1 + channel softmax + expectation: three tensor passes over [B,D,h,w] and, with ATen, four layout copies).
//
// forward : per 32-pixel tile  logits[d, p] = W[d, :] . E[:, p] + bias[d]  on v_mfma_f32_32x32x2_f32 with the energy
//           maps read straight from their planar [B,Q,N] layout as the B operand (lane = pixel: coalesced 128-byte
//           rows), W held in LDS as the A operand.  The accumulator layout leaves each pixel's logits in one lane
//           pair (lane, lane^32), so the softmax and the expectation are in-register; only pred [B,N] is written.
// backward: recomputes the tile's logits, then
//           dlogit[d,p] = prob[d,p] * g[p] * (c[d] - pred[p])
//           dE[q,p]     = sum_d W[d,q] dlogit[d,p]     MFMA fed from the accumulator registers (contraction over the
//                                                      producer's row dimension — no LDS round trip)
//           dW[d,q]     = sum_p dlogit[d,p] E[q,p]     MFMA, dlogit transposed through a per-wave LDS tile, E re-read
//                                                      as float4 along the pixels
//           dbias[d], dcenters[b,d]                    row sums taken while the dW operand is read
//           per-workgroup partials + one deterministic reduction kernel.
// Roofline: HBM — forward reads 4*Q B/px, writes 4; backward reads 4*Q (+4*Q from L2), writes 4*Q.
#include "sqd_common.h"
#include "sqd_f16x2.h"

namespace {
using namespace sqd;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PITCH = 36;                        // floats per row of the per-wave [d][32 pixels] tile (16-byte aligned rows)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// uniform base + 32-bit byte offset (saddr + voffset addressing: no 64-bit address arithmetic per access)
__device__ __forceinline__ float ldg(const float *__restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ float4 ldg4(const float *__restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void stg(float *__restrict__ base, unsigned byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// raw buffer accesses on one image's energy planes [Q][N]: a plane index beyond Q is beyond the descriptor's extent, a lane
// beyond the last pixel starts from a 2 GiB offset — both read zeros / drop the store, no branch around the access
typedef int bins_i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned BINS_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bins_rsrc(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float ldb(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

struct BinsDims {
    int B, Q, D, N;
};

template <int DT, int QT>
__device__ __forceinline__ void stage_weights(float *Wl, float *bl, const float *__restrict__ W, const float *__restrict__ bias, int D,
                                              int Q) {
    constexpr int QP = QT * 32 + 1;
    for (int idx = threadIdx.x; idx < DT * 32 * QP; idx += 256) {
        const int d = idx / QP, q = idx - d * QP;
        Wl[idx] = (d < D && q < Q) ? W[d * Q + q] : 0.f;
    }
    for (int d = threadIdx.x; d < DT * 32; d += 256) bl[d] = d < D ? bias[d] : -INFINITY;      // (see tile_softmax)
}

// logits of the 32 pixels p0..p0+31 of image b: acc[dt][r] = row d = dt*32 + acc_row(r, lane>>5), column = lane & 31
template <int DT, int QT>
__device__ __forceinline__ void tile_logits(const float *Wl, const float *__restrict__ Eb, int Q, int N, int p, bool pv, int lane,
                                            f32x16 (&acc)[DT]) {
    constexpr int QP = QT * 32 + 1;
    const int i = lane & 31, kk = lane >> 5;
    constexpr int G = 8, NG = QT * 16 / G;                 // energy planes are fetched G k-steps (2G planes) at a time
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    const unsigned step = 2u * N * 4u;
    const __amdgpu_buffer_rsrc_t e_r = bins_rsrc(Eb, (unsigned)(Q * N) * 4u);
    unsigned off = pv ? ((unsigned)kk * N + p) * 4u : BINS_OOB;
    auto fetch = [&](int g, float *e) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            e[u] = ldb(e_r, off);                           // plane 2(gG+u)+kk >= Q: beyond the extent
            off += step;
        }
    };
    float e0[G], e1[G];
    fetch(0, e0);
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
        fetch(g + 1, e1);                                   // NG is even; planes past Q are masked
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma32(Wl[(dt * 32 + i) * QP + 2 * (g * G + u) + kk], e0[u], acc[dt]);
        if (g + 2 < NG) fetch(g + 2, e0);
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma32(Wl[(dt * 32 + i) * QP + 2 * ((g + 1) * G + u) + kk], e1[u], acc[dt]);
    }
}

// in: acc * scale = the products W.E; out: acc = exp(logit - max) (0 for padded d), returns 1/sum and the expectation.  bl[d] = bias[d], -inf
// for padded rows (their products are 0: the logit of a padded row is -inf and its exponential 0, no compare needed); the four rows of a
// register group r = 4 g .. 4 g + 3 are consecutive d: bias and centres come as 16-byte LDS reads.  (Round 5: the per-row `d < D ? acc + bl[d]
// : -inf` of rounds 1-4 compiled to 64 exec-masked branches, each around one ds_read_b32 and its wait — ~10 000 cycles per tile.)
template <int DT>
__device__ __forceinline__ void tile_softmax(f32x16 (&acc)[DT], const float *bl, const float *cl, int lane, float scale, float &inv_sum,
                                             float &pred) {
    const int h = lane >> 5;
    float m = -INFINITY;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4 *>(bl + dt * 32 + 8 * g + 4 * h);
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = fmaf(acc[dt][4 * g + j], scale, bv[j]);      // (scale is a power of two: the same bits as mul + add)
                acc[dt][4 * g + j] = v;
                m = fmaxf(m, v);
            }
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float se = 0.f, dot = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 c4 = *reinterpret_cast<const float4 *>(cl + dt * 32 + 8 * g + 4 * h);
            const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ex = __expf(acc[dt][4 * g + j] - m);
                acc[dt][4 * g + j] = ex;
                se += ex;
                dot = fmaf(ex, cv[j], dot);
            }
        }
    se += __shfl_xor(se, 32, 64);
    dot += __shfl_xor(dot, 32, 64);
    inv_sum = 1.f / se;
    pred = dot * inv_sum;
}

template <int DT, int QT>
__global__ __launch_bounds__(256) void bins_fwd_kernel(const float *__restrict__ E, const float *__restrict__ W,
                                                       const float *__restrict__ bias, const float *__restrict__ centers,
                                                       float *__restrict__ pred_out, BinsDims dm, int tiles_per_image) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int QP = QT * 32 + 1;
    float *Wl = smem, *bl = Wl + DT * 32 * QP, *cl = bl + DT * 32;
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_weights<DT, QT>(Wl, bl, W, bias, dm.D, dm.Q);
    for (int d = threadIdx.x; d < DT * 32; d += 256) cl[d] = d < dm.D ? centers[b * dm.D + d] : 0.f;
    __syncthreads();
    const float *Eb = E + (size_t)b * dm.Q * dm.N;
    for (int tile = blockIdx.x * 4 + wave; tile < tiles_per_image; tile += gridDim.x * 4) {
        const int p = tile * 32 + (lane & 31);
        const bool pv = p < dm.N;
        f32x16 acc[DT];
        tile_logits<DT, QT>(Wl, Eb, dm.Q, dm.N, p, pv, lane, acc);
        float inv_sum, pred;
        tile_softmax<DT>(acc, bl, cl, lane, 1.f, inv_sum, pred);
        if (pv && lane < 32) pred_out[(size_t)b * dm.N + p] = pred;
    }
}

template <int DT, int QT>
__global__ __launch_bounds__(256) void bins_bwd_kernel(const float *__restrict__ E, const float *__restrict__ W,
                                                       const float *__restrict__ bias, const float *__restrict__ centers,
                                                       const float *__restrict__ g_pred, float *__restrict__ dE,
                                                       float *__restrict__ part_w, float *__restrict__ part_v, BinsDims dm,
                                                       int tiles_per_image) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int QP = QT * 32 + 1, DP = DT * 32;
    float *Wl = smem, *bl = Wl + DP * QP, *cl = bl + DP;
    float *predl = cl + DP;                                   // [4 waves][32]
    float *tiles = predl + 128;                               // [4 waves][DP][PITCH]: prob*g of the wave's tile, [d][pixel]
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    stage_weights<DT, QT>(Wl, bl, W, bias, dm.D, dm.Q);
    for (int d = threadIdx.x; d < DP; d += 256) cl[d] = d < dm.D ? centers[b * dm.D + d] : 0.f;
    __syncthreads();
    const float *Eb = E + (size_t)b * dm.Q * dm.N;
    float *dEb = dE + (size_t)b * dm.Q * dm.N;
    const __amdgpu_buffer_rsrc_t eb_r = bins_rsrc(Eb, (unsigned)(dm.Q * dm.N) * 4u), de_r = bins_rsrc(dEb, (unsigned)(dm.Q * dm.N) * 4u);
    float *tl = tiles + wave * DP * PITCH, *pl = predl + wave * 32;
    const bool vec_ok = (dm.N & 3) == 0;

    // The dW accumulator of a whole [D][Q] tile is DT * QT * 16 registers — all 256 at D = Q = 128 (round 2: 512 registers + 340 B
    // of scratch, 1.27 ms).  So the waves share it: wave w owns the 32 filters of d-tile wd = w % DT (QT * 16 registers) and, after
    // a workgroup barrier, accumulates them over the pixel tiles of its group of DT waves (wg = w / DT) from the prob*g tiles those
    // waves left in LDS; logits, softmax and the dE product stay per wave on its own pixel tile.
    const int wd = wave % DT, wgp = wave / DT;
    f32x16 accW[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[qt][r] = 0.f;
    float dbia = 0.f, dcen = 0.f;
    const float ci = cl[wd * 32 + i];
    const int iters = (tiles_per_image + gridDim.x * 4 - 1) / (gridDim.x * 4);

    for (int it = 0; it < iters; ++it) {                       // (uniform trip count: the loop holds workgroup barriers)
        const int tile0 = (it * gridDim.x + blockIdx.x) * 4, tile = tile0 + wave;
        const int p0 = tile * 32, p = p0 + i;
        const bool pv = p < dm.N;                               // (a tile beyond the image: every access masked, zero contributions)
        {
            f32x16 acc[DT];
            tile_logits<DT, QT>(Wl, Eb, dm.Q, dm.N, p, pv, lane, acc);
            float inv_sum, pred;
            tile_softmax<DT>(acc, bl, cl, lane, 1.f, inv_sum, pred);
            const float g = pv ? g_pred[(size_t)b * dm.N + p] * inv_sum : 0.f;      // g[p] / sum: acc holds un-normalised exp
            if (lane < 32) pl[i] = pred;
            // prob*g -> LDS tile [d][pixel] (operand of the dW product); dlogit stays in acc (operand of the dE product)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = dt * 32 + acc_row(r, h);
                    const float pg = acc[dt][r] * g;
                    tl[d * PITCH + i] = pg;
                    acc[dt][r] = pg * (cl[d] - pred);
                }
            // ---- dE[q, p] = sum_d W[d, q] * dlogit[d, p]
#pragma unroll 1
            for (int qt = 0; qt < QT; ++qt) {
                f32x16 accE;
#pragma unroll
                for (int r = 0; r < 16; ++r) accE[r] = 0.f;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        accE = mfma32(Wl[(dt * 32 + acc_row(r, h)) * QP + qt * 32 + i], acc[dt][r], accE);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 32 + acc_row(r, h);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(accE[r]), de_r, pv ? ((unsigned)q * dm.N + p) * 4u : BINS_OOB, 0, 0);
                }
            }
        }
        __syncthreads();                                        // every wave's prob*g tile and pred row are in LDS
        // ---- dW[d, q] += sum_p dlogit[d, p] * E[q, p] for d-tile wd over the group's DT pixel tiles; k-step (gq, e): half-wave 0 takes
        // pixel 8gq+e, half-wave 1 pixel 8gq+4+e
#pragma unroll 1
        for (int tt = 0; tt < DT; ++tt) {
            const int wsrc = wgp * DT + tt;                     // the wave whose pixel tile this is
            const int q0 = (tile0 + wsrc) * 32;
            const float *tls = tiles + wsrc * DP * PITCH, *pls = predl + wsrc * 32;
#pragma unroll 1
            for (int gq = 0; gq < 4; ++gq) {
                const int px = 8 * gq + 4 * h;
                const float4 pr4 = *reinterpret_cast<const float4 *>(pls + px);
                const float prv[4] = {pr4.x, pr4.y, pr4.z, pr4.w};
                float dlg[4], ev[QT][4];
                const float4 t4 = *reinterpret_cast<const float4 *>(tls + (wd * 32 + i) * PITCH + px);
                const float pgv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dlg[e] = pgv[e] * (ci - prv[e]);
                    dcen += pgv[e];
                    dbia += dlg[e];
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int q = qt * 32 + i;
                    const unsigned off = ((unsigned)q * dm.N + q0 + px) * 4u;
                    if (vec_ok) {                               // N % 4 == 0: a float4 lies inside a plane or beyond the last pixel
                        const bins_i32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(eb_r, q0 + px < dm.N ? off : BINS_OOB, 0, 0);
                        ev[qt][0] = __int_as_float(v4.x); ev[qt][1] = __int_as_float(v4.y);
                        ev[qt][2] = __int_as_float(v4.z); ev[qt][3] = __int_as_float(v4.w);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ev[qt][e] = ldb(eb_r, q0 + px + e < dm.N ? off + 4u * e : BINS_OOB);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) accW[qt] = mfma32(dlg[e], ev[qt][e], accW[qt]);
            }
        }
        __syncthreads();                                        // the tiles are rewritten by the next iteration
    }

    // ---- workgroup partials: the 4 / DT groups add their d-tiles into one LDS image in a fixed order (group 0 stores, the others add)
    float *red = tiles;                                        // reuse: DP x QT*32 floats <= 4 * DP * PITCH
    constexpr int QW = QT * 32;
    for (int gsel = 0; gsel < 4 / DT; ++gsel) {
        if (wgp == gsel) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *dst = red + (wd * 32 + acc_row(r, h)) * QW + qt * 32 + i;
                    *dst = (gsel == 0 ? 0.f : *dst) + accW[qt][r];
                }
        }
        __syncthreads();
    }
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    float *pw = part_w + (size_t)wg * dm.D * dm.Q;
    for (int idx = threadIdx.x; idx < dm.D * dm.Q; idx += 256) {
        const int d = idx / dm.Q, q = idx - d * dm.Q;
        pw[idx] = red[d * QW + q];
    }
    // dbias / dcenters: lane (d = wd*32 + i, half h) holds the sum over its half of the pixels of its group's tiles
    __syncthreads();
    float *vred = red;                                         // [4 / DT groups][2][DP]
    {
        const float sb = dbia + __shfl_xor(dbia, 32, 64), sc = dcen + __shfl_xor(dcen, 32, 64);
        if (h == 0) {
            vred[(wgp * 2 + 0) * DP + wd * 32 + i] = sb;
            vred[(wgp * 2 + 1) * DP + wd * 32 + i] = sc;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * DP; idx += 256) {
        const int which = idx / DP, d = idx - which * DP;
        if (d < dm.D) {
            float sum = 0.f;
            for (int w = 0; w < 4 / DT; ++w) sum += vred[(w * 2 + which) * DP + d];
            part_v[((size_t)wg * 2 + which) * dm.D + d] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the same kernels on two-term fp16 operands (sqd_f16x2.h).  The fp32 kernels above spend DT * QT * 16 matrix instructions of
// 64 cycles on a tile's logits (16 384 cycles at D = Q = 128: bins_fwd ran at 0.43, bins_bwd at 0.48 of the fp32 matrix peak, three
// rounds without movement); v_mfma_f32_32x32x16_f16 does 8x the work of v_mfma_f32_32x32x2_f32 in half its cycles, so three of them per
// product (h h + h l + l h) are 5.3x less matrix time and the kernels become what their header says: HBM-bound.
//   W      d-major in LDS as fp16 high / low planes [DP][QH] (row pitch 16 bytes beyond a multiple of 64: ds_read_b128 of 16 consecutive rows
//          is conflict-free), scaled by one power of two from max |W| (taken while staging);
//   logits A = W rows (ds_read_b128: 8 consecutive q), B = energy planes read straight from [B,Q,N] — k-slot (h, j) of step ks is plane
//          16 ks + 8 h + j — split in registers with a PER-PIXEL scale (the scale of B may vary along its columns);
//   dE     contraction over d: B = dlogit from the accumulator registers of the logits (k-slot (h, j) of step (dt, u) is row
//          dt 32 + 16 u + 4 h + (j & 3) + 8 (j >> 2): the lane's own registers 8 u .. 8 u + 7), per-pixel scale again; A = W^T, which the
//          d-major LDS image yields through ds_read_b64_tr_b16 (the LDS transpose read: a 16-lane group reads a [4 d][16 q] block, each lane
//          receives one q column) — one LDS image serves both contractions;
//   dW     contraction over the pixels, accumulated over all tiles of the workgroup: stays on the fp32 instruction (its operand scales
//          would have to be uniform over the whole launch), a third of the old matrix time.
constexpr unsigned BINS_H_MAX_BYTES = 0x7fffffffu;             // Q N 4 must stay below 2 GiB (masked lanes start at BINS_OOB and add plane steps)

template <int DT, int QT>
__device__ __forceinline__ unsigned stage_weights_h(unsigned short *Wh, unsigned short *Wlo, float *bl, unsigned *red4,
                                                    const float *__restrict__ W, const float *__restrict__ bias, int D, int Q) {
    constexpr int QH = QT * 32 + 8, DP = DT * 32;
    unsigned m = 0u;
    for (int idx = threadIdx.x; idx < D * Q; idx += 256) m = max(m, abs_bits(W[idx]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = m;
    __syncthreads();
    m = max(max(red4[0], red4[1]), max(red4[2], red4[3]));
    const unsigned be = h2_scale_exp(m);
    const float s = h2_scale(be);
    for (int idx = threadIdx.x; idx < DP * QT * 8; idx += 256) {
        const int d = idx / (QT * 8), q0 = 4 * (idx - d * (QT * 8));
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (d < D && q0 + k < Q) ? W[d * Q + q0 + k] : 0.f;
        unsigned h0, h1, l0, l1;
        h2_split2(v[0], v[1], s, h0, l0);
        h2_split2(v[2], v[3], s, h1, l1);
        *reinterpret_cast<uint2 *>(Wh + d * QH + q0) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(Wlo + d * QH + q0) = make_uint2(l0, l1);
    }
    for (int d = threadIdx.x; d < DP; d += 256) bl[d] = d < D ? bias[d] : -INFINITY;      // (see tile_softmax)
    return be;
}

// logits of the 32 pixels p0..p0+31 of one image, as tile_logits (acc[dt][r] = row dt*32 + acc_row(r, lane>>5), column lane & 31)
template <int DT, int QT>
__device__ __forceinline__ float tile_logits_h(const unsigned short *Wh, const unsigned short *Wlo, unsigned beW, __amdgpu_buffer_rsrc_t e_r, int N, int p,
                                              bool pv, int lane, f32x16 (&acc)[DT]) {
    constexpr int QH = QT * 32 + 8, KS = QT * 2;
    const int i = lane & 31, h = lane >> 5;
    const unsigned step = (unsigned)N * 4u;
    float e[KS][8];
    unsigned off = pv ? ((unsigned)(8 * h) * N + p) * 4u : BINS_OOB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) e[ks][j] = ldb(e_r, off + (unsigned)j * step);       // plane 16 ks + 8 h + j >= Q: beyond the extent
        off += 16u * step;
    }
    unsigned m = 0u;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) m = max(m, abs_bits(e[ks][j]));
    m = max(m, (unsigned)__shfl_xor((int)m, 32, 64));
    const unsigned beP = h2_scale_exp(m);
    const float sp = h2_scale(beP);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    // the A operands of step ks + 1 are fetched from LDS while the matrix instructions of step ks run (left to itself the compiler
    // emits read - wait - multiply 64 times per tile: the waves of round 5's first version spent half of their life at those waits)
    u32x4 ah[2][DT], al[2][DT];
    const unsigned short *wh_lane = Wh + i * QH + 8 * h, *wl_lane = Wlo + i * QH + 8 * h;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        ah[0][dt] = *reinterpret_cast<const u32x4 *>(wh_lane + dt * 32 * QH);
        al[0][dt] = *reinterpret_cast<const u32x4 *>(wl_lane + dt * 32 * QH);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                ah[(ks + 1) & 1][dt] = *reinterpret_cast<const u32x4 *>(wh_lane + dt * 32 * QH + (ks + 1) * 16);
                al[(ks + 1) & 1][dt] = *reinterpret_cast<const u32x4 *>(wl_lane + dt * 32 * QH + (ks + 1) * 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // (the reads stay above the step's arithmetic)
        u32x4 bh, bl_;
        h2_split8(e[ks], sp, bh, bl_);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma_h16(ah[ks & 1][dt], bh, acc[dt]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma_h16(ah[ks & 1][dt], bl_, acc[dt]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma_h16(al[ks & 1][dt], bh, acc[dt]);
    }
    return h2_inv_scale(beW) * h2_inv_scale(beP);              // acc holds s_W s_p times the products
}

template <int DT, int QT>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void bins_fwd_h_kernel(const float *__restrict__ E, const float *__restrict__ W,
                                                         const float *__restrict__ bias, const float *__restrict__ centers,
                                                         float *__restrict__ pred_out, BinsDims dm, int tiles_per_image) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int QH = QT * 32 + 8, DP = DT * 32;
    unsigned short *Wh = reinterpret_cast<unsigned short *>(smem), *Wlo = Wh + DP * QH;
    float *bl = reinterpret_cast<float *>(Wlo + DP * QH), *cl = bl + DP;
    unsigned *red4 = reinterpret_cast<unsigned *>(cl + DP);
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned beW = stage_weights_h<DT, QT>(Wh, Wlo, bl, red4, W, bias, dm.D, dm.Q);
    for (int d = threadIdx.x; d < DP; d += 256) cl[d] = d < dm.D ? centers[b * dm.D + d] : 0.f;
    __syncthreads();
    const float *Eb = E + (size_t)b * dm.Q * dm.N;
    const __amdgpu_buffer_rsrc_t e_r = bins_rsrc(Eb, (unsigned)(dm.Q * dm.N) * 4u);
    const int iters = (tiles_per_image + gridDim.x * 4 - 1) / (gridDim.x * 4);
    for (int it = 0; it < iters; ++it) {
        const int tile = (it * gridDim.x + blockIdx.x) * 4 + wave;
        const int p = tile * 32 + (lane & 31);
        const bool pv = p < dm.N;
#ifdef SQD_BINS_FWD_BARRIER
        __syncthreads();          // the four waves issue their 4 x 128 bytes of every plane together
#endif
        f32x16 acc[DT];
        const float lscale = tile_logits_h<DT, QT>(Wh, Wlo, beW, e_r, dm.N, p, pv, lane, acc);
        float inv_sum, pred;
        tile_softmax<DT>(acc, bl, cl, lane, lscale, inv_sum, pred);
        if (pv && lane < 32) pred_out[(size_t)b * dm.N + p] = pred;
    }
}

template <int DT, int QT>
__global__ __launch_bounds__(256) void bins_bwd_h_kernel(const float *__restrict__ E, const float *__restrict__ W,
                                                         const float *__restrict__ bias, const float *__restrict__ centers,
                                                         const float *__restrict__ g_pred, float *__restrict__ dE,
                                                         float *__restrict__ part_w, float *__restrict__ part_v, BinsDims dm,
                                                         int tiles_per_image) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int QH = QT * 32 + 8, DP = DT * 32;
    unsigned short *Wh = reinterpret_cast<unsigned short *>(smem), *Wlo = Wh + DP * QH;
    float *bl = reinterpret_cast<float *>(Wlo + DP * QH), *cl = bl + DP;
    float *predl = cl + DP;                                   // [4 waves][32]
    float *tiles = predl + 128;                               // [4 waves][DP][PITCH]: prob*g of the wave's tile, [d][pixel]
    unsigned *red4 = reinterpret_cast<unsigned *>(tiles);     // (staging only)
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const unsigned beW = stage_weights_h<DT, QT>(Wh, Wlo, bl, red4, W, bias, dm.D, dm.Q);
    for (int d = threadIdx.x; d < DP; d += 256) cl[d] = d < dm.D ? centers[b * dm.D + d] : 0.f;
    __syncthreads();
    const float *Eb = E + (size_t)b * dm.Q * dm.N;
    float *dEb = dE + (size_t)b * dm.Q * dm.N;
    const __amdgpu_buffer_rsrc_t eb_r = bins_rsrc(Eb, (unsigned)(dm.Q * dm.N) * 4u), de_r = bins_rsrc(dEb, (unsigned)(dm.Q * dm.N) * 4u);
    float *tl = tiles + wave * DP * PITCH, *pl = predl + wave * 32;
    const bool vec_ok = (dm.N & 3) == 0;
    const int wd = wave % DT, wgp = wave / DT;
    f32x16 accW[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[qt][r] = 0.f;
    float dbia = 0.f, dcen = 0.f;
    const float ci = cl[wd * 32 + i];
    const int iters = (tiles_per_image + gridDim.x * 4 - 1) / (gridDim.x * 4);
    // the transpose reads of the dE product: lane t of a 16-lane group addresses row t / 4, columns 4 (t % 4).. of the group's [4 d][16 q]
    // block; the group's 16 lanes are the q columns 16 ((lane >> 4) & 1) .. + 15 of the 32-wide q tile, its d rows start at 4 h
    const int tr_off = (4 * h + ((lane & 15) >> 2)) * QH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    for (int it = 0; it < iters; ++it) {                       // (uniform trip count: the loop holds workgroup barriers)
        const int tile0 = (it * gridDim.x + blockIdx.x) * 4, tile = tile0 + wave;
        const int p0 = tile * 32, p = p0 + i;
        const bool pv = p < dm.N;
        {
            f32x16 acc[DT];
            const float lscale = tile_logits_h<DT, QT>(Wh, Wlo, beW, eb_r, dm.N, p, pv, lane, acc);
            float inv_sum, pred;
            tile_softmax<DT>(acc, bl, cl, lane, lscale, inv_sum, pred);
            const float g = pv ? g_pred[(size_t)b * dm.N + p] * inv_sum : 0.f;
            if (lane < 32) pl[i] = pred;
            unsigned m = 0u;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = dt * 32 + acc_row(r, h);
                    const float pg = acc[dt][r] * g;
                    tl[d * PITCH + i] = pg;
                    const float dl = pg * (cl[d] - pred);
                    acc[dt][r] = dl;
                    m = max(m, abs_bits(dl));
                }
            // ---- dE[q, p] = sum_d W[d, q] * dlogit[d, p]: the B operand is the lane's own accumulator registers
            m = max(m, (unsigned)__shfl_xor((int)m, 32, 64));
            const unsigned beL = h2_scale_exp(m);
            const float sl = h2_scale(beL), invE = h2_inv_scale(beW) * h2_inv_scale(beL);
            u32x4 bh[DT][2], blo[DT][2];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = acc[dt][8 * u + j];
                    h2_split8(v, sl, bh[dt][u], blo[dt][u]);
                }
#pragma unroll 1
            for (int qt = 0; qt < QT; ++qt) {
                f32x16 accE;
#pragma unroll
                for (int r = 0; r < 16; ++r) accE[r] = 0.f;
                // step st = (dt, u): 16 d rows; the transpose reads of step st + 1 are issued before the matrix instructions of step st
                u32x4 ath[2], atl[2];
                auto fetch = [&](int st, u32x4 &ah_, u32x4 &al_) {
                    const int o = tr_off + (16 * st) * QH + qt * 32;
                    const uint2 h0 = lds_read_tr16(Wh + o), h1 = lds_read_tr16(Wh + o + 8 * QH);
                    const uint2 l0 = lds_read_tr16(Wlo + o), l1 = lds_read_tr16(Wlo + o + 8 * QH);
                    ah_ = (u32x4){h0.x, h0.y, h1.x, h1.y};
                    al_ = (u32x4){l0.x, l0.y, l1.x, l1.y};
                };
                fetch(0, ath[0], atl[0]);
#pragma unroll
                for (int st = 0; st < 2 * DT; ++st) {
                    if (st + 1 < 2 * DT) fetch(st + 1, ath[(st + 1) & 1], atl[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    accE = mfma_h16x2(ath[st & 1], atl[st & 1], bh[st >> 1][st & 1], blo[st >> 1][st & 1], accE);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qt * 32 + acc_row(r, h);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(accE[r] * invE), de_r, pv ? ((unsigned)q * dm.N + p) * 4u : BINS_OOB, 0, 0);
                }
            }
        }
        __syncthreads();                                        // every wave's prob*g tile and pred row are in LDS
        // ---- dW[d, q] += sum_p dlogit[d, p] * E[q, p] (fp32 instruction, see the header of this section)
#pragma unroll 1
        for (int tt = 0; tt < DT; ++tt) {
            const int wsrc = wgp * DT + tt;
            const int q0 = (tile0 + wsrc) * 32;
            const float *tls = tiles + wsrc * DP * PITCH, *pls = predl + wsrc * 32;
#pragma unroll 1
            for (int gq = 0; gq < 4; ++gq) {
                const int px = 8 * gq + 4 * h;
                const float4 pr4 = *reinterpret_cast<const float4 *>(pls + px);
                const float prv[4] = {pr4.x, pr4.y, pr4.z, pr4.w};
                float dlg[4], ev[QT][4];
                const float4 t4 = *reinterpret_cast<const float4 *>(tls + (wd * 32 + i) * PITCH + px);
                const float pgv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dlg[e] = pgv[e] * (ci - prv[e]);
                    dcen += pgv[e];
                    dbia += dlg[e];
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int q = qt * 32 + i;
                    const unsigned off = ((unsigned)q * dm.N + q0 + px) * 4u;
                    if (vec_ok) {
                        const bins_i32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(eb_r, q0 + px < dm.N ? off : BINS_OOB, 0, 0);
                        ev[qt][0] = __int_as_float(v4.x); ev[qt][1] = __int_as_float(v4.y);
                        ev[qt][2] = __int_as_float(v4.z); ev[qt][3] = __int_as_float(v4.w);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ev[qt][e] = ldb(eb_r, q0 + px + e < dm.N ? off + 4u * e : BINS_OOB);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) accW[qt] = mfma32(dlg[e], ev[qt][e], accW[qt]);
            }
        }
        __syncthreads();                                        // the tiles are rewritten by the next iteration
    }

    // ---- workgroup partials, as bins_bwd_kernel
    float *red = tiles;
    constexpr int QW = QT * 32;
    for (int gsel = 0; gsel < 4 / DT; ++gsel) {
        if (wgp == gsel) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *dst = red + (wd * 32 + acc_row(r, h)) * QW + qt * 32 + i;
                    *dst = (gsel == 0 ? 0.f : *dst) + accW[qt][r];
                }
        }
        __syncthreads();
    }
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    float *pw = part_w + (size_t)wg * dm.D * dm.Q;
    for (int idx = threadIdx.x; idx < dm.D * dm.Q; idx += 256) {
        const int d = idx / dm.Q, q = idx - d * dm.Q;
        pw[idx] = red[d * QW + q];
    }
    __syncthreads();
    float *vred = red;
    {
        const float sb = dbia + __shfl_xor(dbia, 32, 64), sc = dcen + __shfl_xor(dcen, 32, 64);
        if (h == 0) {
            vred[(wgp * 2 + 0) * DP + wd * 32 + i] = sb;
            vred[(wgp * 2 + 1) * DP + wd * 32 + i] = sc;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * DP; idx += 256) {
        const int which = idx / DP, d = idx - which * DP;
        if (d < dm.D) {
            float sum = 0.f;
            for (int w = 0; w < 4 / DT; ++w) sum += vred[(w * 2 + which) * DP + d];
            part_v[((size_t)wg * 2 + which) * dm.D + d] = sum;
        }
    }
}

// out[y][i] = sum_{s < splits} part[(y*splits + s)*n + i]: a block owns 32 columns, its 8 thread groups add the
// partials s = g, g+8, ... in order, then a fixed-order tree over the groups (deterministic)
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, int n, int splits) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    const float *src = part + (size_t)blockIdx.y * splits * n;
    float a = 0.f;
    if (i < n)
        for (int s = grp; s < splits; s += 8) a += src[(size_t)s * n + i];
    red[grp][col] = a;
    __syncthreads();
    for (int w = 4; w >= 1; w >>= 1) {
        if (grp < w) {
            a += red[grp + w][col];
            red[grp][col] = a;
        }
        __syncthreads();
    }
    if (grp == 0 && i < n) out[(size_t)blockIdx.y * n + i] = a;
}
// the backward's three tail sums in ONE launch (round 5; three launches before, same bits): blocks [0, nbw) sum the g_weight partials over
// all workgroups (rows_reduce's arithmetic); a block of the rest owns 32 of the 2 D vector columns and walks the images in order — per image
// the sum over its workgroups' partials (same arithmetic), columns < D added up over the images into g_bias, columns >= D written to
// g_centers[b]
__global__ __launch_bounds__(256) void bins_tail_kernel(const float *__restrict__ part_w, float *__restrict__ g_weight, int nw, int splits_w, int nbw,
                                                        const float *__restrict__ part_v, float *__restrict__ g_bias, float *__restrict__ g_centers,
                                                        int B, int D, int splits_v) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    auto reduce = [&](const float *src, int n, int i, int splits) {
        float a = 0.f;
        if (i < n)
            for (int s = grp; s < splits; s += 8) a += src[(size_t)s * n + i];
        red[grp][col] = a;
        __syncthreads();
        for (int w = 4; w >= 1; w >>= 1) {
            if (grp < w) {
                a += red[grp + w][col];
                red[grp][col] = a;
            }
            __syncthreads();
        }
        return a;                                              // (valid in group 0)
    };
    if ((int)blockIdx.x < nbw) {
        const int i = blockIdx.x * 32 + col;
        const float a = reduce(part_w, nw, i, splits_w);
        if (grp == 0 && i < nw) g_weight[i] = a;
        return;
    }
    const int n = 2 * D, i = ((int)blockIdx.x - nbw) * 32 + col;
    float bias_acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float a = reduce(part_v + (size_t)b * splits_v * n, n, i, splits_v);
        if (grp == 0 && i < n) {
            if (i < D) bias_acc += a;
            else g_centers[(size_t)b * D + (i - D)] = a;
        }
        __syncthreads();
    }
    if (grp == 0 && i < D) g_bias[i] = bias_acc;
}
// tmp [B][2][D] (per-image sums of the dbias / dcenters partials) -> dbias [D] (summed over the images), dcenters [B][D]
__global__ __launch_bounds__(256) void bins_vec_finalize_kernel(const float *__restrict__ tmp, float *__restrict__ dbias,
                                                                float *__restrict__ dcenters, int B, int D) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < D) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += tmp[(size_t)b * 2 * D + idx];
        dbias[idx] = s;
    } else if (idx < D + B * D) {
        const int t = idx - D, b = t / D, d = t - b * D;
        dcenters[t] = tmp[(size_t)b * 2 * D + D + d];
    }
}

struct BinsPlan {
    int dt, wg_per_image, tiles_per_image, wg_fwd_h, wg_bwd_h;
    size_t smem_fwd, smem_bwd, smem_fwd_h, smem_bwd_h;
};
BinsPlan plan_bins(int B, int Q, int D, int N) {
    BinsPlan p;
    p.dt = (Q <= 64 && D <= 64) ? 2 : 4;
    p.tiles_per_image = (N + 31) / 32;
    // ~3 workgroups per CU overall, at least 2 tiles per wave
    int wg = (768 + B - 1) / B;
    const int max_wg = (p.tiles_per_image + 7) / 8;
    if (wg > max_wg) wg = max_wg;
    if (wg < 1) wg = 1;
    p.wg_per_image = wg;
    // the two-term kernels: ONE round of resident workgroups (every workgroup stages W once, 768 workgroups on 512 slots were 1.5 rounds) —
    // forward 2 (D, Q > 64: 72 KB of LDS) or 4 per CU, backward 1 or 2 (its fp32 tiles)
    auto one_round = [&](int per_cu) {
        int w = (256 * per_cu) / B;
        if (w > (p.tiles_per_image + 3) / 4) w = (p.tiles_per_image + 3) / 4;
        return w < 1 ? 1 : w;
    };
    p.wg_fwd_h = one_round(p.dt == 2 ? 4 : 2);
    p.wg_bwd_h = one_round(p.dt == 2 ? 2 : 1);
    const int DP = p.dt * 32, QP = p.dt * 32 + 1, QH = p.dt * 32 + 8;
    p.smem_fwd = (size_t)(DP * QP + 2 * DP) * 4;
    p.smem_bwd = (size_t)(DP * QP + 2 * DP + 128 + 4 * DP * PITCH) * 4;
    p.smem_fwd_h = (size_t)2 * DP * QH * 2 + (size_t)(2 * DP + 4) * 4;
    p.smem_bwd_h = (size_t)2 * DP * QH * 2 + (size_t)(2 * DP + 128 + 4 * DP * PITCH) * 4;
    return p;
}
// 1 (default): two-term fp16 operands for the logits and the dE product (bins_*_h_kernel); 0: the fp32 matrix instruction throughout
int g_bins_arith = 1;
bool bins_h_ok(int Q, int N) { return g_bins_arith == 1 && (long long)Q * N * 4 <= (long long)BINS_H_MAX_BYTES; }
template <typename K>
int set_smem(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
               ? 0
               : 1;
}
}  // namespace

extern "C" int sqd_bins_supported(int Q, int D) { return (Q >= 1 && Q <= 128 && D >= 1 && D <= 128) ? 1 : 0; }

extern "C" int sqd_bins_set_arith(int arith) {
    SQD_CHECK_ARG(arith == 0 || arith == 1, "sqd_bins_set_arith: 0 (fp32 matrix instruction) or 1 (two-term fp16 operands), got %d", arith);
    g_bins_arith = arith;
    return SQD_OK;
}

extern "C" int sqd_bins_workspace(int B, int Q, int D, int N, int64_t *part_floats) {
    SQD_CHECK_ARG(sqd_bins_supported(Q, D), "sqd_bins: Q=%d and D=%d must be in 1..128", Q, D);
    const BinsPlan p = plan_bins(B, Q, D, N);
    const int wg = p.wg_per_image > p.wg_bwd_h ? p.wg_per_image : p.wg_bwd_h;          // (either arithmetic may run: sqd_bins_set_arith)
    if (part_floats) *part_floats = (int64_t)wg * B * ((int64_t)D * Q + 2 * D) + (int64_t)B * 2 * D;
    return SQD_OK;
}

// energy [B,Q,N], weight [D,Q], bias [D], centers [B,D] -> pred [B,N]
extern "C" int sqd_bins_fwd(const float *energy, const float *weight, const float *bias, const float *centers, float *pred, int B,
                            int Q, int D, int N, void *stream) {
    SQD_CHECK_ARG(energy && weight && bias && centers && pred, "sqd_bins_fwd: null pointer");
    SQD_CHECK_ARG(B > 0 && N > 0 && (long long)N * 128 * 4 < (1ll << 32) && sqd_bins_supported(Q, D),
                  "sqd_bins_fwd: unsupported dims B=%d Q=%d D=%d N=%d", B, Q, D, N);
    const BinsPlan p = plan_bins(B, Q, D, N);
    const BinsDims dm = {B, Q, D, N};
    const dim3 grid(p.wg_per_image, B), grid_h(p.wg_fwd_h, B);
    (void)hipGetLastError();
    if (bins_h_ok(Q, N)) {
        if (p.dt == 2) {
            static int once = set_smem(bins_fwd_h_kernel<2, 2>, plan_bins(1, 64, 64, 32).smem_fwd_h);
            SQD_CHECK_ARG(once == 0, "sqd_bins_fwd: cannot reserve %zu bytes of LDS", p.smem_fwd_h);
            hipLaunchKernelGGL((bins_fwd_h_kernel<2, 2>), grid_h, dim3(256), p.smem_fwd_h, (hipStream_t)stream, energy, weight, bias, centers, pred,
                               dm, p.tiles_per_image);
        } else {
            static int once = set_smem(bins_fwd_h_kernel<4, 4>, plan_bins(1, 128, 128, 32).smem_fwd_h);
            SQD_CHECK_ARG(once == 0, "sqd_bins_fwd: cannot reserve %zu bytes of LDS", p.smem_fwd_h);
            hipLaunchKernelGGL((bins_fwd_h_kernel<4, 4>), grid_h, dim3(256), p.smem_fwd_h, (hipStream_t)stream, energy, weight, bias, centers, pred,
                               dm, p.tiles_per_image);
        }
    } else if (p.dt == 2) {
        hipLaunchKernelGGL((bins_fwd_kernel<2, 2>), grid, dim3(256), p.smem_fwd, (hipStream_t)stream, energy, weight, bias, centers, pred,
                           dm, p.tiles_per_image);
    } else {
        static int once = set_smem(bins_fwd_kernel<4, 4>, plan_bins(1, 128, 128, 32).smem_fwd);
        SQD_CHECK_ARG(once == 0, "sqd_bins_fwd: cannot reserve %zu bytes of LDS", p.smem_fwd);
        hipLaunchKernelGGL((bins_fwd_kernel<4, 4>), grid, dim3(256), p.smem_fwd, (hipStream_t)stream, energy, weight, bias, centers, pred,
                           dm, p.tiles_per_image);
    }
    SQD_CHECK_LAUNCH("sqd_bins_fwd");
    return SQD_OK;
}

// g_pred [B,N] -> g_energy [B,Q,N], g_weight [D,Q], g_bias [D], g_centers [B,D]; part: sqd_bins_workspace floats
extern "C" int sqd_bins_bwd(const float *energy, const float *weight, const float *bias, const float *centers, const float *g_pred,
                            float *g_energy, float *g_weight, float *g_bias, float *g_centers, float *part, int B, int Q, int D,
                            int N, void *stream) {
    SQD_CHECK_ARG(energy && weight && bias && centers && g_pred && g_energy && g_weight && g_bias && g_centers && part,
                  "sqd_bins_bwd: null pointer");
    SQD_CHECK_ARG(B > 0 && N > 0 && (long long)N * 128 * 4 < (1ll << 32) && sqd_bins_supported(Q, D),
                  "sqd_bins_bwd: unsupported dims B=%d Q=%d D=%d N=%d", B, Q, D, N);
    const BinsPlan p = plan_bins(B, Q, D, N);
    const BinsDims dm = {B, Q, D, N};
    const int wgi = bins_h_ok(Q, N) ? p.wg_bwd_h : p.wg_per_image;
    const dim3 grid(wgi, B);
    float *part_w = part, *part_v = part + (size_t)wgi * B * D * Q;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    if (bins_h_ok(Q, N)) {
        if (p.dt == 2) {
            static int once = set_smem(bins_bwd_h_kernel<2, 2>, plan_bins(1, 64, 64, 32).smem_bwd_h);
            SQD_CHECK_ARG(once == 0, "sqd_bins_bwd: cannot reserve %zu bytes of LDS", p.smem_bwd_h);
            hipLaunchKernelGGL((bins_bwd_h_kernel<2, 2>), grid, dim3(256), p.smem_bwd_h, st, energy, weight, bias, centers, g_pred, g_energy,
                               part_w, part_v, dm, p.tiles_per_image);
        } else {
            static int once = set_smem(bins_bwd_h_kernel<4, 4>, plan_bins(1, 128, 128, 32).smem_bwd_h);
            SQD_CHECK_ARG(once == 0, "sqd_bins_bwd: cannot reserve %zu bytes of LDS", p.smem_bwd_h);
            hipLaunchKernelGGL((bins_bwd_h_kernel<4, 4>), grid, dim3(256), p.smem_bwd_h, st, energy, weight, bias, centers, g_pred, g_energy,
                               part_w, part_v, dm, p.tiles_per_image);
        }
    } else if (p.dt == 2) {
        hipLaunchKernelGGL((bins_bwd_kernel<2, 2>), grid, dim3(256), p.smem_bwd, st, energy, weight, bias, centers, g_pred, g_energy,
                           part_w, part_v, dm, p.tiles_per_image);
    } else {
        static int once = set_smem(bins_bwd_kernel<4, 4>, plan_bins(1, 128, 128, 32).smem_bwd);
        SQD_CHECK_ARG(once == 0, "sqd_bins_bwd: cannot reserve %zu bytes of LDS", p.smem_bwd);
        hipLaunchKernelGGL((bins_bwd_kernel<4, 4>), grid, dim3(256), p.smem_bwd, st, energy, weight, bias, centers, g_pred, g_energy,
                           part_w, part_v, dm, p.tiles_per_image);
    }
    const int nbw = (D * Q + 31) / 32;
    hipLaunchKernelGGL(bins_tail_kernel, dim3(nbw + (2 * D + 31) / 32), dim3(256), 0, st, part_w, g_weight, D * Q, wgi * B, nbw, part_v, g_bias,
                       g_centers, B, D, wgi);
    SQD_CHECK_LAUNCH("sqd_bins_bwd");
    return SQD_OK;
}

// ---------------------------------------------------------------------------------------------------
// bin centres from the regressor's raw outputs (norm == "linear"; reference networks/depth_decoder_QTR.py:56-66):
//   v = relu(y) + 0.1;  w = v / sum(v);  widths = (max - min) * w;  edges = cumsum([min, widths]);  centers = midpoints
// [B, D] numbers, D <= 128: one launch instead of ~8 element-wise / scan launches forward and ~12 backward.  One wavefront
// per row; the two scans run serially on lane 0 (left to right, the order of torch.cumsum).
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void bin_centers_fwd_kernel(const float *__restrict__ y, float *__restrict__ centers,
                                                             float *__restrict__ sums, int D, float vmin, float vmax) {
    __shared__ float v[128], c[129];
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int d = lane; d < D; d += 64) v[d] = fmaxf(y[(size_t)b * D + d], 0.f) + 0.1f;
    __syncthreads();
    if (lane == 0) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += v[d];
        sums[b] = s;
        float e = vmin;                                   // edges[0] = min, edges[d+1] = edges[d] + width[d]
        c[0] = e;
        for (int d = 0; d < D; ++d) {
            e += (vmax - vmin) * (v[d] / s);
            c[d + 1] = e;
        }
    }
    __syncthreads();
    for (int d = lane; d < D; d += 64) centers[(size_t)b * D + d] = 0.5f * (c[d] + c[d + 1]);
}

__global__ __launch_bounds__(64) void bin_centers_bwd_kernel(const float *__restrict__ y, const float *__restrict__ sums,
                                                             const float *__restrict__ g_centers, float *__restrict__ g_y, int D,
                                                             float vmin, float vmax) {
    __shared__ float v[128], gw[128];
    __shared__ float dot;
    const int b = blockIdx.x, lane = threadIdx.x;
    const float s = sums[b], R = vmax - vmin;
    for (int d = lane; d < D; d += 64) {
        v[d] = fmaxf(y[(size_t)b * D + d], 0.f) + 0.1f;
        // d centers / d edges: edge k (k = 1..D) is shared by the centres k-1 and k
        const float gc0 = g_centers[(size_t)b * D + d], gc1 = d + 1 < D ? g_centers[(size_t)b * D + d + 1] : 0.f;
        gw[d] = 0.5f * (gc0 + gc1);                        // gradient of edge d+1, before the reverse scan
    }
    __syncthreads();
    if (lane == 0) {
        float acc = 0.f, dt = 0.f;
        for (int d = D - 1; d >= 0; --d) {                // width d feeds every edge after it
            acc += gw[d];
            gw[d] = acc;
            dt += acc * v[d];
        }
        dot = dt;
    }
    __syncthreads();
    for (int d = lane; d < D; d += 64) {
        const float gv = R * (gw[d] - dot / s) / s;       // w = v / s
        g_y[(size_t)b * D + d] = y[(size_t)b * D + d] > 0.f ? gv : 0.f;
    }
}
}  // namespace

// y [B,D] raw regressor outputs -> centers [B,D] (bin mid-points between vmin and vmax), sums [B] (for the backward); D <= 128
extern "C" int sqd_bin_centers_fwd(const float *y, float *centers, float *sums, int B, int D, float vmin, float vmax, void *stream) {
    SQD_CHECK_ARG(y && centers && sums && B >= 1 && D >= 1 && D <= 128, "sqd_bin_centers_fwd: bad arguments (B=%d, D=%d)", B, D);
    (void)hipGetLastError();
    hipLaunchKernelGGL(bin_centers_fwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, y, centers, sums, D, vmin, vmax);
    SQD_CHECK_LAUNCH("sqd_bin_centers_fwd");
    return SQD_OK;
}
extern "C" int sqd_bin_centers_bwd(const float *y, const float *sums, const float *g_centers, float *g_y, int B, int D, float vmin,
                                   float vmax, void *stream) {
    SQD_CHECK_ARG(y && sums && g_centers && g_y && B >= 1 && D >= 1 && D <= 128, "sqd_bin_centers_bwd: bad arguments (B=%d, D=%d)", B, D);
    (void)hipGetLastError();
    hipLaunchKernelGGL(bin_centers_bwd_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, y, sums, g_centers, g_y, D, vmin, vmax);
    SQD_CHECK_LAUNCH("sqd_bin_centers_bwd");
    return SQD_OK;
}
