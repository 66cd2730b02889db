// photo_fwd_pk.hip — the fused warp+SSIM forward, packed-math edition (S = 2 source frames).
//
// Same algorithm, data flow and results as the scalar kernel it replaces (photometric.hip documents
// the column march); what changes is the instruction stream, because the kernel is VALU-issue-bound
// on gfx950 (PMC: 30 M VALU instructions per launch, ~4 cycles each, profiles/r01b_*):
//   * the two sources travel together as float2 -> v_pk_fma/mul/add_f32 for the projection chains,
//     the bilinear blend, the vertical window sums, the SSIM algebra and its gradient;
//   * true divisions keep their bits but lose their overhead: the arithmetic core of the compiler's
//     own IEEE expansion (v_rcp + one Newton step + two residual corrections) is written out, the
//     range scaling / fix-up wrappers (irrelevant for pixel-range operands) are dropped, and the
//     reciprocal of the shared divisor z, of (W-1) and of (H-1) is computed once;
//   * sigma_x + sigma_y only needs sum(w^2 + t^2): 21 instead of 24 quantities go through box7;
//   * the DPP chains of three quantities are interleaved so no hazard nop is needed;
//   * 32-bit unsigned offsets off uniform base pointers (saddr addressing, no 64-bit VALU math).
#include "sqd_common.h"

namespace {
using namespace sqd;
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
constexpr float INV49 = 1.0f / 49.0f;
constexpr int OWN0 = 3, OWN1 = 61;

// 32-bit unsigned BYTE offsets off a wave-uniform base -> global_load/store with saddr + voffset
// (no 64-bit address arithmetic on the VALU); every tensor here is < 4 GiB (checked on the host)
__device__ __forceinline__ float ldg(const float *__restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void stg(float *__restrict__ base, unsigned byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
__device__ __forceinline__ v2f splat(float x) { return v2f{x, x}; }
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// ---- correctly rounded division, core of the IEEE expansion (valid for normal-range operands) -----
__device__ __forceinline__ float rcp_refined(float b) {
    const float r = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r, 1.0f), r, r);
}
__device__ __forceinline__ v2f rcp_refined2(v2f b) {
    const v2f r = v2f{__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    return pfma(pfma(-b, r, splat(1.0f)), r, r);
}
__device__ __forceinline__ v2f div_core2(v2f a, v2f b, v2f r) {      // a / b with r = rcp_refined(b)
    v2f q = a * r;
    v2f e = pfma(-b, q, a);
    q = pfma(e, r, q);
    e = pfma(-b, q, a);
    return pfma(e, r, q);
}

// three interleaved centred 7-tap box sums across lanes (see sqd::box7)
__device__ __forceinline__ void box7x3(float &a, float &b, float &c) {
    float ra = a + wave_shr1(a), rb = b + wave_shr1(b), rc = c + wave_shr1(c);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc);
    float ua = a + wave_shl1(a), ub = b + wave_shl1(b), uc = c + wave_shl1(c);
    ua = a + wave_shl1(ua); ub = b + wave_shl1(ub); uc = c + wave_shl1(uc);
    a = ra + wave_shl1(ua); b = rb + wave_shl1(ub); c = rc + wave_shl1(uc);
}
__device__ __forceinline__ void box7x3(v2f &a, v2f &b, v2f &c) {
    float ax = a.x, bx = b.x, cx = c.x, ay = a.y, by = b.y, cy = c.y;
    box7x3(ax, bx, cx);
    box7x3(ay, by, cy);
    a = v2f{ax, ay};
    b = v2f{bx, by};
    c = v2f{cx, cy};
}

struct StripPk {
    int b, y_begin, cx, xr;
    bool own_col;
};

// (forcing 4 waves/SIMD through __launch_bounds__ spills 62 VGPRs into the row loop: 174 us vs 100 us)
template <int MODE>
__global__ __launch_bounds__(256) void photo_fwd_pk_kernel(sqd_photo_args a, const float *__restrict__ noise, int TH,
                                                           int nsx, int nsy, int ntasks) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (task >= ntasks) return;
    const int H = a.H, W = a.W;
    const unsigned HW = (unsigned)(H * W);
    StripPk st;
    {
        const int sx = task % nsx, t2 = task / nsx, sy = t2 % nsy;
        st.b = t2 / nsy;
        st.y_begin = sy * TH;
        st.cx = sx * SQD_STRIP_COLS - 3 + lane;
        st.xr = reflect_idx(st.cx, W);
        st.own_col = lane >= OWN0 && lane < OWN1 && st.cx >= 0 && st.cx < W;
    }
    const int b = st.b;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const bool want_grad = MODE == 1 && a.coef != nullptr;

    const float *__restrict__ tgt = a.target + (size_t)b * 3 * HW;
    const float *__restrict__ dep = MODE ? a.depth + (size_t)b * HW : nullptr;
    const float *__restrict__ src0 = a.sources[0] + (size_t)b * 3 * HW;
    const float *__restrict__ src1 = a.sources[1] + (size_t)b * 3 * HW;
    float ik[9];
    v2f P[12];                       // (source 0, source 1) projection matrices
    v2f rW, rH;
    if (MODE) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) ik[i * 3 + k] = a.inv_K[(size_t)b * 16 + i * 4 + k];
#pragma unroll
        for (int k = 0; k < 12; ++k) P[k] = v2f{a.P[((size_t)b * 2 + 0) * 12 + k], a.P[((size_t)b * 2 + 1) * 12 + k]};
        rW = splat(rcp_refined(wm1));
        rH = splat(rcp_refined(hm1));
    }

    float rt[7][3];                  // ring: target rgb
    v2f rw[7][3];                    // ring: (pred_0, pred_1) rgb
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rt[k][c] = 0.f;
            rw[k][c] = splat(0.f);
        }
    float loss_acc = 0.f;
    const int nrows = TH + 6;

#pragma nounroll
    for (int j = 0; j < nrows; ++j) {
        const int ycell = st.y_begin - 3 + j;
        const int yr = reflect_idx(ycell, H);
        const unsigned off = (unsigned)(yr * W + st.xr);
        const bool own_cell = st.own_col && j >= 3 && j < TH + 3 && ycell < H;
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rt[k][c] = rt[k + 1][c];
                rw[k][c] = rw[k + 1][c];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) rt[6][c] = ldg(tgt, (off + c * HW) * 4u);
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rw[6][c] = v2f{ldg(src0, (off + c * HW) * 4u), ldg(src1, (off + c * HW) * 4u)};
        } else {
            // ---- camera ray and point (shared by both sources) — layers.py:211-212
            const float fx = (float)st.xr, fy = (float)yr;
            float X[3];
            const float d = ldg(dep, off * 4u);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float acc = ik[i * 3 + 0] * fx;
                acc = fmaf(ik[i * 3 + 1], fy, acc);
                acc = fmaf(ik[i * 3 + 2], 1.0f, acc);
                X[i] = d * acc;
            }
            // ---- projection of both sources at once — layers.py:250-257 (FMA chain k = 0..3)
            v2f cam[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                v2f acc = P[i * 4 + 0] * splat(X[0]);
                acc = pfma(P[i * 4 + 1], splat(X[1]), acc);
                acc = pfma(P[i * 4 + 2], splat(X[2]), acc);
                acc = pfma(P[i * 4 + 3], splat(1.0f), acc);
                cam[i] = acc;
            }
            const v2f z = cam[2] + splat(1e-7f);
            const v2f rz = rcp_refined2(z);
            const v2f u = div_core2(cam[0], z, rz), v = div_core2(cam[1], z, rz);
            const v2f un = div_core2(u, splat(wm1), rW), vn = div_core2(v, splat(hm1), rH);
            const v2f gx = (un - splat(0.5f)) * splat(2.0f), gy = (vn - splat(0.5f)) * splat(2.0f);
            v2f ix = ((gx + splat(1.0f)) * splat(0.5f)) * splat(wm1);
            v2f iy = ((gy + splat(1.0f)) * splat(0.5f)) * splat(hm1);
            ix = v2f{fminf(wm1, fmaxf(ix.x, 0.f)), fminf(wm1, fmaxf(ix.y, 0.f))};
            iy = v2f{fminf(hm1, fmaxf(iy.x, 0.f)), fminf(hm1, fmaxf(iy.y, 0.f))};
            const v2f fx0 = v2f{floorf(ix.x), floorf(ix.y)}, fy0 = v2f{floorf(iy.x), floorf(iy.y)};
            const v2f ax = ix - fx0, ay = iy - fy0;
            const v2f bx = (fx0 + splat(1.f)) - ix, by = (fy0 + splat(1.f)) - iy;
            const int x00 = (int)fx0.x, y00 = (int)fy0.x, x01 = (int)fx0.y, y01 = (int)fy0.y;
            const bool xin0 = x00 + 1 < W, yin0 = y00 + 1 < H, xin1 = x01 + 1 < W, yin1 = y01 + 1 < H;
            v2f wnw = bx * by, wne = ax * by, wsw = bx * ay, wse = ax * ay;
            // out-of-range taps are skipped by grid_sample: weight exactly 0 and a clamped (in-range) address
            wne = v2f{xin0 ? wne.x : 0.f, xin1 ? wne.y : 0.f};
            wsw = v2f{yin0 ? wsw.x : 0.f, yin1 ? wsw.y : 0.f};
            wse = v2f{(xin0 && yin0) ? wse.x : 0.f, (xin1 && yin1) ? wse.y : 0.f};
            const unsigned a00 = (unsigned)(y00 * W + x00), b00 = (unsigned)(y01 * W + x01);
            const unsigned a01 = a00 + (xin0 ? 1u : 0u), a10 = a00 + (yin0 ? (unsigned)W : 0u), a11 = a10 + (xin0 ? 1u : 0u);
            const unsigned b01 = b00 + (xin1 ? 1u : 0u), b10 = b00 + (yin1 ? (unsigned)W : 0u), b11 = b10 + (xin1 ? 1u : 0u);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const unsigned co = c * HW;
                v2f acc = v2f{ldg(src0, (a00 + co) * 4u), ldg(src1, (b00 + co) * 4u)} * wnw;
                acc = pfma(v2f{ldg(src0, (a01 + co) * 4u), ldg(src1, (b01 + co) * 4u)}, wne, acc);
                acc = pfma(v2f{ldg(src0, (a10 + co) * 4u), ldg(src1, (b10 + co) * 4u)}, wsw, acc);
                acc = pfma(v2f{ldg(src0, (a11 + co) * 4u), ldg(src1, (b11 + co) * 4u)}, wse, acc);
                rw[6][c] = acc;
            }
            if (own_cell) {
                const size_t q = (size_t)b * HW + off;
                if (a.sample[0]) *reinterpret_cast<float2 *>(a.sample[0] + q * 2) = make_float2(gx.x, gy.x);
                if (a.sample[1]) *reinterpret_cast<float2 *>(a.sample[1] + q * 2) = make_float2(gx.y, gy.y);
                if (a.x0y0[0]) *reinterpret_cast<int2 *>(a.x0y0[0] + q * 2) = make_int2(x00, y00);
                if (a.x0y0[1]) *reinterpret_cast<int2 *>(a.x0y0[1] + q * 2) = make_int2(x01, y01);
                if (a.warped[0]) {
                    float *w0 = a.warped[0] + (size_t)b * 3 * HW, *w1 = a.warped[1] + (size_t)b * 3 * HW;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        stg(w0, (off + c * HW) * 4u, rw[6][c].x);
                        stg(w1, (off + c * HW) * 4u, rw[6][c].y);
                    }
                }
            }
        }

        if (j >= 6) {
            const int yo = ycell - 3;
            v2f ssim_sum = splat(0.f), l1 = splat(0.f);
            float gs0[9], gs1[9];      // d loss / d window sums of source 0 / source 1
            float Stv[3];
            v2f Swv[3], Sqv[3], Swtv[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // ---- vertical sums over the 7 ring rows
                float St = rt[0][c], Stt = rt[0][c] * rt[0][c];
                v2f Sw = rw[0][c], Sq = rw[0][c] * rw[0][c], Swt = rw[0][c] * splat(rt[0][c]);
#pragma unroll
                for (int k = 1; k < 7; ++k) {
                    const float t = rt[k][c];
                    const v2f w = rw[k][c];
                    St += t;
                    Stt = fmaf(t, t, Stt);
                    Sw += w;
                    Sq = pfma(w, w, Sq);
                    Swt = pfma(w, splat(t), Swt);
                }
                Stv[c] = St;
                Swv[c] = Sw;
                Sqv[c] = Sq + splat(Stt);               // sigma_x + sigma_y only needs sum(w^2 + t^2)
                Swtv[c] = Swt;
            }
            // ---- horizontal sums (DPP), three independent chains at a time
            box7x3(Stv[0], Stv[1], Stv[2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) box7x3(Swv[c], Sqv[c], Swtv[c]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const v2f Sw = Swv[c], Sq = Sqv[c], Swt = Swtv[c];
                // ---- SSIM of (pred_s, target) for both s — layers.py:35-46
                const float mt = Stv[c] * INV49;
                const v2f mw = Sw * splat(INV49);
                const v2f mwmt = mw * splat(mt);
                const v2f sxy = Swt * splat(INV49) - mwmt;
                const v2f mm = pfma(mw, mw, splat(mt * mt));
                const v2f A1 = pfma(splat(2.f), mwmt, splat(C1)), A2 = pfma(splat(2.f), sxy, splat(C2));
                const v2f B1 = mm + splat(C1), B2 = (Sq * splat(INV49) - mm) + splat(C2);
                const v2f iB1 = v2f{__builtin_amdgcn_rcpf(B1.x), __builtin_amdgcn_rcpf(B1.y)};
                const v2f iB2 = v2f{__builtin_amdgcn_rcpf(B2.x), __builtin_amdgcn_rcpf(B2.y)};
                const v2f iB = iB1 * iB2;                 // (reciprocals: gradient coefficients only)
                const v2f Bd = B1 * B2;
                const v2f Sv = div_core2(A1 * A2, Bd, rcp_refined2(Bd));      // SSIM_n / SSIM_d as the reference divides (layers.py:46)
                const v2f r = (splat(1.f) - Sv) * splat(0.5f);
                ssim_sum += v2f{fminf(fmaxf(r.x, 0.f), 1.f), fminf(fmaxf(r.y, 0.f), 1.f)};
                const v2f df = splat(rt[3][c]) - rw[3][c];
                l1 += v2f{fabsf(df.x), fabsf(df.y)};
                if (want_grad) {
                    const v2f k = v2f{(r.x >= 0.f && r.x <= 1.f) ? -0.5f * INV49 : 0.f, (r.y >= 0.f && r.y <= 1.f) ? -0.5f * INV49 : 0.f};
                    const v2f dmu = splat(2.f) * (splat(mt) * (A2 - A1) * iB + mw * Sv * (iB2 - iB1));
                    const v2f g0 = k * dmu, g1 = k * (-Sv * iB2), g2 = k * (splat(2.f) * A1 * iB);
                    gs0[c] = g0.x; gs1[c] = g0.y;
                    gs0[3 + c] = g1.x; gs1[3 + c] = g1.y;
                    gs0[6 + c] = g2.x; gs1[6 + c] = g2.y;
                }
            }
            const v2f lossv = splat(0.85f) * (ssim_sum * splat(1.f / 3.f)) + splat(0.15f) * (l1 * splat(1.f / 3.f));
            const bool own_out = st.own_col && yo >= 0 && yo < H && yo < st.y_begin + TH;
            if (own_out) {
                const unsigned qo = (unsigned)(yo * W + st.cx);
                if (MODE == 0) {
                    float *o = a.sel + (size_t)b * 2 * HW;
                    const float *nz = noise ? noise + (size_t)b * 2 * HW : nullptr;
                    stg(o, qo * 4u, lossv.x + (nz ? ldg(nz, qo * 4u) : 0.f) * 0.00001f);            // trainer.py:514-517
                    stg(o, (qo + HW) * 4u, lossv.y + (nz ? ldg(nz, (qo + HW) * 4u) : 0.f) * 0.00001f);
                } else {
                    const float *idm = a.identity + (size_t)b * 2 * HW;
                    float best = ldg(idm, qo * 4u);
                    int bi = 0;
                    const float v1 = ldg(idm, (qo + HW) * 4u);
                    if (v1 < best) { best = v1; bi = 1; }
                    if (lossv.x < best) { best = lossv.x; bi = 2; }
                    if (lossv.y < best) { best = lossv.y; bi = 3; }
                    if (a.reproj) {
                        a.reproj[(size_t)b * 2 * HW + qo] = lossv.x;
                        a.reproj[(size_t)b * 2 * HW + HW + qo] = lossv.y;
                    }
                    loss_acc += best;
                    if (a.sel) a.sel[(size_t)b * HW + qo] = bi > 1 ? 1.f : 0.f;
                    if (a.idx) a.idx[(size_t)b * HW + qo] = (uint8_t)bi;
                    if (want_grad && bi >= 2) {
                        float *co = a.coef + (size_t)b * 9 * HW;
#pragma unroll
                        for (int k = 0; k < 9; ++k) stg(co, (qo + k * HW) * 4u, (bi == 2 ? gs0[k] : gs1[k]) * (0.85f / 3.f));
                    }
                }
            }
        }
    }
    if (MODE == 1 && a.loss_part) {
        loss_acc = wave_sum(loss_acc);
        if (lane == 0) a.loss_part[task] = loss_acc;
    }
}
}  // namespace

namespace sqd {
void launch_photo_fwd_pk(const sqd_photo_args &a, const float *noise, int mode, int TH, int nsx, int nsy, int ntasks,
                         hipStream_t stream) {
    if (mode == 0)
        hipLaunchKernelGGL((photo_fwd_pk_kernel<0>), dim3((ntasks + 3) / 4), dim3(256), 0, stream, a, noise, TH, nsx, nsy, ntasks);
    else
        hipLaunchKernelGGL((photo_fwd_pk_kernel<1>), dim3((ntasks + 3) / 4), dim3(256), 0, stream, a, noise, TH, nsx, nsy, ntasks);
}
}  // namespace sqd
