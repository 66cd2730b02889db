/* ORACLE (test infrastructure — never linked into or called by the product path).
 *
 * Plain-C restatement of the per-pixel geometry chain of the reference, with the floating-point
 * operation order written out explicitly so that integer sampling taps are reproducible bit for bit:
 *
 *   F.interpolate(disp,[H,W],bilinear,align_corners=False)      reference trainer.py:395-399
 *   BackprojectDepth.forward                                     reference layers.py:210-215
 *   Project3D.forward                                            reference layers.py:247-258
 *   F.grid_sample(..., padding_mode="border", align_corners=True) reference trainer.py:431-435
 *
 * Canonical arithmetic (SURVEY.md §7 "Bit-exact indices", probed on torch-CPU/MKL): the big-N
 * products inv_K[3x3]@pix and P[3x4]@pts are FMA chains in k order (acc = a0*b0; acc = fma(a_k,b_k,acc));
 * every other operation is a single correctly-rounded fp32 op; true divisions stay divisions.
 * P = (K@T)[:, :3] is an INPUT (12 floats per image and source) so its own rounding never matters.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC warp_chain.c -o _build/libsqd_oracle.so -lm
 * (see oracle/Makefile).  Pinned against the imported reference by tests/test_oracle_c_chain.py
 * through the golden groups G2/G3/G4/G7 (indices equal outside the recorded fragile mask).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* ---- bilinear resize, align_corners=False (ATen area_pixel_compute_source_index) ---------------- */
void sqo_depth_up(const float *disp, float *depth, int B, int h, int w, int H, int W)
{
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            float fy = sy * ((float)y + 0.5f) - 0.5f;
            if (fy < 0.f) fy = 0.f;
            int y0 = (int)fy;
            int y1 = y0 + (y0 < h - 1 ? 1 : 0);
            float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
            for (int x = 0; x < W; ++x) {
                float fx = sx * ((float)x + 0.5f) - 0.5f;
                if (fx < 0.f) fx = 0.f;
                int x0 = (int)fx;
                int x1 = x0 + (x0 < w - 1 ? 1 : 0);
                float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
                const float *p = disp + (size_t)b * h * w;
                /* Four pre-multiplied weights, accumulated v01 first then v00, v10, v11 as an FMA chain:
                 * probed bit-identical to F.interpolate(bilinear, align_corners=False) on torch 2.10 CPU
                 * for every golden pixel (tests/test_oracle_c_chain.py). */
                float w00 = ly0 * lx0, w01 = ly0 * lx1, w10 = ly1 * lx0, w11 = ly1 * lx1;
                float acc = w01 * p[y0 * w + x1];
                acc = fmaf(w00, p[y0 * w + x0], acc);
                acc = fmaf(w10, p[y1 * w + x0], acc);
                acc = fmaf(w11, p[y1 * w + x1], acc);
                depth[((size_t)b * H + y) * W + x] = acc;
            }
        }
}

/* ---- backproject -> project -> normalised grid -> taps -> bilinear border sample ---------------
 * depth [B,H,W]; inv_K [B,4,4]; P [B,3,4] (one source); src [B,C,H,W]
 * outputs: grid [B,H,W,2]; x0,y0 int32 [B,H,W]; warped [B,C,H,W] (may be NULL to skip sampling) */
void sqo_warp(const float *depth, const float *inv_K, const float *P, const float *src,
              float *grid, int32_t *x0o, int32_t *y0o, float *warped, int B, int C, int H, int W)
{
    const float eps = 1e-7f;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    for (int b = 0; b < B; ++b) {
        const float *ik = inv_K + b * 16, *p = P + b * 12;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t q = ((size_t)b * H + y) * W + x;
                float fx = (float)x, fy = (float)y, d = depth[q];
                float c[3], X[3], cam[3];
                for (int i = 0; i < 3; ++i) {           /* layers.py:211, FMA chain k = 0..2 */
                    float acc = ik[i * 4 + 0] * fx;
                    acc = fmaf(ik[i * 4 + 1], fy, acc);
                    acc = fmaf(ik[i * 4 + 2], 1.0f, acc);
                    c[i] = acc;
                    X[i] = d * c[i];                    /* layers.py:212 */
                }
                for (int i = 0; i < 3; ++i) {           /* layers.py:250, FMA chain k = 0..3 */
                    float acc = p[i * 4 + 0] * X[0];
                    acc = fmaf(p[i * 4 + 1], X[1], acc);
                    acc = fmaf(p[i * 4 + 2], X[2], acc);
                    acc = fmaf(p[i * 4 + 3], 1.0f, acc);
                    cam[i] = acc;
                }
                float z = cam[2] + eps;                 /* layers.py:252 */
                float u = cam[0] / z, v = cam[1] / z;
                u = u / wm1;                            /* :255 */
                v = v / hm1;                            /* :256 */
                float gx = (u - 0.5f) * 2.0f;           /* :257 */
                float gy = (v - 0.5f) * 2.0f;
                grid[q * 2 + 0] = gx;
                grid[q * 2 + 1] = gy;
                /* grid_sampler_unnormalize(align_corners=True) + clip_coordinates (border) */
                float ix = ((gx + 1.0f) / 2.0f) * wm1;
                float iy = ((gy + 1.0f) / 2.0f) * hm1;
                ix = fminf(wm1, fmaxf(ix, 0.f));
                iy = fminf(hm1, fmaxf(iy, 0.f));
                float fx0 = floorf(ix), fy0 = floorf(iy);
                int xi = (int)fx0, yi = (int)fy0;
                if (x0o) x0o[q] = xi;
                if (y0o) y0o[q] = yi;
                if (!warped) continue;
                float fx1 = fx0 + 1.f, fy1 = fy0 + 1.f;
                float nw = (fx1 - ix) * (fy1 - iy), ne = (ix - fx0) * (fy1 - iy);
                float sw = (fx1 - ix) * (iy - fy0), se = (ix - fx0) * (iy - fy0);
                int xin = xi + 1 < W, yin = yi + 1 < H;
                for (int ch = 0; ch < C; ++ch) {
                    const float *s = src + ((size_t)b * C + ch) * H * W;
                    float acc = 0.f;
                    acc = fmaf(s[yi * W + xi], nw, acc);
                    if (xin) acc = fmaf(s[yi * W + xi + 1], ne, acc);
                    if (yin) acc = fmaf(s[(yi + 1) * W + xi], sw, acc);
                    if (xin && yin) acc = fmaf(s[(yi + 1) * W + xi + 1], se, acc);
                    warped[((size_t)b * C + ch) * H * W + (size_t)y * W + x] = acc;
                }
            }
    }
}
