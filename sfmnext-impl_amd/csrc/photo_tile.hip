// photo_tile.hip — the fused warp + SSIM forward of the photometric chain (source frames in pairs), tile edition.
//
// replaces (reference): BackprojectDepth -> Project3D -> grid_sample (layers.py:186-258, trainer.py:420-435), SSIM + L1
// (layers.py:13-46, trainer.py:441-453) and the per-pixel minimum / auto-mask (trainer.py:474-532) in ONE launch.
//
// Execution shape.  A workgroup of 8 wavefronts (4 on images of fewer than 56 rows) owns a tile of 58..61 columns x 24..28 rows of
// the target image (make_tiling: a balanced cut — 1024 tiles = 4 per CU at configs[1]).
//   phase 1 (per cell, once): every cell of the tile + its 3-pixel halo — 64 columns (one per lane) x (rows + 6), rows dealt
//     round-robin to the waves — is back-projected, projected into both source views and bilinearly sampled (canonical fp32 order of
//     oracle/warp_chain.c: bit-exact integer taps); the six warped colours go to LDS (8-byte (source 0, source 1) pairs,
//     lane-contiguous: conflict-free); grid and colours of the cells the tile owns go to HBM.
//   phase 2 (per output pixel): each wave takes pairs of output rows; the 7x7 window statistics are vertical register sums over LDS
//     rows (two outputs share six of their seven rows) followed by the horizontal 7-tap sum on wavefront shuffles (six
//     v_add_f32_dpp per quantity: no LDS traffic, no barrier), then the SSIM algebra, L1, the minimum against the identity maps,
//     identity_selection, the argmin byte and the loss partial.  ReflectionPad2d(3): rows are reflected when they are fetched
//     from LDS; at the left / right image border the wavefront holds the border columns itself and adds the mirrored terms from
//     the shuffle chain's own intermediates, so border strips own 61 columns instead of 58 (640 columns = 11 strips, not 12).
// Kernels of the forward (sqd_photo_set_fwd_variant; all of them write the same bits):
//   photo_tile_kernel<1, 8, true, true>  the LEAN edition — what a training step of the default loss options runs (round 6): target
//                                        rows staged into LDS by 16-byte loads whose round trip runs under the warps, warped colours
//                                        stored from LDS 16 bytes per lane between the row pairs, option-free selection, reflection-
//                                        free row addressing for tiles inside the image, hand-scheduled shuffle blocks (box7x7);
//   photo_tile_kernel<1, 8> / <1, 4>     round 5's kernel: every option set, tap dumps, images narrower than 64 columns;
//   photo_fwd_c_kernel, photo_tile_kernel<1, 8, true, false>, photo_fwd_s_kernel, photo_tile_resident_kernel
//                                        round 6's measured experiments (colour-serial phase 2 at 6 waves per SIMD, wide accesses
//                                        alone, dynamic wave roles with a row in flight under every SSIM step, resident workgroups):
//                                        parity-green and no faster — DESIGN.md 3.1 has the per-wave traces that say why.
// No coefficient planes are written for the backward any more: photo_coef (MODE 2 below) recomputes them from the warped
// images when — and only when — a backward pass runs.
//
// Roofline: HBM.  Algorithmic bytes per target pixel: reads depth-lowres 1 + target 12 + sources 24 +
// identity 8, writes depth 4 (by depth_up) + sample 16 + warped 24 + identity_selection 4 = 93 B.
#include "photo_launch.h"

namespace {
using namespace sqd;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));      // (register triples as SSA vectors: float[3] members end up as private arrays)

constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
constexpr float INV49 = 1.0f / 49.0f;
#ifdef SQD_PHOTO_TRACE      // developer build (tools/build_variant.sh trace photo_tile.hip -DSQD_PHOTO_TRACE): per-wave s_memtime stamps of the forward's phases
__device__ unsigned long long *g_photo_trace;      // [tile][wave][8]
#define PHOTO_STAMP(i)                                                                                              \
    do {                                                                                                            \
        if (g_photo_trace && (threadIdx.x & 63) == 0)                                                               \
            g_photo_trace[((size_t)tile * NW + wave) * 8 + (i)] = (i) == 7 ? (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) : __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PHOTO_STAMP(i) do { } while (0)
#endif
constexpr int TR_MAX = 16;               // rows a tile owns (even)
constexpr int NROW_MAX = TR_MAX + 6;     // + 3 halo rows above and below
constexpr int LDS_BYTES = NROW_MAX * 3 * 64 * 8;

__device__ __forceinline__ float ldg(const float *__restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
typedef float v2f_ua __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float __attribute__((ext_vector_type(2))) ldg2(const float *__restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const v2f_ua *>(reinterpret_cast<const char *>(base) + byte_off);      // 8 bytes, 4-byte aligned
}
typedef float v4f_ua __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float __attribute__((ext_vector_type(4))) ldg4(const float *__restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const v4f_ua *>(reinterpret_cast<const char *>(base) + byte_off);      // 16 bytes, 4-byte aligned
}
__device__ __forceinline__ void stg(float *__restrict__ base, unsigned byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
__device__ __forceinline__ void stg2(void *__restrict__ base, unsigned byte_off, float x, float y) {      // (32-bit byte offsets: the
    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(base) + byte_off) = make_float2(x, y);           //  scalar-base store forms)
}
__device__ __forceinline__ void stg2i(void *__restrict__ base, unsigned byte_off, int x, int y) {
    *reinterpret_cast<int2 *>(reinterpret_cast<char *>(base) + byte_off) = make_int2(x, y);
}
__device__ __forceinline__ v2f splat(float x) { return v2f{x, x}; }
__device__ __forceinline__ v2f pfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// ---- correctly rounded division: the arithmetic core of the compiler's IEEE expansion (v_rcp + Newton + two residual
// corrections), without the range-scaling wrappers that pixel-range operands never need (validated bit-exact against
// the C oracle: tests/test_gpu_photometric.py::test_warp_taps_bit_exact)
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

__device__ __forceinline__ void div_core2x2(v2f a0, v2f a1, v2f b0, v2f b1, v2f r0, v2f r1, v2f &o0, v2f &o1) {      // o_i = a_i / b_i, r_i = rcp_refined(b_i)
    v2f q0 = a0 * r0, q1 = a1 * r1;
    v2f e0 = pfma(-b0, q0, a0), e1 = pfma(-b1, q1, a1);
    q0 = pfma(e0, r0, q0); q1 = pfma(e1, r1, q1);
    e0 = pfma(-b0, q0, a0); e1 = pfma(-b1, q1, a1);
    o0 = pfma(e0, r0, q0); o1 = pfma(e1, r1, q1);
}

// ---- column strips ------------------------------------------------------------------------------------------------
// Lanes of a wavefront = 64 consecutive image columns.  An interior strip owns its middle 58 columns (3 halo lanes per
// side).  The first strip starts at column 0 and the last one ends at column W-1: they own up to 61 columns, and the
// mirrored columns of ReflectionPad2d(3) come from their own lanes (box7x3<LEFT / RIGHT>).  Images of at most 58 columns
// are one "virtual" strip: lanes = columns -3 .. 60, halo lanes loaded at the reflected coordinate.
enum { INTERIOR = 0, LEFT = 1, RIGHT = 2 };
struct StripX {
    int x0;           // column of lane 0 (may be negative)
    int own0, own1;   // owned columns [own0, own1)
    int kind;         // INTERIOR (also the virtual strip) / LEFT / RIGHT
    bool virt;        // halo lanes lie outside the image and are reflected when loaded
};
__host__ __device__ inline int strips_x(int W) { return W <= 58 ? 1 : W <= 122 ? 2 : 2 + (W - 122 + 57) / 58; }
__device__ __forceinline__ StripX strip_x(int k, int nsx, int W) {
    StripX s;
    s.virt = false;
    if (nsx == 1) {
        s.x0 = -3; s.own0 = 0; s.own1 = W; s.kind = INTERIOR; s.virt = true;
    } else if (k == 0) {
        s.x0 = 0; s.own0 = 0; s.own1 = min(61, W - 3); s.kind = LEFT;
    } else if (k == nsx - 1) {
        s.x0 = W - 64; s.own0 = nsx == 2 ? min(61, W - 3) : 61 + 58 * (k - 1); s.own1 = W; s.kind = RIGHT;
    } else {
        s.own0 = 61 + 58 * (k - 1); s.x0 = s.own0 - 3; s.own1 = s.own0 + 58; s.kind = INTERIOR;
    }
    return s;
}

// lane i receives lane i-n / i+n of its row of 16 lanes (0 beyond the row), or the mirror image of its quad
template <int N>
__device__ __forceinline__ float row_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
// quad_perm [3,2,1,0] inside ONE quad — row ROW (16 lanes), bank BANK (4 lanes) — zero everywhere else
template <int ROW, int BANK>
__device__ __forceinline__ float quad_reverse(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x1B, 1 << ROW, 1 << BANK, true));
}

// centred 7-tap sums of three quantities across lanes (interleaved chains: no DPP hazard nops).
// KIND == LEFT: lanes 0..3 are image columns 0..3; the reflected window of column 0 / 1 / 2 additionally holds
// v1+v2+v3 / v1+v2 / v1: prefix sums of the masked lanes 1..3 (two row-shift adds), mirrored inside the quad.
// KIND == RIGHT: lanes 60..63 are columns W-4..W-1, mirror image.
template <int KIND>
__device__ __forceinline__ void box7x3(float &a, float &b, float &c, bool edge) {
    float ea = 0.f, eb = 0.f, ec = 0.f;
    if (KIND == LEFT) {
        const float ma = edge ? a : 0.f, mb = edge ? b : 0.f, mc = edge ? c : 0.f;      // edge: lanes 1..3
        ea = quad_reverse<0, 0>((ma + row_shr<1>(ma)) + row_shr<2>(ma));      // (the prefix sums spill into lanes 4, 5: only
        eb = quad_reverse<0, 0>((mb + row_shr<1>(mb)) + row_shr<2>(mb));      //  the first quad is mirrored)
        ec = quad_reverse<0, 0>((mc + row_shr<1>(mc)) + row_shr<2>(mc));
    } else if (KIND == RIGHT) {
        const float ma = edge ? a : 0.f, mb = edge ? b : 0.f, mc = edge ? c : 0.f;      // edge: lanes 60..62
        ea = quad_reverse<3, 3>((ma + row_shl<1>(ma)) + row_shl<2>(ma));
        eb = quad_reverse<3, 3>((mb + row_shl<1>(mb)) + row_shl<2>(mb));
        ec = quad_reverse<3, 3>((mc + row_shl<1>(mc)) + row_shl<2>(mc));
    }
    float ra = a + wave_shr1(a), rb = b + wave_shr1(b), rc = c + wave_shr1(c);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc);
    float ua = a + wave_shl1(a), ub = b + wave_shl1(b), uc = c + wave_shl1(c);
    ua = a + wave_shl1(ua); ub = b + wave_shl1(ub); uc = c + wave_shl1(uc);
    a = ra + wave_shl1(ua); b = rb + wave_shl1(ub); c = rc + wave_shl1(uc);
    if (KIND != INTERIOR) {
        a += ea; b += eb; c += ec;
    }
}
template <int KIND>
__device__ __forceinline__ void box7x3(v2f &a, v2f &b, v2f &c, bool edge) {
    float ax = a.x, bx = b.x, cx = c.x, ay = a.y, by = b.y, cy = c.y;
    box7x3<KIND>(ax, bx, cx, edge);
    box7x3<KIND>(ay, by, cy, edge);
    a = v2f{ax, ay};
    b = v2f{bx, by};
    c = v2f{cx, cy};
}

// centred 7-tap sums of SEVEN quantities (one colour's St, Sw, Sq, Swt of both sources) as ONE block of 42 v_add_f32_dpp in a fixed
// interleaved order (box7x3's six steps, every step across the seven chains: a chain's shuffle reads a register written seven
// instructions earlier — no hazard nops; the leading s_nop covers a producer right in front of the block).  Round 6: left to the
// compiler, the last two steps of a chain came out as v_mov_b32_dpp + (SLP-paired) v_pk_add_f32 — 7 instructions per quantity and 21
// hazard nops per output row instead of 6 and none.  Same sums in the same order as box7x3.
#define SQD_DPP_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define SQD_DPP_SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
template <int KIND>
__device__ __forceinline__ void box7x7(float (&v)[7], bool edge) {
    float e[7];
    if (KIND != INTERIOR) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const float m = edge ? v[i] : 0.f;
            e[i] = KIND == LEFT ? quad_reverse<0, 0>((m + row_shr<1>(m)) + row_shr<2>(m)) : quad_reverse<3, 3>((m + row_shl<1>(m)) + row_shl<2>(m));
        }
    }
    float r0, r1, r2, r3, r4, r5, r6, u0, u1, u2, u3, u4, u5, u6;
    asm("s_nop 1\n"
        "v_add_f32_dpp %7, %0, %0" SQD_DPP_SHR "v_add_f32_dpp %8, %1, %1" SQD_DPP_SHR "v_add_f32_dpp %9, %2, %2" SQD_DPP_SHR "v_add_f32_dpp %10, %3, %3" SQD_DPP_SHR
        "v_add_f32_dpp %11, %4, %4" SQD_DPP_SHR "v_add_f32_dpp %12, %5, %5" SQD_DPP_SHR "v_add_f32_dpp %13, %6, %6" SQD_DPP_SHR
        "v_add_f32_dpp %7, %7, %0" SQD_DPP_SHR "v_add_f32_dpp %8, %8, %1" SQD_DPP_SHR "v_add_f32_dpp %9, %9, %2" SQD_DPP_SHR "v_add_f32_dpp %10, %10, %3" SQD_DPP_SHR
        "v_add_f32_dpp %11, %11, %4" SQD_DPP_SHR "v_add_f32_dpp %12, %12, %5" SQD_DPP_SHR "v_add_f32_dpp %13, %13, %6" SQD_DPP_SHR
        "v_add_f32_dpp %7, %7, %0" SQD_DPP_SHR "v_add_f32_dpp %8, %8, %1" SQD_DPP_SHR "v_add_f32_dpp %9, %9, %2" SQD_DPP_SHR "v_add_f32_dpp %10, %10, %3" SQD_DPP_SHR
        "v_add_f32_dpp %11, %11, %4" SQD_DPP_SHR "v_add_f32_dpp %12, %12, %5" SQD_DPP_SHR "v_add_f32_dpp %13, %13, %6" SQD_DPP_SHR
        "v_add_f32_dpp %14, %0, %0" SQD_DPP_SHL "v_add_f32_dpp %15, %1, %1" SQD_DPP_SHL "v_add_f32_dpp %16, %2, %2" SQD_DPP_SHL "v_add_f32_dpp %17, %3, %3" SQD_DPP_SHL
        "v_add_f32_dpp %18, %4, %4" SQD_DPP_SHL "v_add_f32_dpp %19, %5, %5" SQD_DPP_SHL "v_add_f32_dpp %20, %6, %6" SQD_DPP_SHL
        "v_add_f32_dpp %14, %14, %0" SQD_DPP_SHL "v_add_f32_dpp %15, %15, %1" SQD_DPP_SHL "v_add_f32_dpp %16, %16, %2" SQD_DPP_SHL "v_add_f32_dpp %17, %17, %3" SQD_DPP_SHL
        "v_add_f32_dpp %18, %18, %4" SQD_DPP_SHL "v_add_f32_dpp %19, %19, %5" SQD_DPP_SHL "v_add_f32_dpp %20, %20, %6" SQD_DPP_SHL
        "v_add_f32_dpp %0, %14, %7" SQD_DPP_SHL "v_add_f32_dpp %1, %15, %8" SQD_DPP_SHL "v_add_f32_dpp %2, %16, %9" SQD_DPP_SHL "v_add_f32_dpp %3, %17, %10" SQD_DPP_SHL
        "v_add_f32_dpp %4, %18, %11" SQD_DPP_SHL "v_add_f32_dpp %5, %19, %12" SQD_DPP_SHL "v_add_f32_dpp %6, %20, %13" SQD_DPP_SHL
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5),
          "=&v"(r6), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6));
    if (KIND != INTERIOR) {
#pragma unroll
        for (int i = 0; i < 7; ++i) v[i] += e[i];
    }
}

// one row of a lane's column: target rgb and the (pred_0, pred_1) rgb pairs
struct Raw {
    v3f t;
    v2f w[3];
};
// the window statistics of one lane's column over some rows: sum t, sum t^2, sum w, sum w^2, sum w t (24 numbers; sum t^2 joins
// sum w^2 before the horizontal pass — sigma_x + sigma_y only needs their sum — so 21 quantities cross the lanes)
struct Sums {
    v3f St, Stt;
    v2f Sw[3], Sq[3], Swt[3];
};
__device__ __forceinline__ void clear(Sums &A) {
    A.St = v3f{0.f, 0.f, 0.f};
    A.Stt = v3f{0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) A.Sw[c] = A.Sq[c] = A.Swt[c] = splat(0.f);
}
// A += the products of one row: five instructions per colour (add, fma, packed add, two packed fma) — the products are never
// formed on their own (a packed multiply costs what two scalar ones do on gfx950; a packed fma 1.45x one scalar fma)
__device__ __forceinline__ void accumulate_row(Sums &A, const Raw &R) {
    A.St += R.t;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        A.Stt[c] = fmaf(R.t[c], R.t[c], A.Stt[c]);
        A.Sw[c] += R.w[c];
        A.Sq[c] = pfma(R.w[c], R.w[c], A.Sq[c]);
        A.Swt[c] = pfma(R.w[c], splat(R.t[c]), A.Swt[c]);
    }
}

// A launch handles a PAIR of source frames (the two halves of the packed-math registers).  S = 2 (frame_ids [0,-1,1]) is one
// launch; S = 3 (--use_stereo: frame_ids [0,-1,1,"s"], reference trainer.py:52-53) runs pairs (0,1) and (2,2): the running
// per-pixel minimum and its argmin travel between the launches through the `sel` / `idx` outputs, in the candidate order of
// torch.cat + torch.min (identity_0..S-1, reproj_0..S-1 — first minimum wins).
// Tiling of a launch.  Uniform: every (image, column strip) is cut into nsy tiles of TR rows.  Balanced (n_lo > 0; the fused forward's
// default where it applies): the first n_hi_cols of the B * nsx image columns are cut into n_lo + 1 nearly equal tiles, the others into
// n_lo, so that the launch has a whole number of tiles per CU — all of (nearly) the same cost: 1024 tiles of 24..28 rows at config B
// instead of 1584 of 16 (6.19 per CU) or 924 of 28 (3.61 per CU): -3 % (49.6 against 51.2 us).
struct Tiling {
    int TR, nsx, nsy, ntiles, nblk8;
    int n_lo, n_hi_cols;
    int skew;          // fused forward: cycles the second workgroup of a CU waits at its start (0: none)
    int pipe;          // lean forward: two rows of a wave in flight in phase 1 (warp_tile_pipe)
    int late;          // lean forward: the rows that do not fill a round of the waves are warped after the barrier (warp_tile)
};
__device__ __forceinline__ void tile_of(const Tiling &tl, int tile, int H, int &b, int &tx, int &y0, int &own_rows) {
    if (tl.n_lo <= 0) {
        tx = tile % tl.nsx;
        const int t2 = tile / tl.nsx, ty = t2 % tl.nsy;
        b = t2 / tl.nsy;
        y0 = ty * tl.TR;
        own_rows = min(tl.TR, H - y0);
        return;
    }
    const int nhi = tl.n_hi_cols * (tl.n_lo + 1);
    int col, k, n;
    if (tile < nhi) { n = tl.n_lo + 1; col = tile / n; k = tile - col * n; }
    else { n = tl.n_lo; const int t2 = tile - nhi; col = tl.n_hi_cols + t2 / n; k = t2 - (t2 / n) * n; }
    b = col / tl.nsx;
    tx = col - b * tl.nsx;
    const int Hh = (H + 1) / 2;                          // tiles start on even rows (phase 2 works on row pairs)
    y0 = 2 * ((k * Hh) / n);
    own_rows = min(H, 2 * (((k + 1) * Hh) / n)) - y0;
}

struct PairPass {
    int s0, s1;        // source indices of the .x / .y halves (s1 == s0: an odd source count's last pass)
    int S;             // number of source frames
    int first, last;   // first pass: start from the identity maps; last pass: emit identity_selection and the loss partial
};

// MODE 0: identity maps ("pred" = the source frames themselves; output = loss + 1e-5 * noise)    trainer.py:480-487,514-517
// MODE 1: fused warp + SSIM + L1 + per-pixel min / auto-mask                                     trainer.py:386-532
// MODE 2: d loss / d (window sums) of the winning source ("coefficient planes") for the backward, from the stored warps
template <int MODE>
struct Ctx {
    __amdgpu_buffer_rsrc_t tgt;       // image b of the target, [3][H][W]
    __amdgpu_buffer_rsrc_t p0, p1;    // MODE 0: sources, MODE 2: warped images (image b)
    const v2f *wl;                    // MODE 1: LDS tile [row][3][64]
    const float *tt;                  // MODE 1, wide edition: the target rows of the tile in LDS, [row][3][64] (nullptr: from memory)
    float *selp;                      // MODE 1, fast edition: image b of identity_selection / argmin (p0 = descriptor of its identity maps)
    uint8_t *idxp;
    float *w0, *w1;                   //   and of the warped outputs
    int H, W, y0, x, lane;            // tile's first owned row, this lane's column
    const int *late_ctr = nullptr;    // MODE 1, fast edition: LDS count of the late rows warped so far, the first late tile row, their number (warp_tile)
    int late_r0 = 1 << 20, late_n = 0;
    unsigned HW;
    unsigned xoff;                    // byte offset of the lane's column — beyond every buffer for lanes outside the image
};
__device__ __forceinline__ float bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int MODE, bool WIDE = false, bool NOREFL = false, bool PX = false>
__device__ __forceinline__ void load_row(const Ctx<MODE> &k, int r, Raw &R) {
    // tile row r <-> image row y0 - 3 + r, reflected into the image (ReflectionPad2d(3), layers.py:26).  Raw buffer loads: a
    // lane outside the image (W < 64 only) carries an offset beyond the descriptor and reads zeros — no branch.
    // (NOREFL: a tile whose halo rows all lie inside the image — no reflection arithmetic, row offsets are compile-time multiples)
    const int yr = NOREFL ? k.y0 - 3 + r : reflect_idx(k.y0 - 3 + r, k.H);
    const unsigned row = (unsigned)(yr * k.W) * 4u;           // wave-uniform: travels in the scalar offset
    if (MODE == 1 && WIDE) {
        const int rr = yr - (k.y0 - 3);
#pragma unroll
        for (int c = 0; c < 3; ++c) R.t[c] = k.tt[(rr * 3 + c) * 64 + k.lane];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) R.t[c] = bld(k.tgt, k.xoff, row + c * k.HW * 4u);
    }
    if (MODE == 1) {
        const int rr = yr - (k.y0 - 3);
#pragma unroll
        for (int c = 0; c < 3; ++c) R.w[c] = k.wl[(rr * 3 + c) * 64 + k.lane];
    } else if (MODE == 0 && PX) {       // pixel-interleaved sources: the lane's three colours are 12 consecutive bytes
        typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 s0 = __builtin_amdgcn_raw_buffer_load_b96(k.p0, k.xoff * 3u, row * 3u, 0), s1 = __builtin_amdgcn_raw_buffer_load_b96(k.p1, k.xoff * 3u, row * 3u, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) R.w[c] = v2f{__uint_as_float(s0[c]), __uint_as_float(s1[c])};
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) R.w[c] = v2f{bld(k.p0, k.xoff, row + c * k.HW * 4u), bld(k.p1, k.xoff, row + c * k.HW * 4u)};
    }
}

struct SsimOut {
    v2f loss;              // 0.85 * ssim.mean(1) + 0.15 * l1.mean(1) for (pred_0, pred_1)       trainer.py:446-451
    float g0[9], g1[9];    // GRAD: d loss_s / d (sum w_c, sum w_c^2, sum w_c t_c), c = 0..2
};

// SSIM + L1 of both predictions against the target from the window sums (already box-summed) and the centre row
// Forward paths (identity maps, fused forward): the same quotient on the RAW window sums.  With mu = S / 49 every factor of
// SSIM_n / SSIM_d carries 49^-2, which cancels: A1' = 2 Sw St + 49^2 C1, A2' = 2 (49 Swt - Sw St) + 49^2 C2, B1' = Sw^2 + St^2 + 49^2 C1,
// B2' = 49 (Sww + Stt) - (Sw^2 + St^2) + 49^2 C2 — 11 packed instructions per colour instead of 18 (no means are formed), the
// quotient as v_rcp + one Newton step on the quotient (<= 1 ulp of the correctly rounded value; the conditioning of the sigma
// terms, 49 Sww - Sw^2 against 49^2 C2, is that of layers.py:35-46's own mu / sigma form).
__device__ __forceinline__ v2f ssim_l1_fwd(const Sums &S, const Raw &ctr, int flags) {
    constexpr float K1 = C1 * 2401.f, K2 = C2 * 2401.f;
    // (every step for the three colours before the next step: a packed instruction that reads the previous one's result costs an s_nop
    //  on gfx950, and the compiler left the third colour's chain — 20 of them per output row — alone at the end)
    v2f p[3], A1[3], A2[3], q[3], B1[3], B2[3], num[3], den[3], rd[3], q0[3], Sv[3], r[3], df[3];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = S.Sw[c] * splat(S.St[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = pfma(S.Sw[c], S.Sw[c], splat(S.St[c] * S.St[c]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) A1[c] = pfma(splat(2.f), p[c], splat(K1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) A2[c] = pfma(splat(49.f), S.Swt[c], -p[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) B1[c] = q[c] + splat(K1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) B2[c] = pfma(splat(49.f), S.Sq[c], splat(K2) - q[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) A2[c] = pfma(splat(2.f), A2[c], splat(K2));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) den[c] = B1[c] * B2[c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) num[c] = A1[c] * A2[c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) rd[c] = v2f{__builtin_amdgcn_rcpf(den[c].x), __builtin_amdgcn_rcpf(den[c].y)};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) df[c] = splat(ctr.t[c]) - ctr.w[c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) q0[c] = num[c] * rd[c];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) Sv[c] = pfma(-den[c], q0[c], num[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) Sv[c] = pfma(Sv[c], rd[c], q0[c]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = pfma(splat(-0.5f), Sv[c], splat(0.5f));
    __builtin_amdgcn_sched_barrier(0);
    v2f ssim_sum = splat(0.f), l1 = splat(0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ssim_sum += v2f{__builtin_amdgcn_fmed3f(r[c].x, 0.f, 1.f), __builtin_amdgcn_fmed3f(r[c].y, 0.f, 1.f)};      // torch.clamp(., 0, 1)
        l1 += v2f{fabsf(df[c].x), fabsf(df[c].y)};
    }
    if (flags & SQD_LOSS_NO_SSIM) return l1 * splat(1.f / 3.f);          // --no_ssim: the L1 term alone (trainer.py:447-448)
    return splat(0.85f) * (ssim_sum * splat(1.f / 3.f)) + splat(0.15f) * (l1 * splat(1.f / 3.f));
}

template <bool GRAD>
__device__ __forceinline__ void ssim_l1(const Sums &S, const Raw &ctr, SsimOut &o, int flags) {
    if constexpr (!GRAD) {
        o.loss = ssim_l1_fwd(S, ctr, flags);
        return;
    }
    v2f ssim_sum = splat(0.f), l1 = splat(0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const v2f Sw = S.Sw[c], Sq = S.Sq[c], Swt = S.Swt[c];
        // layers.py:35-46
        const float mt = S.St[c] * INV49;
        const v2f mw = Sw * splat(INV49);
        const v2f mwmt = mw * splat(mt);
        const v2f sxy = Swt * splat(INV49) - mwmt;
        const v2f mm = pfma(mw, mw, splat(mt * mt));
        const v2f A1 = pfma(splat(2.f), mwmt, splat(C1)), A2 = pfma(splat(2.f), sxy, splat(C2));
        const v2f B1 = mm + splat(C1), B2 = (Sq * splat(INV49) - mm) + splat(C2);
        const v2f Bd = B1 * B2;
        const v2f Sv = div_core2(A1 * A2, Bd, rcp_refined2(Bd));       // SSIM_n / SSIM_d, the reference's division (layers.py:46)
        const v2f r = (splat(1.f) - Sv) * splat(0.5f);
        ssim_sum += v2f{__builtin_amdgcn_fmed3f(r.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(r.y, 0.f, 1.f)};      // torch.clamp(., 0, 1)
        const v2f df = splat(ctr.t[c]) - ctr.w[c];
        l1 += v2f{fabsf(df.x), fabsf(df.y)};
        if (GRAD) {
            const v2f iB1 = v2f{__builtin_amdgcn_rcpf(B1.x), __builtin_amdgcn_rcpf(B1.y)};
            const v2f iB2 = v2f{__builtin_amdgcn_rcpf(B2.x), __builtin_amdgcn_rcpf(B2.y)};
            const v2f iB = iB1 * iB2;
            // torch.clamp passes the gradient at the bounds; window mean -> sum
            const v2f kk = v2f{(r.x >= 0.f && r.x <= 1.f) ? -0.5f * INV49 : 0.f, (r.y >= 0.f && r.y <= 1.f) ? -0.5f * INV49 : 0.f};
            const v2f dmu = splat(2.f) * (splat(mt) * (A2 - A1) * iB + mw * Sv * (iB2 - iB1));
            const v2f g0 = kk * dmu, g1 = kk * (-Sv * iB2), g2 = kk * (splat(2.f) * A1 * iB);
            o.g0[c] = g0.x; o.g1[c] = g0.y;
            o.g0[3 + c] = g1.x; o.g1[3 + c] = g1.y;
            o.g0[6 + c] = g2.x; o.g1[6 + c] = g2.y;
        }
    }
    o.loss = splat(0.85f) * (ssim_sum * splat(1.f / 3.f)) + splat(0.15f) * (l1 * splat(1.f / 3.f));
    if (flags & SQD_LOSS_NO_SSIM) o.loss = l1 * splat(1.f / 3.f);          // --no_ssim: the L1 term alone (trainer.py:447-448)
}

// the per-pixel minimum over [identity_0..S-1, reproj_0..S-1] (torch.cat + torch.min, first minimum wins: trainer.py:519-526) for the
// pair's two reprojection losses, identity_selection, the argmin byte and the loss partial.
// The argmin byte: < NS an identity map won (auto-mask: no gradient), NS + s reprojection of source s won; under --avg_reprojection NS
// stands for "the mean of all reprojections" (every source, weight 1 / S)
__device__ __forceinline__ void select_store(const sqd_photo_args &a, const PairPass &pp, v2f loss, int b, unsigned qo, unsigned HW, float &loss_acc) {
    const int flags = a.loss_flags, NS = pp.S;
    const bool avg = flags & SQD_LOSS_AVG_REPROJECTION;
    float best;
    int bi;
    const bool automask = !(flags & SQD_LOSS_NO_AUTOMASK);
    if (avg) {
        best = INFINITY;                                           // (set below, by the last pair pass)
        bi = 0;
    } else if (pp.first) {
        best = INFINITY;                                           // --disable_automasking: no identity candidates (trainer.py:520-521)
        bi = 0;
        if (automask) {
            const float *idm = a.identity + (size_t)b * (avg ? 1 : NS) * HW;
            best = ldg(idm, qo * 4u);
#pragma unroll
            for (int i = 1; i < SQD_MAX_SOURCES; ++i)
                if (i < NS && !avg) {
                    const float v = ldg(idm, (qo + i * HW) * 4u);
                    if (v < best) { best = v; bi = i; }
                }
        }
    } else {                                                       // the running minimum of the earlier pairs
        best = a.sel[(size_t)b * HW + qo];
        bi = a.idx[(size_t)b * HW + qo];
    }
    if (avg) {
        // trainer.py:508-509: the mean over the S reprojection losses is the one reprojection candidate.  More than two sources: the sum
        // travels through `sel` between the pair passes, the identity candidate is only looked at by the last one
        float sum = pp.s1 != pp.s0 ? loss.x + loss.y : loss.x;
        if (!pp.first) sum += a.sel[(size_t)b * HW + qo];
        if (!pp.last) {
            if (a.reproj) {
                a.reproj[((size_t)b * NS + pp.s0) * HW + qo] = loss.x;
                a.reproj[((size_t)b * NS + pp.s1) * HW + qo] = loss.y;
            }
            a.sel[(size_t)b * HW + qo] = sum;
            return;
        }
        const float m = NS == 2 ? sum * 0.5f : sum / (float)NS;
        best = INFINITY;
        bi = 0;
        if (automask) best = ldg(a.identity + (size_t)b * HW, qo * 4u);
        if (m < best) { best = m; bi = NS; }
    } else {
        if (loss.x < best) { best = loss.x; bi = NS + pp.s0; }
        if (loss.y < best) { best = loss.y; bi = NS + pp.s1; }
    }
    if (a.reproj) {
        a.reproj[((size_t)b * NS + pp.s0) * HW + qo] = loss.x;
        a.reproj[((size_t)b * NS + pp.s1) * HW + qo] = loss.y;
    }
    if (pp.last) {
        loss_acc += best;
        if (a.sel) a.sel[(size_t)b * HW + qo] = bi >= NS ? 1.f : 0.f;   // trainer.py:529-530
    } else {
        a.sel[(size_t)b * HW + qo] = best;
    }
    if (a.idx) a.idx[(size_t)b * HW + qo] = (uint8_t)bi;
}

// select_store for the configuration every training step of the reference's default options runs (two source frames in one pair pass, auto-mask
// on, per-pixel minimum, no reprojection dump): the ~150 instructions select_store spends per output row on wave-uniform option tests (exec
// mask bookkeeping, 64-bit pointer arithmetic per candidate) become 2 loads, 3 compare / select pairs and 2 stores.  `ident`: descriptor of
// image b of the identity maps ([2][H][W]); `selp` / `idxp`: image b of the outputs.
__device__ __forceinline__ void select_store_fast(__amdgpu_buffer_rsrc_t ident, float *__restrict__ selp, uint8_t *__restrict__ idxp, v2f loss, unsigned qo, unsigned HW,
                                                  float &loss_acc) {
    float best = bld(ident, qo * 4u, 0);
    const float v1 = bld(ident, qo * 4u, HW * 4u);
    int bi = 0;
    if (v1 < best) { best = v1; bi = 1; }
    if (loss.x < best) { best = loss.x; bi = 2; }
    if (loss.y < best) { best = loss.y; bi = 3; }
    loss_acc += best;
    selp[qo] = bi >= 2 ? 1.f : 0.f;                                   // trainer.py:529-530
    idxp[qo] = (uint8_t)bi;
}

template <int MODE, int KIND, bool FAST = false>
__device__ __forceinline__ void finish_row(const sqd_photo_args &a, const PairPass &pp, const float *noise, const Ctx<MODE> &k, Sums &S, const Raw &ctr,
                                           bool edge, int b, int yo, bool own, float &loss_acc) {
    if constexpr (FAST) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float q[7] = {S.St[c], S.Sw[c].x, S.Sw[c].y, S.Sq[c].x, S.Sq[c].y, S.Swt[c].x, S.Swt[c].y};
            box7x7<KIND>(q, edge);
            S.St[c] = q[0];
            S.Sw[c] = v2f{q[1], q[2]}; S.Sq[c] = v2f{q[3], q[4]}; S.Swt[c] = v2f{q[5], q[6]};
        }
    } else {
        {
            float s0 = S.St.x, s1 = S.St.y, s2 = S.St.z;
            box7x3<KIND>(s0, s1, s2, edge);
            S.St = v3f{s0, s1, s2};
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) box7x3<KIND>(S.Sw[c], S.Sq[c], S.Swt[c], edge);
    }
    SsimOut o;
    const int flags = FAST ? 0 : a.loss_flags;
    const bool avg = flags & SQD_LOSS_AVG_REPROJECTION;
    ssim_l1<MODE == 2>(S, ctr, o, flags);
    if (!own) return;
    const unsigned HW = k.HW;
    const unsigned qo = (unsigned)(yo * k.W + k.x);
    const int NS = pp.S;
    if (MODE == 0) {
        const int NI = avg ? 1 : NS;                                   // identity maps (and noise planes) per image
        float *out = a.sel + (size_t)b * NI * HW;                      // (the identity maps travel through `sel`)
        const float *nz = noise ? noise + (size_t)b * NI * HW : nullptr;
        const unsigned q0 = qo + (unsigned)pp.s0 * HW, q1 = qo + (unsigned)pp.s1 * HW;
        if (avg) {       // --avg_reprojection: ONE identity map per image, the mean over the S sources (trainer.py:489-490); more than
            //              two sources: the sum travels through the output plane from pair pass to pair pass
            float sum = pp.s1 != pp.s0 ? o.loss.x + o.loss.y : o.loss.x;
            if (!pp.first) sum += ldg(out, qo * 4u);
            if (pp.last) sum = (NS == 2 ? sum * 0.5f : sum / (float)NS) + (nz ? ldg(nz, qo * 4u) : 0.f) * 0.00001f;
            stg(out, qo * 4u, sum);
        } else {
            stg(out, q0 * 4u, o.loss.x + (nz ? ldg(nz, q0 * 4u) : 0.f) * 0.00001f);              // trainer.py:514-517
            if (pp.s1 != pp.s0) stg(out, q1 * 4u, o.loss.y + (nz ? ldg(nz, q1 * 4u) : 0.f) * 0.00001f);
        }
    } else if (MODE == 1) {
        if constexpr (FAST) select_store_fast(k.p0, k.selp, k.idxp, o.loss, qo, HW, loss_acc);
        else select_store(a, pp, o.loss, b, qo, HW, loss_acc);
    } else {
        // the planes are fully written: zeros where an identity candidate won (the first pass lays them down; a later pass
        // of a 3- or 4-source run only overwrites the pixels its own sources won), so the backward reads them unmasked
        const int bi = a.idx[(size_t)b * HW + qo];
        const float scale = (flags & SQD_LOSS_NO_SSIM) ? 0.f : 0.85f / 3.f;          // --no_ssim: no window terms at all
        if (avg) {                         // every source carries 1 / S of the gradient wherever the mean reprojection won: 9 S planes
            const bool won = bi == NS;
            float *co = a.coef + (size_t)b * 9 * NS * HW;
            const float sc = scale / (float)NS;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                stg(co, (qo + (9 * pp.s0 + j) * HW) * 4u, won ? o.g0[j] * sc : 0.f);
                if (pp.s1 != pp.s0) stg(co, (qo + (9 * pp.s1 + j) * HW) * 4u, won ? o.g1[j] * sc : 0.f);
            }
            return;
        }
        const bool won = bi == NS + pp.s0 || bi == NS + pp.s1;
        if (won || pp.first) {
            float *co = a.coef + (size_t)b * 9 * HW;
            const bool first = bi == NS + pp.s0;
#pragma unroll
            for (int j = 0; j < 9; ++j) stg(co, (qo + j * HW) * 4u, won ? (first ? o.g0[j] : o.g1[j]) * scale : 0.f);
        }
    }
}

template <int NW>
__device__ __forceinline__ void store_warped_wide(const v2f *wl, float *__restrict__ w0, float *__restrict__ w1, int H, int W, int y0, int x0, int own_rows,
                                                  int lane, int wave, int k0 = 0, int k1 = 1 << 20);
// phase 2 for the output rows of one wave: pairs (j, j+1) of tile rows share six of their seven window rows
template <int MODE, int KIND, bool WIDE = false, bool FAST = false, bool NOREFL = false, bool PX = false>
__device__ __forceinline__ void ssim_rows(const sqd_photo_args &a, const PairPass &pp, const float *noise, const Ctx<MODE> &k, bool edge, int b,
                                          int wave, int nwaves, int own_rows, bool own_col, float &loss_acc) {
    int kst = 0;
    bool late_seen = !(MODE == 1 && FAST) || k.late_n == 0;
    for (int p = wave; 2 * p < own_rows; p += nwaves) {
        const int j = 2 * p;                         // tile rows j .. j+7 feed the outputs y0+j (centre j+3) and y0+j+1 (centre j+4)
        if constexpr (MODE == 1 && FAST) {
            if (!late_seen && (!NOREFL || j + 8 > k.late_r0)) {      // this pair's window reaches a row warped after the barrier (warp_tile)
                while (__hip_atomic_load(k.late_ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < k.late_n) __builtin_amdgcn_s_sleep(2);
                late_seen = true;
            }
        }
        if constexpr (MODE == 1 && FAST) {           // three of the wave's 16-byte warped-store iterations per row pair (<= 6 in all: tiles of <= 64 rows)
            store_warped_wide<8>(k.wl, k.w0, k.w1, k.H, k.W, k.y0, k.x - k.lane, own_rows, k.lane, wave, kst, kst + 3);
            kst += 3;
        }
        Sums core;                                   // rows j+1 .. j+6: shared by both outputs
        clear(core);
        {
            Raw R;
#pragma unroll
            for (int i = 1; i < 7; ++i) {
                load_row<MODE, WIDE, NOREFL, PX>(k, j + i, R);
                accumulate_row(core, R);
            }
        }
#pragma nounroll
        for (int o = 0; o < 2; ++o) {
            Raw R, ctr;
            load_row<MODE, WIDE, NOREFL, PX>(k, j + 7 * o, R);
            load_row<MODE, WIDE, NOREFL, PX>(k, j + 3 + o, ctr);
            Sums S = core;
            accumulate_row(S, R);
#pragma unroll
            for (int c = 0; c < 3; ++c) S.Sq[c] += splat(S.Stt[c]);
            finish_row<MODE, KIND, FAST>(a, pp, noise, k, S, ctr, edge, b, k.y0 + j + o, own_col && j + o < own_rows, loss_acc);
        }
    }
    if constexpr (MODE == 1 && FAST) store_warped_wide<8>(k.wl, k.w0, k.w1, k.H, k.W, k.y0, k.x - k.lane, own_rows, k.lane, wave, kst);
}


// ---- phase 1 pieces ---------------------------------------------------------------------------------------------
// One warped cell = one target pixel seen in both source views.  Round 4 layout of the arithmetic:
//  * the projection / normalisation chain is the canonical fp32 order of oracle/warp_chain.c (layers.py:211-212,250-257), with two
//    identities that leave every bit where it was: (gx + 1) * 0.5 * (W - 1) = fl(fl(2 t + 1) * (0.5 (W - 1))) for t = un - 0.5 (a product
//    by 0.5 is exact, 0.5 (W - 1) is representable), and gx = t + t;
//  * grid_sample's border rules need NO masks: the coordinate is clamped into [0, W - 1], so x0 + 1 can only leave the image when
//    ix == W - 1 exactly, where ax = ix - floor(ix) is exactly 0 and with it the weights of both out-of-range taps (ATen skips those
//    taps: adding 0 * finite is the same sum); likewise y.  The tap pairs are fetched through RAW BUFFER loads whose descriptor
//    spans image b of the source: the pair (W - 1, W) of a row reads the first pixel of the next row or plane (finite image data,
//    weight 0), the one pair that would leave the image reads 0 — no address clamp, no weight shift (round 3: 22 selects + 8 compares
//    per cell);
//  * the two taps of a row arrive in consecutive registers, so the blend multiplies PAIRS (nw, ne) * (wnw, wne) and adds the halves —
//    2 packed + 1 scalar instruction per colour and source, no register shuffling (round 3 packed (source 0, source 1): 16 moves).
struct Cell {
    v2f gx, gy;                      // normalised sampling grid (outputs[("sample", f, 0)]) of (source 0, source 1)
    v2f wn0, ws0, wn1, ws1;          // (west, east) bilinear weights of the north / south tap pair, source 0 and source 1
    unsigned o0, o1;                 // byte offset of the north pair inside image b of source 0 / source 1
    int x00, y00, x01, y01;          // integer north-west taps
};
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f bld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));      // 8 bytes, 4-byte aligned
}

__device__ __forceinline__ void project_cell(Cell &c, float d, float fx, float fy, const float *ik, const v2f *P, v2f rW, v2f rH,
                                             float wm1, float hm1, int W) {
    float X[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = ik[i * 3 + 0] * fx;
        acc = fmaf(ik[i * 3 + 1], fy, acc);
        acc = fmaf(ik[i * 3 + 2], 1.0f, acc);
        X[i] = d * acc;
    }
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
    // the x and y quotients as two interleaved chains (div_core2's steps, statement by statement): a packed instruction that reads the
    // previous one's result costs an s_nop on gfx950 — the serial form carried 20 of them per row of cells
    v2f u, v, tx, ty;
    div_core2x2(cam[0], cam[1], z, z, rz, rz, u, v);
    div_core2x2(u, v, splat(wm1), splat(hm1), rW, rH, tx, ty);
    tx -= splat(0.5f);
    ty -= splat(0.5f);
    c.gx = tx + tx;                                                     // (un - 0.5) * 2
    c.gy = ty + ty;
    v2f ix = pfma(splat(2.0f), tx, splat(1.0f)) * splat(0.5f * wm1);   // ((gx + 1) * 0.5) * (W - 1), bit for bit
    v2f iy = pfma(splat(2.0f), ty, splat(1.0f)) * splat(0.5f * hm1);
    ix = v2f{__builtin_amdgcn_fmed3f(ix.x, 0.f, wm1), __builtin_amdgcn_fmed3f(ix.y, 0.f, wm1)};      // border padding: clamp
    iy = v2f{__builtin_amdgcn_fmed3f(iy.x, 0.f, hm1), __builtin_amdgcn_fmed3f(iy.y, 0.f, hm1)};
    const v2f fx0 = v2f{floorf(ix.x), floorf(ix.y)}, fy0 = v2f{floorf(iy.x), floorf(iy.y)};
    const v2f ax = ix - fx0, ay = iy - fy0;
    const v2f bx = (fx0 + splat(1.f)) - ix, by = (fy0 + splat(1.f)) - iy;
    c.x00 = (int)fx0.x; c.y00 = (int)fy0.x; c.x01 = (int)fx0.y; c.y01 = (int)fy0.y;
    const v2f h0 = v2f{bx.x, ax.x}, h1 = v2f{bx.y, ax.y};             // (west, east) of source 0 / source 1
    c.wn0 = h0 * splat(by.x); c.ws0 = h0 * splat(ay.x);
    c.wn1 = h1 * splat(by.y); c.ws1 = h1 * splat(ay.y);
    c.o0 = (unsigned)(c.y00 * W + c.x00) * 4u;
    c.o1 = (unsigned)(c.y01 * W + c.x01) * 4u;
}

// the twelve tap pairs of a cell (2 sources x 3 colours x north / south), issued back to back
__device__ __forceinline__ void gather_taps(const Cell &c, __amdgpu_buffer_rsrc_t r0, __amdgpu_buffer_rsrc_t r1, unsigned HW4, unsigned W4,
                                            v2f t0[3][2], v2f t1[3][2]) {
    const unsigned s0 = c.o0 + W4, s1 = c.o1 + W4;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        t0[ch][0] = bld2(r0, c.o0, ch * HW4);
        t1[ch][0] = bld2(r1, c.o1, ch * HW4);
        t0[ch][1] = bld2(r0, s0, ch * HW4);
        t1[ch][1] = bld2(r1, s1, ch * HW4);
    }
}

// west + east half of a blended pair: a plain v_add_f32 (kept out of the SLP vectorizer's reach — it would pair the six adds of a cell
// into three packed ones behind nine register moves)
__device__ __forceinline__ float hadd(v2f a) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a.x), "v"(a.y));
    return r;
}
__device__ __forceinline__ v2f blend_taps(const Cell &c, const v2f t0[2], const v2f t1[2]) {
    const v2f a = pfma(t0[1], c.ws0, t0[0] * c.wn0), b = pfma(t1[1], c.ws1, t1[0] * c.wn1);
    return v2f{hadd(a), hadd(b)};
}

// what phase 1 stores to HBM for the cells a tile owns; resolved once per tile (round 3 re-read six pointers from the kernel
// arguments for every row, an s_load + wait each)
struct WarpOut {
    float2 *smp0, *smp1;             // image b of sample[s0] / sample[s1] (nullptr: not stored)
    int2 *tap0, *tap1;
    float *w0, *w1;                  // image b of warped[s0] / warped[s1]
};

// bilinear blend, the tile's LDS row, and — for cells the tile owns — sample / warped / taps in HBM
template <bool VIRT, bool WIDE = false, bool FAST = false>
__device__ __forceinline__ void store_cell(const WarpOut &o, const Cell &c, const v2f wv[3], v2f *wl, int r, int lane, bool col_ok, bool own,
                                           unsigned HW, unsigned off) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)      // (lanes beyond a narrow image carry zeros for the shuffles)
        wl[(r * 3 + ch) * 64 + lane] = (!VIRT || col_ok) ? wv[ch] : splat(0.f);
    if (own) {
        if (FAST || o.smp0) stg2(o.smp0, off * 8u, c.gx.x, c.gy.x);
        if (FAST || o.smp1) stg2(o.smp1, off * 8u, c.gx.y, c.gy.y);
        if (!FAST && o.tap0) stg2i(o.tap0, off * 8u, c.x00, c.y00);
        if (!FAST && o.tap1) stg2i(o.tap1, off * 8u, c.x01, c.y01);
        if (!WIDE && o.w0) {       // (wide edition: the warped colours leave from the LDS tile, 16 bytes per lane — store_warped_wide)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) stg(o.w0, (off + ch * HW) * 4u, wv[ch].x);
        }
        if (!WIDE && o.w1) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) stg(o.w1, (off + ch * HW) * 4u, wv[ch].y);
        }
    }
}
template <bool VIRT, bool WIDE = false, bool FAST = false>
__device__ __forceinline__ void finish_cell(const WarpOut &o, const Cell &c, const v2f t0[3][2], const v2f t1[3][2], v2f *wl, int r, int lane,
                                            bool col_ok, bool own, unsigned HW, unsigned off) {
    v2f wv[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) wv[ch] = blend_taps(c, t0[ch], t1[ch]);
    store_cell<VIRT, WIDE, FAST>(o, c, wv, wl, r, lane, col_ok, own, HW, off);
}

// ---- pixel-interleaved sources (SQD_SOURCES_HWC: the frames as [B,H,W,3]) ---------------------------------------------------------------
// The (west, east) taps of a row are 24 consecutive bytes there — R0 G0 B0 R1 | G1 B1 —: one 16-byte and one 8-byte gather per row and
// source, EIGHT vector-memory instructions per cell instead of the planar layout's twelve (phase 1 sits on the address path at 16 clocks
// per gather: DESIGN.md 3.1).  The blend multiplies the register pairs as they arrive — (R0, G0) by the west weights, (B0, R1) by
// (west, east), (G1, B1) by the east weights — and adds the halves across pairs: every product and sum is the planar blend's own
// (west = fma(s, w_sw, n * w_nw), east likewise, colour = west + east), so the two layouts give the same bits.
struct PxTaps {
    v4f n4, s4;                      // R0 G0 B0 R1 of the north / south row
    v2f n2, s2;                      // G1 B1
};
__device__ __forceinline__ v4f bld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));     // 16 bytes, 4-byte aligned
}
__device__ __forceinline__ void gather_taps_px(const Cell &c, __amdgpu_buffer_rsrc_t r0, __amdgpu_buffer_rsrc_t r1, unsigned W12, PxTaps &t0,
                                               PxTaps &t1) {
    const unsigned n0 = c.o0 * 3u, n1 = c.o1 * 3u, s0 = n0 + W12, s1 = n1 + W12;
    t0.n4 = bld4(r0, n0, 0);
    t1.n4 = bld4(r1, n1, 0);
    t0.s4 = bld4(r0, s0, 0);
    t1.s4 = bld4(r1, s1, 0);
    t0.n2 = bld2(r0, n0 + 16u, 0);
    t1.n2 = bld2(r1, n1 + 16u, 0);
    t0.s2 = bld2(r0, s0 + 16u, 0);
    t1.s2 = bld2(r1, s1 + 16u, 0);
}
__device__ __forceinline__ float add_f32(float a, float b) {      // (out of the SLP vectorizer's reach, as hadd)
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void blend_px(const PxTaps &t, v2f wn, v2f ws, float &R, float &G, float &B) {
    const v2f a = pfma(v2f{t.s4.x, t.s4.y}, splat(ws.x), v2f{t.n4.x, t.n4.y} * splat(wn.x));      // (R, G) west halves
    const v2f b = pfma(v2f{t.s4.z, t.s4.w}, ws, v2f{t.n4.z, t.n4.w} * wn);                        // (B west, R east)
    const v2f e = pfma(t.s2, splat(ws.y), t.n2 * splat(wn.y));                                    // (G, B) east halves
    R = add_f32(a.x, b.y);
    G = add_f32(a.y, e.x);
    B = add_f32(b.x, e.y);
}
template <bool VIRT, bool WIDE = false, bool FAST = false>
__device__ __forceinline__ void finish_cell_px(const WarpOut &o, const Cell &c, const PxTaps &t0, const PxTaps &t1, v2f *wl, int r, int lane,
                                               bool col_ok, bool own, unsigned HW, unsigned off) {
    float p[3], q[3];
    blend_px(t0, c.wn0, c.ws0, p[0], p[1], p[2]);
    blend_px(t1, c.wn1, c.ws1, q[0], q[1], q[2]);
    const v2f wv[3] = {v2f{p[0], q[0]}, v2f{p[1], q[1]}, v2f{p[2], q[2]}};
    store_cell<VIRT, WIDE, FAST>(o, c, wv, wl, r, lane, col_ok, own, HW, off);
}

// phase 1 of a tile: every cell of the tile + halo is warped once, rows dealt round-robin to the NW waves; the depth of a wave's next
// row is fetched under the current row's projection
// LATE ROWS (lean forward, Tiling::late — sqd_photo_set_fwd_variant(6); off by default): a 28-row tile warps 34 rows — four rounds of the
// eight waves and two rows more, a fifth round of waves 0 and 1 with the other six waiting at the barrier, in front of a phase 2 in which
// those same two waves own two row pairs and waves 6 and 7 one.  With late rows the rounds that fill all waves run before the barrier; the
// rows left over go to the LAST waves (the ones with a row pair less), which warp them AFTER the barrier while the others are in phase 2,
// and count them done in an LDS word; a row pair whose window reaches a late row (the tile's last pairs; every pair of a tile that
// reflects rows) waits for that count first — by then it has long been reached.  Measured (profiles/r06h): the same bits, the same
// 40.5 us, and still 2 900 cycles of a wave's median at the barrier — the waves of a round do not finish together whatever their row
// counts; the launch follows the CU's VALU issue (phase 2's 126 half-rate shuffles per row), not a wave's critical path.  `mid` = what
// the caller does between the phases (target staging commit, barrier).  Hands back the first late tile row (LATE_NONE: none) and their number.
struct NoMid {
    __device__ __forceinline__ void operator()() const {}
};
constexpr int LATE_NONE = 1 << 20;
template <int NW, bool VIRT, bool WIDE = false, bool FAST = false, bool PX = false, typename Mid = NoMid>
__device__ __forceinline__ void warp_tile(const sqd_photo_args &a, const PairPass &pp, v2f *wl, int b, int y0, int own_rows, int xr, bool col_ok,
                                          bool own_col, int lane, int wave, int *late_ctr = nullptr, int *late_r0 = nullptr, int *late_n = nullptr,
                                          Mid mid = Mid()) {
    const int H = a.H, W = a.W;
    const unsigned HW = (unsigned)(H * W);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const float *__restrict__ dep = a.depth + (size_t)b * HW;
    const unsigned img_bytes = 3u * HW * 4u;
    // (PX: the same bytes per image, laid out [H,W,3])
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s0] + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s1] + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    const bool two = pp.s1 != pp.s0;
    WarpOut o;
    o.smp0 = a.sample[pp.s0] ? reinterpret_cast<float2 *>(a.sample[pp.s0]) + (size_t)b * HW : nullptr;
    o.smp1 = two && a.sample[pp.s1] ? reinterpret_cast<float2 *>(a.sample[pp.s1]) + (size_t)b * HW : nullptr;
    o.tap0 = a.x0y0[pp.s0] ? reinterpret_cast<int2 *>(a.x0y0[pp.s0]) + (size_t)b * HW : nullptr;
    o.tap1 = two && a.x0y0[pp.s1] ? reinterpret_cast<int2 *>(a.x0y0[pp.s1]) + (size_t)b * HW : nullptr;
    o.w0 = a.warped[pp.s0] ? a.warped[pp.s0] + (size_t)b * 3 * HW : nullptr;
    o.w1 = two && a.warped[pp.s0] ? a.warped[pp.s1] + (size_t)b * 3 * HW : nullptr;
    float ik[9];
    v2f P[12];                       // (source 0, source 1) projection matrices
    // (read through the constant address space: wave-uniform, never written by this launch — scalar loads into SGPRs whatever the
    //  compiler can or cannot prove about the output pointers)
    typedef const __attribute__((address_space(4))) float *cfp;
    const cfp ikp = (cfp)(a.inv_K + (size_t)b * 16);
    const cfp p0 = (cfp)(a.P + ((size_t)b * pp.S + pp.s0) * 12), p1 = (cfp)(a.P + ((size_t)b * pp.S + pp.s1) * 12);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ik[i * 3 + j] = ikp[i * 4 + j];
#pragma unroll
    for (int j = 0; j < 12; ++j) P[j] = v2f{p0[j], p1[j]};
    const v2f rW = splat(rcp_refined(wm1)), rH = splat(rcp_refined(hm1));
    const int xc = VIRT ? min(max(xr, 0), W - 1) : xr;      // (lanes beyond a narrow image compute a valid column and store zeros)
    const float fx = (float)xc;
    // rows of the tile + halo that lie inside the image (rows outside are reflections of rows inside the tile): r_lo .. r_hi
    const int r_lo = max(0, 3 - y0), r_hi = min(own_rows + 6, H - y0 + 3);
    // late rows: what is left after the rounds that fill all waves, if the owned rows (whose warped colours the waves store from LDS
    // right after the barrier) are not among them
    int n_late = 0;
    if (FAST && late_ctr) {
        const int n = (r_hi - r_lo) % NW;
        if (n > 0 && r_hi - r_lo > NW && own_rows + 3 <= r_hi - n) n_late = n;
    }
    const int r_main = r_hi - n_late;
    const int r_mine = n_late && wave >= NW - n_late ? r_main + (NW - 1 - wave) : -1;      // the last wave takes the first late row
    auto warp_row = [&](int r, float d) {
        const int yA = y0 - 3 + r;
        const unsigned off = (unsigned)(yA * W + xc);
        Cell c;
        project_cell(c, d, fx, (float)yA, ik, P, rW, rH, wm1, hm1, W);
        if constexpr (PX) {
            PxTaps t0, t1;
            gather_taps_px(c, r0, r1, (unsigned)W * 12u, t0, t1);
            finish_cell_px<VIRT, WIDE, FAST>(o, c, t0, t1, wl, r, lane, col_ok, own_col && r >= 3 && r < own_rows + 3, HW, off);
        } else {
            v2f t0[3][2], t1[3][2];
            gather_taps(c, r0, r1, HW * 4u, (unsigned)W * 4u, t0, t1);
            finish_cell<VIRT, WIDE, FAST>(o, c, t0, t1, wl, r, lane, col_ok, own_col && r >= 3 && r < own_rows + 3, HW, off);
        }
    };
    int r = r_lo + wave;
    float d_next = r < r_main ? ldg(dep, (unsigned)((y0 - 3 + r) * W + xc) * 4u) : 0.f;
    const float d_late = r_mine >= 0 ? ldg(dep, (unsigned)((y0 - 3 + r_mine) * W + xc) * 4u) : 0.f;
    for (; r < r_main; r += NW) {
        const float d = d_next;
        if (r + NW < r_main) d_next = ldg(dep, (unsigned)((y0 - 3 + r + NW) * W + xc) * 4u);
        warp_row(r, d);
    }
    if constexpr (FAST) {
        if (late_ctr && threadIdx.x == 0) *late_ctr = 0;
        mid();
        if (r_mine >= 0) {
            warp_row(r_mine, d_late);
            // (the row's LDS writes and this increment are DS operations of one wave: they execute in order)
            if (lane == 0) __hip_atomic_fetch_add(late_ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (late_r0) {
            *late_r0 = n_late ? r_main : LATE_NONE;
            *late_n = n_late;
        }
    }
}

// phase 1 of the lean forward with TWO rows of a wave in flight: the gathers of the wave's next row leave before the current row is
// blended, so that a workgroup alone on its CU's address path (the other one busy in phase 2: Tiling::skew) keeps it fed with its eight
// waves — with one row in flight eight waves reach 57 % of the path's rate (profiles/r06a: phase 1 takes 14 300 cycles with one workgroup
// per CU, 16 000 with two).  Grid stores before the gathers, warped colours to LDS only (store_warped_wide).
struct WarpRow {
    v2f t0[3][2], t1[3][2];
    PxTaps p0, p1;                   // (pixel-interleaved sources)
    v2f wn0, ws0, wn1, ws1;
};
template <int NW, bool PX = false>
__device__ __forceinline__ void warp_tile_pipe(const sqd_photo_args &a, const PairPass &pp, v2f *wl, int b, int y0, int own_rows, int x, bool own_col, int lane,
                                               int wave) {
    const int H = a.H, W = a.W;
    const unsigned HW = (unsigned)(H * W);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const float *__restrict__ dep = a.depth + (size_t)b * HW;
    const unsigned img_bytes = 3u * HW * 4u;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s0] + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s1] + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    float2 *smp0 = reinterpret_cast<float2 *>(a.sample[pp.s0]) + (size_t)b * HW, *smp1 = reinterpret_cast<float2 *>(a.sample[pp.s1]) + (size_t)b * HW;
    float ik[9];
    v2f P[12];
    typedef const __attribute__((address_space(4))) float *cfp;
    const cfp ikp = (cfp)(a.inv_K + (size_t)b * 16);
    const cfp p0 = (cfp)(a.P + ((size_t)b * pp.S + pp.s0) * 12), p1 = (cfp)(a.P + ((size_t)b * pp.S + pp.s1) * 12);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ik[i * 3 + j] = ikp[i * 4 + j];
#pragma unroll
    for (int j = 0; j < 12; ++j) P[j] = v2f{p0[j], p1[j]};
    const v2f rW = splat(rcp_refined(wm1)), rH = splat(rcp_refined(hm1));
    const float fx = (float)x;
    const unsigned HW4 = HW * 4u, W4 = (unsigned)W * 4u;
    const int r_lo = max(0, 3 - y0), r_hi = min(own_rows + 6, H - y0 + 3);
    auto issue = [&](WarpRow &f, int r, float d) {
        const int yA = y0 - 3 + r;
        const unsigned off = (unsigned)(yA * W + x);
        Cell c;
        project_cell(c, d, fx, (float)yA, ik, P, rW, rH, wm1, hm1, W);
        if (own_col && r >= 3 && r < own_rows + 3) {
            stg2(smp0, off * 8u, c.gx.x, c.gy.x);
            stg2(smp1, off * 8u, c.gx.y, c.gy.y);
        }
        f.wn0 = c.wn0; f.ws0 = c.ws0; f.wn1 = c.wn1; f.ws1 = c.ws1;
        if constexpr (PX) {
            gather_taps_px(c, r0, r1, (unsigned)W * 12u, f.p0, f.p1);
            return;
        }
        const unsigned s0 = c.o0 + W4, s1 = c.o1 + W4;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            f.t0[ch][0] = bld2(r0, c.o0, ch * HW4);
            f.t1[ch][0] = bld2(r1, c.o1, ch * HW4);
            f.t0[ch][1] = bld2(r0, s0, ch * HW4);
            f.t1[ch][1] = bld2(r1, s1, ch * HW4);
        }
    };
    auto finish = [&](const WarpRow &f, int r) {
        if constexpr (PX) {
            float p[3], q[3];
            blend_px(f.p0, f.wn0, f.ws0, p[0], p[1], p[2]);
            blend_px(f.p1, f.wn1, f.ws1, q[0], q[1], q[2]);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) wl[(r * 3 + ch) * 64 + lane] = v2f{p[ch], q[ch]};
            return;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const v2f ea = pfma(f.t0[ch][1], f.ws0, f.t0[ch][0] * f.wn0), eb = pfma(f.t1[ch][1], f.ws1, f.t1[ch][0] * f.wn1);
            wl[(r * 3 + ch) * 64 + lane] = v2f{hadd(ea), hadd(eb)};
        }
    };
    auto depth_of = [&](int r) { return r < r_hi ? ldg(dep, (unsigned)((y0 - 3 + r) * W + x) * 4u) : 0.f; };
    int r = r_lo + wave;
    if (r >= r_hi) return;
    WarpRow A, B;
    float d1 = depth_of(r + NW);
    issue(A, r, depth_of(r));
    for (;;) {
        float d2 = depth_of(r + 2 * NW);
        if (r + NW < r_hi) issue(B, r + NW, d1);
        finish(A, r);
        r += NW;
        if (r >= r_hi) break;
        d1 = depth_of(r + 2 * NW);
        if (r + NW < r_hi) issue(A, r + NW, d2);
        finish(B, r);
        r += NW;
        if (r >= r_hi) break;
    }
}

// ---- wide edition (round 6): every vector-memory instruction of a wave costs the CU's address path 16 clocks whatever it moves (4 lanes
// per clock), and the round-5 kernel issued 43 of them per 64-pixel output row — 30 us of the launch's 51 before any arithmetic, the two
// adding up rather than overlapping.  Two families of 4-byte accesses become 16-byte ones that go through LDS:
//  * the TARGET rows of the tile + halo are staged once, a lane fetching four consecutive pixels of one of four rows (one instruction
//    = 4 rows x 64 columns; 26 per tile instead of phase 2's 15 dword loads per output row), and phase 2 reads them from LDS;
//  * the WARPED colours, which phase 1 has put into the LDS tile anyway, leave from there as 16-byte stores (all 64 columns of the
//    strip: the three halo columns on either side are the neighbouring tile's own cells, warped to the same bits — the two tiles sit
//    on the same XCD and the duplicates merge in its L2).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// the same in two halves, for at most 4 units per wave (tiles of up to 36 rows + halo on 8 waves): the loads leave before phase 1's row loop
// and land in LDS after it — their round trip (4 800 cycles in the first wide build's trace) runs under the warps
struct StagedRows { u32x4 v[4]; };
template <int NW>
__device__ __forceinline__ void stage_target_issue(StagedRows &sr, const float *__restrict__ tgt, int H, int W, int y0, int x0, int r_lo, int r_hi, int lane, int wave) {
    const unsigned HW = (unsigned)(H * W);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tgt), 0, 3u * HW * 4u, 0x00020000);
    const int rsub = lane >> 4, c4 = (lane & 15) * 4;
    const int ngrp = (r_hi - r_lo + 3) >> 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = wave + NW * u;
        const int p = it / ngrp, g = it - p * ngrp;
        const int r = r_lo + 4 * g + rsub;
        const unsigned voff = it < 3 * ngrp && r < r_hi ? (unsigned)((y0 - 3 + r) * W + x0 + c4) * 4u : 0x80000000u;
        sr.v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)p * HW * 4u, 0);
    }
}
template <int NW>
__device__ __forceinline__ void stage_target_commit(const StagedRows &sr, float *tt, int r_lo, int r_hi, int lane, int wave) {
    const int rsub = lane >> 4, c4 = (lane & 15) * 4;
    const int ngrp = (r_hi - r_lo + 3) >> 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = wave + NW * u;
        const int p = it / ngrp, g = it - p * ngrp;
        const int r = r_lo + 4 * g + rsub;
        if (it < 3 * ngrp && r < r_hi) *reinterpret_cast<u32x4 *>(tt + (r * 3 + p) * 64 + c4) = sr.v[u];
    }
}
template <int NW>
__device__ __forceinline__ void stage_target_wide(float *tt, const float *__restrict__ tgt, int H, int W, int y0, int x0, int r_lo, int r_hi, int lane, int wave) {
    const unsigned HW = (unsigned)(H * W);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tgt), 0, 3u * HW * 4u, 0x00020000);
    const int rsub = lane >> 4, c4 = (lane & 15) * 4;
    const int ngrp = (r_hi - r_lo + 3) >> 2;
    for (int it = wave; it < 3 * ngrp; it += NW) {
        const int p = it / ngrp, g = it - p * ngrp;
        const int r = r_lo + 4 * g + rsub;
        const unsigned voff = r < r_hi ? (unsigned)((y0 - 3 + r) * W + x0 + c4) * 4u : 0x80000000u;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)p * HW * 4u, 0);
        if (r < r_hi) *reinterpret_cast<u32x4 *>(tt + (r * 3 + p) * 64 + c4) = v;
    }
}
template <int NW>
__device__ __forceinline__ void store_warped_wide(const v2f *wl, float *__restrict__ w0, float *__restrict__ w1, int H, int W, int y0, int x0, int own_rows,
                                                  int lane, int wave, int k0, int k1) {      // (the wave's iterations k0 .. k1 - 1)
    const unsigned HW = (unsigned)(H * W);
    const int rsub = lane >> 4, c4 = (lane & 15) * 4;
    const int ngrp = (own_rows + 3) >> 2;
    for (int it = wave + NW * k0; it < 3 * ngrp && it < wave + NW * k1; it += NW) {
        const int ch = it / ngrp, g = it - ch * ngrp;
        const int j = 4 * g + rsub;                          // owned row j of the tile = tile row j + 3
        if (j < own_rows) {
            const v4f lo = *reinterpret_cast<const v4f *>(wl + ((j + 3) * 3 + ch) * 64 + c4), hi = *reinterpret_cast<const v4f *>(wl + ((j + 3) * 3 + ch) * 64 + c4 + 2);
            const size_t off = (size_t)ch * HW + (size_t)(y0 + j) * W + x0 + c4;
            typedef float v4f_ua __attribute__((ext_vector_type(4), aligned(4)));
            if (w0) *reinterpret_cast<v4f_ua *>(w0 + off) = v4f{lo.x, lo.z, hi.x, hi.z};
            if (w1) *reinterpret_cast<v4f_ua *>(w1 + off) = v4f{lo.y, lo.w, hi.y, hi.w};
        }
    }
}

// (4 workgroups of 4 waves per CU: the register allocator is held to 128 VGPRs; the kernel needs 118.  NW = 8: two workgroups of 8
//  waves on tiles of up to 32 rows — the same waves per SIMD, 38 instead of 2 x 22 warped rows per 32 output rows)
template <int MODE, int NW, bool WIDE, bool FAST, bool PIPE = false, bool PX = false>
__device__ __forceinline__ void photo_tile_body(const sqd_photo_args &a, const PairPass &pp, const float *__restrict__ noise, const Tiling &tl, v2f *wl, int tile) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, W = a.W, nsx = tl.nsx;
    const unsigned HW = (unsigned)(H * W);
    int tx, b, y0, own_rows;
    tile_of(tl, tile, H, b, tx, y0, own_rows);
    const StripX sx = strip_x(tx, nsx, W);
    const int x = sx.x0 + lane;                                   // this lane's column
    const int xr = sx.virt ? reflect_idx(x, W) : x;               // column it loads (virtual strip: reflected halo)
    const bool col_ok = xr >= 0 && xr < W;
    const bool own_col = x >= sx.own0 && x < sx.own1;
    const float *__restrict__ tgt = a.target + (size_t)b * 3 * HW;

    if (MODE == 1 && tl.skew > 0 && (((blockIdx.x >> 3) >> 5) & 1)) {
        // the two workgroups that start together on a CU are blocks 8 i + x and 8 (i + 32) + x of XCD x (profiles/r06a): the second one
        // waits out `skew` cycles so that its address-bound phase 1 runs beside the first one's issue-bound phase 2 from then on
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)tl.skew) __builtin_amdgcn_s_sleep(32);
    }
    int *late_ctr = nullptr;
    int late_r0 = LATE_NONE, late_n = 0;
    PHOTO_STAMP(0);
    PHOTO_STAMP(7);                                                                 // HW_ID: which CU / SIMD the wave runs on
#ifdef SQD_PHOTO_TRACE
    if (g_photo_trace && lane == 0) g_photo_trace[((size_t)tile * NW + wave) * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));      // XCC_ID
#endif
    if (MODE == 1) {
        // ---------------------------------------------------------------- phase 1: warp every cell of the tile once
        // (images of fewer than 64 columns: lanes of a strip lie outside the image — they compute a clamped column and store zeros)
        if (WIDE) {                                   // (launched for W >= 64 only)
            float *tt = reinterpret_cast<float *>(wl + (tl.TR + 6) * 192);
            const int r_lo = max(0, 3 - y0), r_hi = min(own_rows + 6, H - y0 + 3);
            if constexpr (FAST) {
                StagedRows sr;
                stage_target_issue<NW>(sr, tgt, H, W, y0, sx.x0, r_lo, r_hi, lane, wave);
                PHOTO_STAMP(1);
                if constexpr (PIPE) {
                    warp_tile_pipe<NW, PX>(a, pp, wl, b, y0, own_rows, x, own_col, lane, wave);
                    stage_target_commit<NW>(sr, tt, r_lo, r_hi, lane, wave);
                    PHOTO_STAMP(2);
                    __syncthreads();
                    PHOTO_STAMP(3);
                } else {
                    auto mid = [&]() {
                        stage_target_commit<NW>(sr, tt, r_lo, r_hi, lane, wave);
                        PHOTO_STAMP(2);
                        __syncthreads();
                        PHOTO_STAMP(3);
                    };
                    late_ctr = reinterpret_cast<int *>(wl + (tl.TR + 6) * 288);      // (the word behind the warped + target rows)
                    warp_tile<NW, false, true, true, PX>(a, pp, wl, b, y0, own_rows, xr, col_ok, own_col, lane, wave, tl.late ? late_ctr : nullptr, &late_r0, &late_n, mid);
                }
            } else {
                stage_target_wide<NW>(tt, tgt, H, W, y0, sx.x0, r_lo, r_hi, lane, wave);
                PHOTO_STAMP(1);
                warp_tile<NW, false, true>(a, pp, wl, b, y0, own_rows, xr, col_ok, own_col, lane, wave);
                PHOTO_STAMP(2);
                __syncthreads();
                PHOTO_STAMP(3);
            }
            const bool two = pp.s1 != pp.s0;
            if (!FAST)      // (fast edition: the stores leave between the wave's row pairs — ssim_rows —, under the other waves' arithmetic)
                store_warped_wide<NW>(wl, a.warped[pp.s0] ? a.warped[pp.s0] + (size_t)b * 3 * HW : nullptr,
                                      two && a.warped[pp.s0] ? a.warped[pp.s1] + (size_t)b * 3 * HW : nullptr, H, W, y0, sx.x0, own_rows, lane, wave);
        } else {
            PHOTO_STAMP(1);
            if (W < 64) warp_tile<NW, true>(a, pp, wl, b, y0, own_rows, xr, col_ok, own_col, lane, wave);
            else warp_tile<NW, false>(a, pp, wl, b, y0, own_rows, xr, col_ok, own_col, lane, wave);
            PHOTO_STAMP(2);
            __syncthreads();
            PHOTO_STAMP(3);
        }
        PHOTO_STAMP(4);
    }

    // -------------------------------------------------------------------- phase 2: window statistics, SSIM, selection
    Ctx<MODE> k;
    const unsigned img_bytes = 3u * HW * 4u;
    k.tgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tgt), 0, img_bytes, 0x00020000);
    const float *q0 = MODE == 0 ? a.sources[pp.s0] : MODE == 2 ? a.warped[pp.s0] : a.target;
    const float *q1 = MODE == 0 ? a.sources[pp.s1] : MODE == 2 ? a.warped[pp.s1] : a.target;
    k.p0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(q0 + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    k.p1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(q1 + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    if constexpr (FAST && MODE == 1) {
        k.p0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.identity + (size_t)b * 2 * HW), 0, 2u * HW * 4u, 0x00020000);
        k.selp = a.sel + (size_t)b * HW;
        k.idxp = a.idx + (size_t)b * HW;
        k.w0 = a.warped[pp.s0] + (size_t)b * 3 * HW;
        k.w1 = a.warped[pp.s1] + (size_t)b * 3 * HW;
    }
    k.wl = wl;
    k.tt = MODE == 1 && WIDE ? reinterpret_cast<const float *>(wl + (tl.TR + 6) * 192) : nullptr;
    k.H = H; k.W = W; k.y0 = y0; k.x = x; k.lane = lane; k.HW = HW;
    k.late_ctr = late_ctr; k.late_r0 = late_r0; k.late_n = late_n;
    k.xoff = col_ok ? (unsigned)xr * 4u : 0x80000000u;
    float loss_acc = 0.f;
    if (FAST && y0 >= 3 && y0 + own_rows + 3 <= H) {       // every row of the tile + halo inside the image: the reflection-free row addressing
        if (sx.kind == LEFT)
            ssim_rows<MODE, LEFT, WIDE, FAST, true, PX>(a, pp, noise, k, lane >= 1 && lane <= 3, b, wave, NW, own_rows, own_col, loss_acc);
        else if (sx.kind == RIGHT)
            ssim_rows<MODE, RIGHT, WIDE, FAST, true, PX>(a, pp, noise, k, lane >= 60 && lane <= 62, b, wave, NW, own_rows, own_col, loss_acc);
        else
            ssim_rows<MODE, INTERIOR, WIDE, FAST, true, PX>(a, pp, noise, k, false, b, wave, NW, own_rows, own_col, loss_acc);
    } else if (sx.kind == LEFT)
        ssim_rows<MODE, LEFT, WIDE, FAST, false, PX>(a, pp, noise, k, lane >= 1 && lane <= 3, b, wave, NW, own_rows, own_col, loss_acc);
    else if (sx.kind == RIGHT)
        ssim_rows<MODE, RIGHT, WIDE, FAST, false, PX>(a, pp, noise, k, lane >= 60 && lane <= 62, b, wave, NW, own_rows, own_col, loss_acc);
    else
        ssim_rows<MODE, INTERIOR, WIDE, FAST, false, PX>(a, pp, noise, k, false, b, wave, NW, own_rows, own_col, loss_acc);
    PHOTO_STAMP(5);
    if (MODE == 1 && a.loss_part && pp.last) {       // (8-wave tilings: 16 partials per tile — the stream kernel's one per row pair)
        loss_acc = wave_sum(loss_acc);
        if (lane == 0) {
            a.loss_part[tile * (NW == 8 ? 16 : NW) + wave] = loss_acc;
            if (NW == 8) a.loss_part[tile * 16 + 8 + wave] = 0.f;
        }
    }
}
template <int MODE, int NW = 4, bool WIDE = false, bool FAST = false, bool PIPE = false, bool PX = false>
__global__ __launch_bounds__(NW * 64, 16 / NW) void photo_tile_kernel(sqd_photo_args a, PairPass pp, const float *__restrict__ noise, Tiling tl) {
    extern __shared__ v2f wl[];                       // MODE 1: [TR + 6][3][64] (source 0, source 1) warped colours (+ WIDE: [TR + 6][3][64] target)
    // consecutive tiles (which share halo rows / columns) on the same XCD: workgroup i runs on XCD i % 8
    const int tile = (blockIdx.x & 7) * tl.nblk8 + (blockIdx.x >> 3);
    if (tile >= tl.ntiles) return;
    photo_tile_body<MODE, NW, WIDE, FAST, PIPE, PX>(a, pp, noise, tl, wl, tile);
}
// the lean forward as RESIDENT workgroups (two per CU = 64 per XCD), each walking its XCD's tile list with stride 64: the per-CU traces of
// round 6 show a freed workgroup slot idle for ~5 000 cycles (2 us of a 45 us launch) before its successor starts
template <int NW>
__global__ __launch_bounds__(NW * 64, 16 / NW) void photo_tile_resident_kernel(sqd_photo_args a, PairPass pp, Tiling tl, int per_xcd) {
    extern __shared__ v2f wl[];
    // (the launch arguments are re-read per tile through a pointer the compiler cannot see through: with `a` inlined the tile loop kept
    //  every tile-independent address live across iterations — 122 registers and 92 SGPR spills, 49.0 against 44.7 us)
    typedef const __attribute__((address_space(4))) sqd_photo_args *argp_t;
    argp_t ap = (argp_t)__builtin_amdgcn_kernarg_segment_ptr();
    for (int j = blockIdx.x >> 3; j < tl.nblk8; j += per_xcd) {
        const int tile = (blockIdx.x & 7) * tl.nblk8 + j;
        asm volatile("" : "+s"(ap));
        if (tile < tl.ntiles) photo_tile_body<1, NW, true, true>(*(const sqd_photo_args *)ap, pp, nullptr, tl, wl, tile);
        __syncthreads();
    }
}

// =====================================================================================================================
// Round 6: the fused forward, COLOUR-SERIAL phase 2 (photo_fwd_c_kernel).  Same tiles, same phase 1, same arithmetic in the same
// order as photo_tile_kernel<1, 8> (the two kernels' outputs are equal bit for bit: tests/test_gpu_photometric.py) — what changes
// is what a wave holds at a time.  SSIM is separable over the colour planes until the final mean, so phase 2 walks the three
// colours one after the other: one colour's window sums are 8 registers (St, Stt, and (source 0, source 1) pairs of Sw, Sq, Swt)
// instead of 24, its eight window rows 24 registers loaded as ONE batch (8 LDS reads + 8 target loads in flight), the horizontal
// pass 7 shuffle chains (4 + 3 interleaved), and only the running (ssim, l1) sums of the row pair's two outputs (8 registers)
// survive a colour.  The kernel fits 64 VGPRs: EIGHT waves per SIMD = four 8-wave workgroups per CU (4 x 34 KB of LDS), i.e. all
// 1024 tiles of configs[1] resident at once — the records of rounds 3-5 say the forward follows its resident waves (3 waves per
// SIMD 65.8 us, 4 waves 51 us) and that both of its phases are per-wave dependent chains, not pipe-bound.
// phase 1 of the colour-serial kernel: warp_tile with the stores that do not depend on the gathers (sampling grid, integer taps)
// issued BEFORE the gathers — eight registers less across the tap round trip
template <int NW, bool VIRT>
__device__ __forceinline__ void warp_tile_c(const sqd_photo_args &a, const PairPass &pp, v2f *wl, int b, int y0, int own_rows, int xr, bool col_ok,
                                            bool own_col, int lane, int wave) {
    const int H = a.H, W = a.W;
    const unsigned HW = (unsigned)(H * W);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    const float *__restrict__ dep = a.depth + (size_t)b * HW;
    const unsigned img_bytes = 3u * HW * 4u;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s0] + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s1] + (size_t)b * 3 * HW), 0, img_bytes, 0x00020000);
    const bool two = pp.s1 != pp.s0;
    WarpOut o;
    o.smp0 = a.sample[pp.s0] ? reinterpret_cast<float2 *>(a.sample[pp.s0]) + (size_t)b * HW : nullptr;
    o.smp1 = two && a.sample[pp.s1] ? reinterpret_cast<float2 *>(a.sample[pp.s1]) + (size_t)b * HW : nullptr;
    o.tap0 = a.x0y0[pp.s0] ? reinterpret_cast<int2 *>(a.x0y0[pp.s0]) + (size_t)b * HW : nullptr;
    o.tap1 = two && a.x0y0[pp.s1] ? reinterpret_cast<int2 *>(a.x0y0[pp.s1]) + (size_t)b * HW : nullptr;
    o.w0 = a.warped[pp.s0] ? a.warped[pp.s0] + (size_t)b * 3 * HW : nullptr;
    o.w1 = two && a.warped[pp.s0] ? a.warped[pp.s1] + (size_t)b * 3 * HW : nullptr;
    float ik[9];
    v2f P[12];
    typedef const __attribute__((address_space(4))) float *cfp;
    const cfp ikp = (cfp)(a.inv_K + (size_t)b * 16);
    const cfp p0 = (cfp)(a.P + ((size_t)b * pp.S + pp.s0) * 12), p1 = (cfp)(a.P + ((size_t)b * pp.S + pp.s1) * 12);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ik[i * 3 + j] = ikp[i * 4 + j];
#pragma unroll
    for (int j = 0; j < 12; ++j) P[j] = v2f{p0[j], p1[j]};
    const v2f rW = splat(rcp_refined(wm1)), rH = splat(rcp_refined(hm1));
    const int xc = VIRT ? min(max(xr, 0), W - 1) : xr;
    const float fx = (float)xc;
    const int r_lo = max(0, 3 - y0), r_hi = min(own_rows + 6, H - y0 + 3);
    int r = r_lo + wave;
    float d_next = r < r_hi ? ldg(dep, (unsigned)((y0 - 3 + r) * W + xc) * 4u) : 0.f;
    for (; r < r_hi; r += NW) {
        const int yA = y0 - 3 + r;
        const unsigned off = (unsigned)(yA * W + xc);
        const float d = d_next;
        if (r + NW < r_hi) d_next = ldg(dep, (off + (unsigned)(NW * W)) * 4u);
        const bool own = own_col && r >= 3 && r < own_rows + 3;
        v2f wn0, ws0, wn1, ws1;
        unsigned o0, o1;
        {
            Cell c;
            project_cell(c, d, fx, (float)yA, ik, P, rW, rH, wm1, hm1, W);
            if (own) {
                if (o.smp0) stg2(o.smp0, off * 8u, c.gx.x, c.gy.x);
                if (o.smp1) stg2(o.smp1, off * 8u, c.gx.y, c.gy.y);
                if (o.tap0) stg2i(o.tap0, off * 8u, c.x00, c.y00);
                if (o.tap1) stg2i(o.tap1, off * 8u, c.x01, c.y01);
            }
            wn0 = c.wn0; ws0 = c.ws0; wn1 = c.wn1; ws1 = c.ws1; o0 = c.o0; o1 = c.o1;
        }
        const unsigned HW4 = HW * 4u, s0 = o0 + (unsigned)W * 4u, s1 = o1 + (unsigned)W * 4u;
        v2f t0[3][2], t1[3][2];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            t0[ch][0] = bld2(r0, o0, ch * HW4);
            t1[ch][0] = bld2(r1, o1, ch * HW4);
            t0[ch][1] = bld2(r0, s0, ch * HW4);
            t1[ch][1] = bld2(r1, s1, ch * HW4);
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const v2f ea = pfma(t0[ch][1], ws0, t0[ch][0] * wn0), eb = pfma(t1[ch][1], ws1, t1[ch][0] * wn1);
            const v2f wv = v2f{hadd(ea), hadd(eb)};
            wl[(r * 3 + ch) * 64 + lane] = (!VIRT || col_ok) ? wv : splat(0.f);
            if (own) {
                if (o.w0) stg(o.w0, (off + ch * HW) * 4u, wv.x);
                if (o.w1) stg(o.w1, (off + ch * HW) * 4u, wv.y);
            }
        }
    }
}

struct CSum {
    float St, Stt;
    v2f Sw, Sq, Swt;
};
// centred 7-tap sums of FOUR quantities across lanes (box7x3's chains, four interleaved)
template <int KIND>
__device__ __forceinline__ void box7x4(float &a, float &b, float &c, float &d, bool edge) {
    float ea = 0.f, eb = 0.f, ec = 0.f, ed = 0.f;
    if (KIND == LEFT) {
        const float ma = edge ? a : 0.f, mb = edge ? b : 0.f, mc = edge ? c : 0.f, md = edge ? d : 0.f;
        ea = quad_reverse<0, 0>((ma + row_shr<1>(ma)) + row_shr<2>(ma));
        eb = quad_reverse<0, 0>((mb + row_shr<1>(mb)) + row_shr<2>(mb));
        ec = quad_reverse<0, 0>((mc + row_shr<1>(mc)) + row_shr<2>(mc));
        ed = quad_reverse<0, 0>((md + row_shr<1>(md)) + row_shr<2>(md));
    } else if (KIND == RIGHT) {
        const float ma = edge ? a : 0.f, mb = edge ? b : 0.f, mc = edge ? c : 0.f, md = edge ? d : 0.f;
        ea = quad_reverse<3, 3>((ma + row_shl<1>(ma)) + row_shl<2>(ma));
        eb = quad_reverse<3, 3>((mb + row_shl<1>(mb)) + row_shl<2>(mb));
        ec = quad_reverse<3, 3>((mc + row_shl<1>(mc)) + row_shl<2>(mc));
        ed = quad_reverse<3, 3>((md + row_shl<1>(md)) + row_shl<2>(md));
    }
    float ra = a + wave_shr1(a), rb = b + wave_shr1(b), rc = c + wave_shr1(c), rd = d + wave_shr1(d);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc); rd = d + wave_shr1(rd);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc); rd = d + wave_shr1(rd);
    float ua = a + wave_shl1(a), ub = b + wave_shl1(b), uc = c + wave_shl1(c), ud = d + wave_shl1(d);
    ua = a + wave_shl1(ua); ub = b + wave_shl1(ub); uc = c + wave_shl1(uc); ud = d + wave_shl1(ud);
    a = ra + wave_shl1(ua); b = rb + wave_shl1(ub); c = rc + wave_shl1(uc); d = rd + wave_shl1(ud);
    if (KIND != INTERIOR) {
        a += ea; b += eb; c += ec; d += ed;
    }
}
// one colour's term of ssim_l1_fwd (same instructions, same order)
__device__ __forceinline__ void ssim_colour(float St, v2f Sw, v2f Sq, v2f Swt, float tc, v2f wc, v2f &ssim_sum, v2f &l1) {
    constexpr float K1 = C1 * 2401.f, K2 = C2 * 2401.f;
    const v2f p = Sw * splat(St);
    const v2f A1 = pfma(splat(2.f), p, splat(K1));
    const v2f A2 = pfma(splat(2.f), pfma(splat(49.f), Swt, -p), splat(K2));
    const v2f q = pfma(Sw, Sw, splat(St * St));
    const v2f B1 = q + splat(K1), B2 = pfma(splat(49.f), Sq, splat(K2) - q);
    const v2f num = A1 * A2, den = B1 * B2;
    const v2f rd = v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const v2f q0 = num * rd;
    const v2f Sv = pfma(pfma(-den, q0, num), rd, q0);
    const v2f r = pfma(splat(-0.5f), Sv, splat(0.5f));
    ssim_sum += v2f{__builtin_amdgcn_fmed3f(r.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(r.y, 0.f, 1.f)};      // torch.clamp(., 0, 1)
    const v2f df = splat(tc) - wc;
    l1 += v2f{fabsf(df.x), fabsf(df.y)};
}
// S = core + one more row, Sigma t^2 folded into Sigma w^2, the horizontal pass, the colour's SSIM / L1 terms
template <int KIND>
__device__ __forceinline__ void finish_colour(const CSum &core, float t, v2f w, float tc, v2f wc, bool edge, v2f &ssim_sum, v2f &l1) {
    float St = core.St + t;
    const float Stt = fmaf(t, t, core.Stt);
    v2f Sw = core.Sw + w;
    v2f Sq = pfma(w, w, core.Sq);
    v2f Swt = pfma(w, splat(t), core.Swt);
    Sq += splat(Stt);
    float swx = Sw.x, sqx = Sq.x, stx = Swt.x, swy = Sw.y, sqy = Sq.y, sty = Swt.y;
    box7x4<KIND>(St, swx, sqx, stx, edge);
    box7x3<KIND>(swy, sqy, sty, edge);
    ssim_colour(St, v2f{swx, swy}, v2f{sqx, sqy}, v2f{stx, sty}, tc, wc, ssim_sum, l1);
}

template <int KIND>
__device__ __forceinline__ void ssim_rows_c(const sqd_photo_args &a, const PairPass &pp, const Ctx<1> &k, bool edge, int b, int wave, int nwaves,
                                            int own_rows, bool own_col, float &loss_acc) {
    const int flags = a.loss_flags;
    for (int p = wave; 2 * p < own_rows; p += nwaves) {
        const int j = 2 * p;                         // tile rows j .. j+7 feed the outputs y0+j (centre j+3) and y0+j+1 (centre j+4)
        v2f ss0 = splat(0.f), ss1 = splat(0.f), la0 = splat(0.f), la1 = splat(0.f);
        // rows of the window in the image / in the LDS tile (wave-uniform; ReflectionPad2d(3): layers.py:26)
        unsigned trow[8];
        int lrow[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int yr = reflect_idx(k.y0 - 3 + j + i, k.H);
            trow[i] = (unsigned)(yr * k.W) * 4u;
            lrow[i] = (yr - (k.y0 - 3)) * 3 * 64;
        }
#pragma nounroll
        for (int c = 0; c < 3; ++c) {
            float t[8];
            v2f w[8];
            const unsigned plane = (unsigned)c * k.HW * 4u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[i] = bld(k.tgt, k.xoff, trow[i] + plane);
                w[i] = k.wl[lrow[i] + c * 64 + k.lane];
            }
            CSum core;                               // rows j+1 .. j+6: shared by both outputs
            core.St = 0.f; core.Stt = 0.f; core.Sw = core.Sq = core.Swt = splat(0.f);
#pragma unroll
            for (int i = 1; i < 7; ++i) {
                core.St += t[i];
                core.Stt = fmaf(t[i], t[i], core.Stt);
                core.Sw += w[i];
                core.Sq = pfma(w[i], w[i], core.Sq);
                core.Swt = pfma(w[i], splat(t[i]), core.Swt);
            }
            finish_colour<KIND>(core, t[0], w[0], t[3], w[3], edge, ss0, la0);
            finish_colour<KIND>(core, t[7], w[7], t[4], w[4], edge, ss1, la1);
        }
        const unsigned HW = k.HW;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const v2f ssim_sum = o ? ss1 : ss0, l1 = o ? la1 : la0;
            v2f loss = splat(0.85f) * (ssim_sum * splat(1.f / 3.f)) + splat(0.15f) * (l1 * splat(1.f / 3.f));
            if (flags & SQD_LOSS_NO_SSIM) loss = l1 * splat(1.f / 3.f);
            if (own_col && j + o < own_rows) select_store(a, pp, loss, b, (unsigned)((k.y0 + j + o) * k.W + k.x), HW, loss_acc);
        }
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 8) void photo_fwd_c_kernel(sqd_photo_args a, PairPass pp, Tiling tl) {
    extern __shared__ v2f wl[];                       // [TR + 6][3][64] (source 0, source 1) warped colours
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = (blockIdx.x & 7) * tl.nblk8 + (blockIdx.x >> 3);      // consecutive tiles (shared halo rows / columns) on one XCD
    if (tile >= tl.ntiles) return;
    const int H = a.H, W = a.W, nsx = tl.nsx;
    const unsigned HW = (unsigned)(H * W);
    int tx, b, y0, own_rows;
    tile_of(tl, tile, H, b, tx, y0, own_rows);
    const StripX sx = strip_x(tx, nsx, W);
    const int x = sx.x0 + lane;
    const int xr = sx.virt ? reflect_idx(x, W) : x;
    const bool col_ok = xr >= 0 && xr < W;
    const bool own_col = x >= sx.own0 && x < sx.own1;
    if (W < 64) warp_tile_c<NW, true>(a, pp, wl, b, y0, own_rows, xr, col_ok, own_col, lane, wave);
    else warp_tile_c<NW, false>(a, pp, wl, b, y0, own_rows, xr, col_ok, own_col, lane, wave);
    __syncthreads();
    Ctx<1> k;
    k.tgt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.target + (size_t)b * 3 * HW), 0, 3u * HW * 4u, 0x00020000);
    k.p0 = k.p1 = k.tgt;
    k.wl = wl;
    k.tt = nullptr;
    k.H = H; k.W = W; k.y0 = y0; k.x = x; k.lane = lane; k.HW = HW;
    k.xoff = col_ok ? (unsigned)xr * 4u : 0x80000000u;
    float loss_acc = 0.f;
    if (sx.kind == LEFT)
        ssim_rows_c<LEFT>(a, pp, k, lane >= 1 && lane <= 3, b, wave, NW, own_rows, own_col, loss_acc);
    else if (sx.kind == RIGHT)
        ssim_rows_c<RIGHT>(a, pp, k, lane >= 60 && lane <= 62, b, wave, NW, own_rows, own_col, loss_acc);
    else
        ssim_rows_c<INTERIOR>(a, pp, k, false, b, wave, NW, own_rows, own_col, loss_acc);
    if (a.loss_part && pp.last) {
        loss_acc = wave_sum(loss_acc);
        if (lane == 0) {
            a.loss_part[tile * 16 + wave] = loss_acc;
            a.loss_part[tile * 16 + 8 + wave] = 0.f;
        }
    }
}

// =====================================================================================================================
// Round 6: the fused forward with DYNAMIC WAVE ROLES and a row in flight under every SSIM step (photo_fwd_s_kernel).
//
// What the per-wave s_memtime traces of round 6 say about photo_tile_kernel<1, 8> (profiles/r06a_photo_fwd_phase_trace.md): a wave alone
// needs 3 400 cycles per warped row — 209 instructions, 950 cycles of issue, the rest is the tap round trip it stands at — and 9 600 per
// row pair; two resident workgroups finish a tile every 25 000 cycles, three every 22 000, one every 38 000: the launch follows the
// waves in flight, the VALU is busy 42 % and the address path 60 %, and the barrier between the phases plus 34 rows / 13 pairs dealt to
// 8 waves leave 15..25 % of the wave slots idle.  Neither more resident waves (colour-serial, 6 per SIMD: 50.7 us) nor fewer memory
// instructions (wide edition, 43 -> 25 per output row: 50.4 us) moves it while a wave still waits out every row's gathers.
// Here the tile's work is three LDS queues of UNITS that the eight waves claim in order whenever a unit's inputs are ready (per-unit
// done flags + a done prefix per queue; a claim is a compare-and-swap AFTER the readiness test, so no wave holds a unit it cannot run):
//   stage: 4 rows of target + depth into LDS, 16 bytes per lane (phase 2 reads no memory; depth waits in the row's own warped slot);
//   row:   warp one row of cells (phase 1's loop body) from the staged depth -> warped colours into LDS, grid + colours to HBM;
//   pair:  two output rows, colour by colour (one colour's window sums are 8 registers) — and BETWEEN the colours of its pair the wave
//          claims a row, projects it and issues its twelve gathers, runs the next colour's ~500 instructions while they fly, then blends:
//          the tap round trip of a row is covered by its own wave's SSIM arithmetic, not by hoping for another wave.
// No workgroup barrier after the prologue.  The loss partial is one slot per PAIR (16 per tile): the sum order does not depend on which
// wave ran what, results are run-to-run identical.
constexpr int ROW_WL = 3 * 64;                 // v2f per LDS row
struct DynCtl {                                // (volatile ints in LDS)
    int stage_next, stage_done, row_next, row_done, pair_next, pair_done, pad0, pad1;
    int stage_flag[16], row_flag[48], pair_flag[24];
};
__device__ __forceinline__ int ldu(const volatile int *p) { return __builtin_amdgcn_readfirstlane(*p); }
// claim unit `expect` of a queue (all lanes call; true for the whole wave if this wave got it)
__device__ __forceinline__ bool claim_unit(volatile int *next, int expect, int lane) {
    int got = expect + 1;
    if (lane == 0) got = atomicCAS(const_cast<int *>(next), expect, expect + 1);
    return __builtin_amdgcn_readfirstlane(got) == expect;
}
// unit i of a queue is complete: set its flag, then push the queue's done prefix over every leading complete unit (whoever finishes a unit
// re-reads the flags after writing its own, so the prefix never stops short of a complete run)
__device__ __forceinline__ void unit_done(volatile int *flag, volatile int *prefix, int i, int n, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
        flag[i] = 1;
        for (;;) {
            const int p = *prefix;
            if (p < n && flag[p]) atomicCAS(const_cast<int *>(prefix), p, p + 1);
            else break;
        }
    }
    asm volatile("" ::: "memory");
}

struct StreamTile {
    int b, y0, own_rows, x0, x, lane;
    int r_lo, r_hi, npairs, nstage, nrows;
    bool own_col;
    unsigned HW;
};

typedef const __attribute__((address_space(4))) float *cfp;
typedef const __attribute__((address_space(4))) sqd_photo_args *argp;
// The launch arguments are re-read from the kernel-argument segment (scalar loads) by every unit that needs them, through a pointer the
// compiler cannot see through: the ~60 SGPRs of pointers a row unit uses would otherwise be loaded in the prologue, stay live through
// the pair units and spill (first build: 287 SGPR + 44 VGPR spills).
__device__ __forceinline__ const sqd_photo_args &fresh_args(argp ap) {
    asm volatile("" : "+s"(ap));
    return *(const sqd_photo_args *)ap;
}

// stage unit g: tile rows r_lo + 4 g .. + 3 (target planes + depth), one 16-byte load per lane and plane
__device__ __forceinline__ void stage_unit(argp ap, const StreamTile &t, v2f *wl, float *tt, int g) {
    const sqd_photo_args &a = fresh_args(ap);
    const int W = a.W;
    const int rsub = t.lane >> 4, c4 = (t.lane & 15) * 4;
    const int r = t.r_lo + 4 * g + rsub;
    const bool ok = r < t.r_hi;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.target + (size_t)t.b * 3 * t.HW), 0, 3u * t.HW * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.depth + (size_t)t.b * t.HW), 0, t.HW * 4u, 0x00020000);
    const unsigned voff = ok ? (unsigned)((t.y0 - 3 + r) * W + t.x0 + c4) * 4u : 0x80000000u;
    u32x4 v[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) v[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)p * t.HW * 4u, 0);
    const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rd, voff, 0, 0);
    if (ok) {
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4 *>(tt + (r * 3 + p) * 64 + c4) = v[p];
        unsigned *dw = reinterpret_cast<unsigned *>(wl + r * ROW_WL + c4);       // the depth of column c waits in .x of the row's first colour slot
        dw[0] = d.x; dw[2] = d.y; dw[4] = d.z; dw[6] = d.w;
    }
}

struct RowInFlight {                 // a warped row between the issue of its gathers and its blend
    v2f t0[3][2], t1[3][2];
    v2f wn0, ws0, wn1, ws1;
    unsigned off;
    int r;
    bool own;
};

// row unit, first half: projection, grid / tap stores, the twelve gathers (phase 1's loop body up to the tap round trip)
__device__ __forceinline__ void row_issue(argp ap, const PairPass &pp, const StreamTile &t, const v2f *wl, int r, RowInFlight &f) {
    const sqd_photo_args &a = fresh_args(ap);
    const int W = a.W, H = a.H;
    const unsigned HW = t.HW;
    const int yA = t.y0 - 3 + r;
    f.r = r;
    f.off = (unsigned)(yA * W + t.x);
    const float d = reinterpret_cast<const float *>(wl + r * ROW_WL + t.lane)[0];
    f.own = t.own_col && r >= 3 && r < t.own_rows + 3;
    const bool two = pp.s1 != pp.s0;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    unsigned o0, o1;
    {
        const cfp ikp = (cfp)(a.inv_K + (size_t)t.b * 16);
        const cfp p0 = (cfp)(a.P + ((size_t)t.b * pp.S + pp.s0) * 12), p1 = (cfp)(a.P + ((size_t)t.b * pp.S + pp.s1) * 12);
        float ik[9];
        v2f P[12];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) ik[i * 3 + jj] = ikp[i * 4 + jj];
#pragma unroll
        for (int jj = 0; jj < 12; ++jj) P[jj] = v2f{p0[jj], p1[jj]};
        const v2f rW = splat(rcp_refined(wm1)), rH = splat(rcp_refined(hm1));
        Cell c;
        project_cell(c, d, (float)t.x, (float)yA, ik, P, rW, rH, wm1, hm1, W);
        if (f.own) {
            if (a.sample[pp.s0]) stg2(reinterpret_cast<float2 *>(a.sample[pp.s0]) + (size_t)t.b * HW, f.off * 8u, c.gx.x, c.gy.x);
            if (two && a.sample[pp.s1]) stg2(reinterpret_cast<float2 *>(a.sample[pp.s1]) + (size_t)t.b * HW, f.off * 8u, c.gx.y, c.gy.y);
            if (a.x0y0[pp.s0]) stg2i(reinterpret_cast<int2 *>(a.x0y0[pp.s0]) + (size_t)t.b * HW, f.off * 8u, c.x00, c.y00);
            if (two && a.x0y0[pp.s1]) stg2i(reinterpret_cast<int2 *>(a.x0y0[pp.s1]) + (size_t)t.b * HW, f.off * 8u, c.x01, c.y01);
        }
        f.wn0 = c.wn0; f.ws0 = c.ws0; f.wn1 = c.wn1; f.ws1 = c.ws1; o0 = c.o0; o1 = c.o1;
    }
    const unsigned img_bytes = 3u * HW * 4u;
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s0] + (size_t)t.b * 3 * HW), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.sources[pp.s1] + (size_t)t.b * 3 * HW), 0, img_bytes, 0x00020000);
    const unsigned HW4 = HW * 4u, s0 = o0 + (unsigned)W * 4u, s1 = o1 + (unsigned)W * 4u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        f.t0[ch][0] = bld2(r0, o0, ch * HW4);
        f.t1[ch][0] = bld2(r1, o1, ch * HW4);
        f.t0[ch][1] = bld2(r0, s0, ch * HW4);
        f.t1[ch][1] = bld2(r1, s1, ch * HW4);
    }
}
// second half: blend, LDS row, warped colours to HBM
__device__ __forceinline__ void row_finish(argp ap, const PairPass &pp, const StreamTile &t, v2f *wl, const RowInFlight &f) {
    const sqd_photo_args &a = fresh_args(ap);
    v2f *row = wl + f.r * ROW_WL;
    float *w0 = a.warped[pp.s0] ? a.warped[pp.s0] + (size_t)t.b * 3 * t.HW : nullptr;
    float *w1 = pp.s1 != pp.s0 && a.warped[pp.s0] ? a.warped[pp.s1] + (size_t)t.b * 3 * t.HW : nullptr;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const v2f ea = pfma(f.t0[ch][1], f.ws0, f.t0[ch][0] * f.wn0), eb = pfma(f.t1[ch][1], f.ws1, f.t1[ch][0] * f.wn1);
        const v2f wv = v2f{hadd(ea), hadd(eb)};
        row[ch * 64 + t.lane] = wv;
        if (f.own) {
            if (w0) stg(w0, (f.off + ch * t.HW) * 4u, wv.x);
            if (w1) stg(w1, (f.off + ch * t.HW) * 4u, wv.y);
        }
    }
}

// the next row of the queue, if its depth is staged: claimed (true) or not
__device__ __forceinline__ bool claim_row(volatile DynCtl *q, const StreamTile &t, int &r) {
    r = ldu(&q->row_next);
    if (r < t.nrows && r < 4 * ldu(&q->stage_done)) return claim_unit(&q->row_next, r, t.lane);
    return false;
}

// pair unit p: output rows y0 + 2 p, + 1 (ssim_rows_c's loop body), with a row unit in flight under every colour
template <int KIND>
__device__ __forceinline__ void pair_unit(argp ap, const PairPass &pp, const StreamTile &t, v2f *wl, const float *tt,
                                          volatile DynCtl *q, bool edge, int tile, int p, int H) {
    const int j = 2 * p;
    v2f ss0 = splat(0.f), ss1 = splat(0.f), la0 = splat(0.f), la1 = splat(0.f);
    int slot[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) slot[i] = (reflect_idx(t.y0 - 3 + j + i, H) - (t.y0 - 3)) * ROW_WL;
#pragma nounroll
    for (int c = 0; c < 3; ++c) {
        RowInFlight f;
        int r;
        const bool have = claim_row(q, t, r);
        if (have) row_issue(ap, pp, t, wl, t.r_lo + r, f);
        {
            float tv[8];
            v2f w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                tv[i] = tt[slot[i] + c * 64 + t.lane];
                w[i] = wl[slot[i] + c * 64 + t.lane];
            }
            CSum core;
            core.St = 0.f; core.Stt = 0.f; core.Sw = core.Sq = core.Swt = splat(0.f);
#pragma unroll
            for (int i = 1; i < 7; ++i) {
                core.St += tv[i];
                core.Stt = fmaf(tv[i], tv[i], core.Stt);
                core.Sw += w[i];
                core.Sq = pfma(w[i], w[i], core.Sq);
                core.Swt = pfma(w[i], splat(tv[i]), core.Swt);
            }
            __builtin_amdgcn_sched_barrier(0);
            finish_colour<KIND>(core, tv[0], w[0], tv[3], w[3], edge, ss0, la0);
            __builtin_amdgcn_sched_barrier(0);
            finish_colour<KIND>(core, tv[7], w[7], tv[4], w[4], edge, ss1, la1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (have) {
            row_finish(ap, pp, t, wl, f);
            unit_done(q->row_flag, &q->row_done, r, t.nrows, t.lane);
        }
    }
    const sqd_photo_args &a = fresh_args(ap);
    const int W = a.W, flags = a.loss_flags;
    float loss_acc = 0.f;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const v2f ssim_sum = o ? ss1 : ss0, l1 = o ? la1 : la0;
        v2f loss = splat(0.85f) * (ssim_sum * splat(1.f / 3.f)) + splat(0.15f) * (l1 * splat(1.f / 3.f));
        if (flags & SQD_LOSS_NO_SSIM) loss = l1 * splat(1.f / 3.f);
        if (t.own_col && j + o < t.own_rows) select_store(a, pp, loss, t.b, (unsigned)((t.y0 + j + o) * W + t.x), t.HW, loss_acc);
    }
    if (a.loss_part && pp.last) {
        loss_acc = wave_sum_to_lane63(loss_acc);
        if (t.lane == 63) a.loss_part[tile * 16 + p] = loss_acc;
    }
}

#ifdef SQD_PHOTO_TRACE      // unit log of the stream kernel: per tile 128 records of (type << 8 | index, wave, start, end), slot 0 = the record count
#define STREAM_T0() const unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define STREAM_T1(type, idx)                                                                          \
    do {                                                                                              \
        if (g_photo_trace && lane == 0) {                                                             \
            unsigned long long *tb = g_photo_trace + (size_t)tile * 512;                               \
            const int n = (int)atomicAdd(reinterpret_cast<unsigned long long *>(tb), 1ull) + 1;       \
            if (n < 128) {                                                                            \
                tb[n * 4 + 0] = ((type) << 8) | (idx);                                                \
                tb[n * 4 + 1] = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                     \
                tb[n * 4 + 2] = t0_;                                                                  \
                tb[n * 4 + 3] = __builtin_amdgcn_s_memtime();                                         \
            }                                                                                         \
        }                                                                                             \
    } while (0)
#else
#define STREAM_T0() do { } while (0)
#define STREAM_T1(type, idx) do { } while (0)
#endif
template <int NW, int KIND>
__device__ __forceinline__ void stream_tile(argp ap, const PairPass &pp, const StreamTile &t, v2f *wl, float *tt, volatile DynCtl *q, bool edge,
                                            int tile, int H) {
    const int lane = t.lane;
    for (;;) {
        asm volatile("" ::: "memory");
        // ---- a stage unit
        const int g = ldu(&q->stage_next);
        if (g < t.nstage) {
            if (claim_unit(&q->stage_next, g, lane)) {
                STREAM_T0();
                stage_unit(ap, t, wl, tt, g);
                unit_done(q->stage_flag, &q->stage_done, g, t.nstage, lane);
                STREAM_T1(0, g);
            }
            continue;
        }
        // ---- a pair whose eight rows are warped
        const int p = ldu(&q->pair_next);
        if (p < t.npairs && ldu(&q->row_done) >= min(2 * p + 8 - t.r_lo, t.nrows)) {
            if (claim_unit(&q->pair_next, p, lane)) {
                STREAM_T0();
                pair_unit<KIND>(ap, pp, t, wl, tt, q, edge, tile, p, H);
                unit_done(q->pair_flag, &q->pair_done, p, t.npairs, lane);
                STREAM_T1(2, p);
            }
            continue;
        }
        // ---- a row whose depth is staged (no pair ready: nothing to run under its gathers)
        int r;
        if (claim_row(q, t, r)) {
            STREAM_T0();
            RowInFlight f;
            row_issue(ap, pp, t, wl, t.r_lo + r, f);
            row_finish(ap, pp, t, wl, f);
            unit_done(q->row_flag, &q->row_done, r, t.nrows, lane);
            STREAM_T1(1, r);
            continue;
        }
        if (p >= t.npairs) break;
        __builtin_amdgcn_s_sleep(2);
    }
}

template <int NW, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void photo_fwd_s_kernel(sqd_photo_args a, PairPass pp, Tiling tl) {
    extern __shared__ v2f wl[];                       // [TR + 6][3][64] warped (source 0, source 1) | [TR + 6][3][64] target | DynCtl
    float *tt = reinterpret_cast<float *>(wl + (tl.TR + 6) * ROW_WL);
    volatile DynCtl *q = reinterpret_cast<volatile DynCtl *>(tt + (tl.TR + 6) * 3 * 64);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = (blockIdx.x & 7) * tl.nblk8 + (blockIdx.x >> 3);
    if (tile >= tl.ntiles) return;
    const int H = a.H, W = a.W;
    StreamTile t;
    int tx;
    tile_of(tl, tile, H, t.b, tx, t.y0, t.own_rows);
    const StripX sx = strip_x(tx, tl.nsx, W);
    t.x0 = sx.x0; t.lane = lane; t.x = sx.x0 + lane;
    t.own_col = t.x >= sx.own0 && t.x < sx.own1;
    t.HW = (unsigned)(H * W);
    t.r_lo = max(0, 3 - t.y0); t.r_hi = min(t.own_rows + 6, H - t.y0 + 3);
    t.nrows = t.r_hi - t.r_lo;
    t.npairs = (t.own_rows + 1) >> 1;
    t.nstage = (t.nrows + 3) >> 2;
    for (int i = threadIdx.x; i < (int)(sizeof(DynCtl) / sizeof(int)); i += NW * 64) reinterpret_cast<volatile int *>(q)[i] = 0;
    if (a.loss_part && pp.last && wave == 0 && lane < 16 && lane >= t.npairs) a.loss_part[tile * 16 + lane] = 0.f;
    const argp ap = (argp)__builtin_amdgcn_kernarg_segment_ptr();      // (`a` is the first kernel argument)
    __syncthreads();
    if (sx.kind == LEFT) stream_tile<NW, LEFT>(ap, pp, t, wl, tt, q, lane >= 1 && lane <= 3, tile, H);
    else if (sx.kind == RIGHT) stream_tile<NW, RIGHT>(ap, pp, t, wl, tt, q, lane >= 60 && lane <= 62, tile, H);
    else stream_tile<NW, INTERIOR>(ap, pp, t, wl, tt, q, false, tile, H);
}

// =====================================================================================================================
// backward of the fused forward w.r.t. depth and the projection matrices, tile edition (both sources of a pair in one pass).
//
// d loss / d warped_s(q) collects the coefficient planes (photo_coef) of every window that contains q and was won by source s:
// the adjoint of ReflectionPad2d(3) + AvgPool2d(7,1) in gather form — a zero-padded 7x7 box sum plus the mirrored terms at the
// image border (rows / columns 1..3 receive the windows of rows / columns 0..3-q once more).  Vertically that is a sum over the
// <= 7 rows around q with a wave-uniform multiplicity (refl_mult), taken straight from L2 (the rows of a tile are shared by its
// four waves and by the neighbouring tiles of the same XCD); horizontally the same six-DPP chain as the forward plus the border
// prefix sums mirrored inside the border quad.  Then, per owned pixel and source: the bilinear adjoint w.r.t. the sampling
// position (clamp masks as ATen's clip_coordinates_set_grad; taps as 8-byte pair loads), the adjoints of Project3D and
// BackprojectDepth; g_depth of both sources is written as ONE plane, the 12 entries of g_P per source are reduced per wave.
// Replaces round 2's column march (one wave per image x source x strip, seven accumulator rows in registers, 172 us at config B).
__device__ __forceinline__ int refl_mult(int r, int q, int n) {
    // number of offsets d in [-3,3] with reflect(r+d) == q, for r,q in [0,n), |r-q| <= 3
    return 1 + (int)(q >= 1 & q + r <= 3) + (int)(q <= n - 2 & (n - 1 - q) + (n - 1 - r) <= 3);
}

// adjoint of the reflected 7-tap window sum along x for three quantities: the zero-padded box7 plus, in a LEFT strip, columns
// 1..3 += g0+g1+g2 / g0+g1 / g0 (prefix sums of lanes 0..2, mirrored inside the first quad; esrc = lanes 0..2, ekill = lane 0,
// which the mirror hands the spill-over of the prefix chain) — RIGHT strips mirror-image (esrc = lanes 61..63, ekill = lane 63).
__device__ __forceinline__ void box7x3_adj(int kind, float &a, float &b, float &c, bool esrc, bool ekill) {
    float ea = 0.f, eb = 0.f, ec = 0.f;
    if (kind != INTERIOR) {                                                   // (wave-uniform)
        const float ma = esrc ? a : 0.f, mb = esrc ? b : 0.f, mc = esrc ? c : 0.f;
        if (kind == LEFT) {
            ea = quad_reverse<0, 0>((ma + row_shr<1>(ma)) + row_shr<2>(ma));
            eb = quad_reverse<0, 0>((mb + row_shr<1>(mb)) + row_shr<2>(mb));
            ec = quad_reverse<0, 0>((mc + row_shr<1>(mc)) + row_shr<2>(mc));
        } else {
            ea = quad_reverse<3, 3>((ma + row_shl<1>(ma)) + row_shl<2>(ma));
            eb = quad_reverse<3, 3>((mb + row_shl<1>(mb)) + row_shl<2>(mb));
            ec = quad_reverse<3, 3>((mc + row_shl<1>(mc)) + row_shl<2>(mc));
        }
        ea = ekill ? 0.f : ea; eb = ekill ? 0.f : eb; ec = ekill ? 0.f : ec;
    }
    float ra = a + wave_shr1(a), rb = b + wave_shr1(b), rc = c + wave_shr1(c);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc);
    ra = a + wave_shr1(ra); rb = b + wave_shr1(rb); rc = c + wave_shr1(rc);
    float ua = a + wave_shl1(a), ub = b + wave_shl1(b), uc = c + wave_shl1(c);
    ua = a + wave_shl1(ua); ub = b + wave_shl1(ub); uc = c + wave_shl1(uc);
    a = (ra + wave_shl1(ua)) + ea; b = (rb + wave_shl1(ub)) + eb; c = (rc + wave_shl1(uc)) + ec;
}

// the adjoint window sums of NINE planes (one source's coefficients) as one hand-scheduled block of 54 v_add_f32_dpp — box7x7's scheme; the
// compiler's own code for box7x3_adj carried one v_mov_b32_dpp per two adds and hazard nops between the three-chain groups
__device__ __forceinline__ void box7x9_adj(int kind, float (&v)[9], bool esrc, bool ekill) {
    float e[9];
    if (kind != INTERIOR) {                                                   // (wave-uniform)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float m = esrc ? v[i] : 0.f;
            const float t = kind == LEFT ? quad_reverse<0, 0>((m + row_shr<1>(m)) + row_shr<2>(m)) : quad_reverse<3, 3>((m + row_shl<1>(m)) + row_shl<2>(m));
            e[i] = ekill ? 0.f : t;
        }
    }
    float r[9], u[9];
    asm("s_nop 1\n"
        "v_add_f32_dpp %9, %0, %0" SQD_DPP_SHR "v_add_f32_dpp %10, %1, %1" SQD_DPP_SHR "v_add_f32_dpp %11, %2, %2" SQD_DPP_SHR "v_add_f32_dpp %12, %3, %3" SQD_DPP_SHR "v_add_f32_dpp %13, %4, %4" SQD_DPP_SHR "v_add_f32_dpp %14, %5, %5" SQD_DPP_SHR "v_add_f32_dpp %15, %6, %6" SQD_DPP_SHR "v_add_f32_dpp %16, %7, %7" SQD_DPP_SHR "v_add_f32_dpp %17, %8, %8" SQD_DPP_SHR
        "v_add_f32_dpp %9, %9, %0" SQD_DPP_SHR "v_add_f32_dpp %10, %10, %1" SQD_DPP_SHR "v_add_f32_dpp %11, %11, %2" SQD_DPP_SHR "v_add_f32_dpp %12, %12, %3" SQD_DPP_SHR "v_add_f32_dpp %13, %13, %4" SQD_DPP_SHR "v_add_f32_dpp %14, %14, %5" SQD_DPP_SHR "v_add_f32_dpp %15, %15, %6" SQD_DPP_SHR "v_add_f32_dpp %16, %16, %7" SQD_DPP_SHR "v_add_f32_dpp %17, %17, %8" SQD_DPP_SHR
        "v_add_f32_dpp %9, %9, %0" SQD_DPP_SHR "v_add_f32_dpp %10, %10, %1" SQD_DPP_SHR "v_add_f32_dpp %11, %11, %2" SQD_DPP_SHR "v_add_f32_dpp %12, %12, %3" SQD_DPP_SHR "v_add_f32_dpp %13, %13, %4" SQD_DPP_SHR "v_add_f32_dpp %14, %14, %5" SQD_DPP_SHR "v_add_f32_dpp %15, %15, %6" SQD_DPP_SHR "v_add_f32_dpp %16, %16, %7" SQD_DPP_SHR "v_add_f32_dpp %17, %17, %8" SQD_DPP_SHR
        "v_add_f32_dpp %18, %0, %0" SQD_DPP_SHL "v_add_f32_dpp %19, %1, %1" SQD_DPP_SHL "v_add_f32_dpp %20, %2, %2" SQD_DPP_SHL "v_add_f32_dpp %21, %3, %3" SQD_DPP_SHL "v_add_f32_dpp %22, %4, %4" SQD_DPP_SHL "v_add_f32_dpp %23, %5, %5" SQD_DPP_SHL "v_add_f32_dpp %24, %6, %6" SQD_DPP_SHL "v_add_f32_dpp %25, %7, %7" SQD_DPP_SHL "v_add_f32_dpp %26, %8, %8" SQD_DPP_SHL
        "v_add_f32_dpp %18, %18, %0" SQD_DPP_SHL "v_add_f32_dpp %19, %19, %1" SQD_DPP_SHL "v_add_f32_dpp %20, %20, %2" SQD_DPP_SHL "v_add_f32_dpp %21, %21, %3" SQD_DPP_SHL "v_add_f32_dpp %22, %22, %4" SQD_DPP_SHL "v_add_f32_dpp %23, %23, %5" SQD_DPP_SHL "v_add_f32_dpp %24, %24, %6" SQD_DPP_SHL "v_add_f32_dpp %25, %25, %7" SQD_DPP_SHL "v_add_f32_dpp %26, %26, %8" SQD_DPP_SHL
        "v_add_f32_dpp %0, %18, %9" SQD_DPP_SHL "v_add_f32_dpp %1, %19, %10" SQD_DPP_SHL "v_add_f32_dpp %2, %20, %11" SQD_DPP_SHL "v_add_f32_dpp %3, %21, %12" SQD_DPP_SHL "v_add_f32_dpp %4, %22, %13" SQD_DPP_SHL "v_add_f32_dpp %5, %23, %14" SQD_DPP_SHL "v_add_f32_dpp %6, %24, %15" SQD_DPP_SHL "v_add_f32_dpp %7, %25, %16" SQD_DPP_SHL "v_add_f32_dpp %8, %26, %17" SQD_DPP_SHL
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7]), "=&v"(u[8]));
    if (kind != INTERIOR) {
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] += e[i];
    }
}

// images of at most 64 columns: BOTH image borders lie inside the wavefront (column 0 in lane `lane0`), and a border's mirrored
// terms may land on columns the other strip owns; they come from three lanes read with bpermute (tiny images only — tests)
__device__ __forceinline__ float virt_border_adj(float v, int x, int W, int lane0) {
    const float l0 = __shfl(v, lane0, 64), l1 = __shfl(v, lane0 + 1, 64), l2 = __shfl(v, lane0 + 2, 64);
    const float r0 = __shfl(v, (lane0 + W - 1) & 63, 64), r1 = __shfl(v, (lane0 + W - 2) & 63, 64), r2 = __shfl(v, (lane0 + W - 3) & 63, 64);
    float e = x == 1 ? (l0 + l1) + l2 : x == 2 ? l0 + l1 : x == 3 ? l0 : 0.f;
    e += x == W - 2 ? (r0 + r1) + r2 : x == W - 3 ? r0 + r1 : x == W - 4 ? r0 : 0.f;
    return e;
}

struct BwdPix {
    float wm1, hm1, gscale, l1w;            // l1w: weight of the L1 term's sign in d loss / d warped
    int W, H;
    unsigned HW;
};

// per pixel and source: d loss / d warped (window terms G + the L1 term where this source won the pixel itself) pulled back
// through grid_sample, Project3D and BackprojectDepth (reference layers.py:186-258; ATen grid_sampler_2d_backward).
// Returns the contribution to g_depth; accumulates the 12 entries of g_P.
template <bool PX = false>
__device__ __forceinline__ float pixel_adjoint(const BwdPix &k, const float *__restrict__ src, const float *__restrict__ smp, const float *P,
                                               const float G[9], const float t[3], const float cr[3], const float X[3], bool l1on, unsigned qo,
                                               float gP[12]) {
    const int W = k.W, H = k.H;
    const float wm1 = k.wm1, hm1 = k.hm1;
    const float2 gs = *reinterpret_cast<const float2 *>(smp + (size_t)qo * 2);
    // recompute taps from the stored grid exactly as the forward did
    float ix = ((gs.x + 1.0f) * 0.5f) * wm1, iy = ((gs.y + 1.0f) * 0.5f) * hm1;
    const bool mx = ix > 0.f && ix < wm1, my = iy > 0.f && iy < hm1;   // clip_coordinates_set_grad
    ix = fminf(wm1, fmaxf(ix, 0.f));
    iy = fminf(hm1, fmaxf(iy, 0.f));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const bool xin = x0 + 1 < W, yin = y0 + 1 < H;
    const float ax = ix - fx0, ay = iy - fy0, bxw = (fx0 + 1.f) - ix, byw = (fy0 + 1.f) - iy;
    // the two taps of a row are 8 contiguous bytes: one (4-byte aligned) load; at the last column the pair starts one pixel
    // earlier and the tap is its second element
    const unsigned o00 = (unsigned)(y0 * W + x0 - (xin ? 0 : 1)), o10 = o00 + (yin ? (unsigned)W : 0u);
    float gix = 0.f, giy = 0.f;
    v2f pnc[3], psc[3];
    if constexpr (PX) {      // pixel-interleaved source: the pair of a row is R0 G0 B0 R1 | G1 B1 — a 16-byte and an 8-byte gather
        const v4f n4 = ldg4(src, o00 * 12u), s4 = ldg4(src, o10 * 12u);
        const v2f n2 = ldg2(src, o00 * 12u + 16u), s2 = ldg2(src, o10 * 12u + 16u);
        pnc[0] = v2f{n4.x, n4.w}; pnc[1] = v2f{n4.y, n2.x}; pnc[2] = v2f{n4.z, n2.y};
        psc[0] = v2f{s4.x, s4.w}; psc[1] = v2f{s4.y, s2.x}; psc[2] = v2f{s4.z, s2.y};
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pnc[c] = ldg2(src, (o00 + c * k.HW) * 4u);
            psc[c] = ldg2(src, (o10 + c * k.HW) * 4u);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const v2f pn = pnc[c], ps = psc[c];
        const float vnw = xin ? pn.x : pn.y, vne = xin ? pn.y : 0.f, vsw = yin ? (xin ? ps.x : ps.y) : 0.f, vse = (xin && yin) ? ps.y : 0.f;
        float wv = vnw * (bxw * byw);
        wv = fmaf(vne, ax * byw, wv);
        wv = fmaf(vsw, bxw * ay, wv);
        wv = fmaf(vse, ax * ay, wv);
        // d to_optimise / d w_c  (window terms + L1 term), trainer.py:441-453
        float gw = G[c] + 2.f * wv * G[3 + c] + t[c] * G[6 + c];
        if (l1on) {
            const float df = wv - t[c];
            gw += k.l1w * (df > 0.f ? 1.f : df < 0.f ? -1.f : 0.f);
        }
        gix += gw * ((vne - vnw) * byw + (vse - vsw) * ay);
        giy += gw * ((vsw - vnw) * bxw + (vse - vne) * ax);
    }
    // unnormalise + clamp adjoints, then (x - 0.5)*2, / (W-1)
    const float ggx = mx ? gix * (wm1 * 0.5f) : 0.f, ggy = my ? giy * (hm1 * 0.5f) : 0.f;
    const float gu = (ggx * 2.f) / wm1, gv = (ggy * 2.f) / hm1;
    const float camz = fmaf(P[11], 1.0f, fmaf(P[10], X[2], fmaf(P[9], X[1], P[8] * X[0])));
    const float z = camz + 1e-7f;
    const float camx = fmaf(P[3], 1.0f, fmaf(P[2], X[2], fmaf(P[1], X[1], P[0] * X[0])));
    const float camy = fmaf(P[7], 1.0f, fmaf(P[6], X[2], fmaf(P[5], X[1], P[4] * X[0])));
    const float iz = 1.f / z;
    float gpx = gu * iz, gpy = gv * iz;
    float gpz = -(gu * camx + gv * camy) * iz * iz;
    gpx *= k.gscale; gpy *= k.gscale; gpz *= k.gscale;
    float gX[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) gX[i] = P[i] * gpx + P[4 + i] * gpy + P[8 + i] * gpz;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gP[i] = fmaf(gpx, X[i], gP[i]);
        gP[4 + i] = fmaf(gpy, X[i], gP[4 + i]);
        gP[8 + i] = fmaf(gpz, X[i], gP[8 + i]);
    }
    gP[3] += gpx; gP[7] += gpy; gP[11] += gpz;
    return cr[0] * gX[0] + cr[1] * gX[1] + cr[2] * gX[2];
}

template <bool AVG, bool PX = false>
__device__ __forceinline__ void bwd_rows(int KIND, const sqd_photo_bwd_args &a, const PairPass &pp, const BwdPix &k, __amdgpu_buffer_rsrc_t coef_r,
                                         __amdgpu_buffer_rsrc_t idx_r, int lane0, int b, int wave, int y0, int own_rows, int x, bool in_col, bool own_col,
                                         int lane, float *gdep, float gP0[12], float gP1[12]) {
    const int W = k.W, H = k.H, NS = pp.S;
    const unsigned HW = k.HW;
    const bool two = pp.s1 != pp.s0;
    const unsigned xoff = in_col ? (unsigned)x * 4u : 0x80000000u, xoffb = in_col ? (unsigned)x : 0x80000000u;
    const bool esrc = KIND == LEFT ? lane <= 2 : lane >= 61, ekill = KIND == LEFT ? lane == 0 : lane == 63;
    float ik[9], P0[12], P1[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ik[i * 3 + j] = a.inv_K[(size_t)b * 16 + i * 4 + j];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        P0[j] = a.P[((size_t)b * NS + pp.s0) * 12 + j];
        P1[j] = a.P[((size_t)b * NS + pp.s1) * 12 + j];
    }
    const float *__restrict__ tgt = a.target + (size_t)b * 3 * HW;
    const float *__restrict__ dep = a.depth + (size_t)b * HW;
    const float *__restrict__ src0 = a.sources[pp.s0] + (size_t)b * 3 * HW, *__restrict__ src1 = a.sources[pp.s1] + (size_t)b * 3 * HW;
    const float *__restrict__ smp0 = a.sample[pp.s0] + (size_t)b * HW * 2, *__restrict__ smp1 = a.sample[pp.s1] + (size_t)b * HW * 2;
    constexpr bool avg = AVG;
    // argmin codes of the pair's reprojection candidates (--avg_reprojection: NS = the mean of both: the two sources read their own
    // nine planes of the 18 sqd_photo_coef wrote)
    const int id0 = avg ? NS : NS + pp.s0, id1 = avg ? NS : two ? NS + pp.s1 : -1;
    // --avg_reprojection: the planes of source s start at 9 s HW (sqd_photo_coef wrote 9 S of them)
    const unsigned c0off = avg ? 9u * (unsigned)pp.s0 * HW : 0u, c1off = avg ? 9u * (unsigned)pp.s1 * HW : 0u;
    for (int j = wave; j < own_rows; j += 4) {
        const int q = y0 + j;
        float G0[9], G1[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) G0[c] = G1[c] = 0.f;
        int idc = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int r = q - 3 + i;
            // branch-free: a row outside the image reads zeros (offset beyond the descriptor) with multiplicity 0, so that the
            // loads of all rows are one straight-line batch
            // (pure arithmetic on purpose: a select on the wave-uniform row test becomes a branch, and a branch per row keeps the
            //  scheduler from batching the loads of the seven rows)
            const int ok = (int)((unsigned)r < (unsigned)H);
            const int rc = min(max(r, 0), H - 1);
            const float m = (float)(refl_mult(rc, q, H) * ok);
            const unsigned row = (unsigned)(rc * W);
            const unsigned oob = (unsigned)(ok - 1) & 0x80000000u;
            const unsigned vo = xoff | oob, vob = xoffb | oob;
            const int id = (int)__builtin_amdgcn_raw_buffer_load_b8(idx_r, vob, row, 0) & 0xff;       // 0 beyond the image
            if (i == 3) idc = id;
            const float w0 = id == id0 ? m : 0.f, w1 = id == id1 ? m : 0.f;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const float v = bld(coef_r, vo, (row + (unsigned)c * HW + c0off) * 4u);
                G0[c] = fmaf(w0, v, G0[c]);
                if constexpr (avg) G1[c] = fmaf(w1, bld(coef_r, vo, (row + (unsigned)c * HW + c1off) * 4u), G1[c]);
                else G1[c] = fmaf(w1, v, G1[c]);
            }
        }
        if (lane0 >= 0) {
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const float e0 = virt_border_adj(G0[c], x, W, lane0), e1 = virt_border_adj(G1[c], x, W, lane0);
                G0[c] = box7(G0[c]) + e0;
                G1[c] = box7(G1[c]) + e1;
            }
        } else {
#ifdef SQD_BWD_ADJ3                                                               // (A/B builds: the compiler-scheduled three-plane sums)
#pragma unroll
            for (int c = 0; c < 9; c += 3) {
                box7x3_adj(KIND, G0[c], G0[c + 1], G0[c + 2], esrc, ekill);
                box7x3_adj(KIND, G1[c], G1[c + 1], G1[c + 2], esrc, ekill);
            }
#else
            box7x9_adj(KIND, G0, esrc, ekill);
            box7x9_adj(KIND, G1, esrc, ekill);
#endif
        }
        if (!own_col) continue;
        const unsigned qo = (unsigned)(q * W + x);
        float t[3], cr[3], X[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) t[c] = ldg(tgt, (qo + c * HW) * 4u);
        const float d = ldg(dep, qo * 4u);
        const float fx = (float)x, fy = (float)q;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float acc = ik[i * 3 + 0] * fx;                 // layers.py:211  (FMA chain k = 0..2)
            acc = fmaf(ik[i * 3 + 1], fy, acc);
            acc = fmaf(ik[i * 3 + 2], 1.0f, acc);
            cr[i] = acc;
            X[i] = d * acc;
        }
        float g = pixel_adjoint<PX>(k, src0, smp0, P0, G0, t, cr, X, idc == id0, qo, gP0);
        if (two) g += pixel_adjoint<PX>(k, src1, smp1, P1, G1, t, cr, X, idc == id1, qo, gP1);
        gdep[qo] = g;
    }
}

// Two workgroups (8 waves) per CU: the scheduler is left free to batch the 70 coefficient loads of a row and both sources' tap
// gathers (140 VGPRs).  Measured at config B (profiles/r03h_photo_bwd_variants.md): 4 waves per SIMD with the loads fenced into
// 128 registers 95 us, 3 waves 98 us, 2 waves 74 us; tile heights 6..16 within 10 % of each other, 16 best.
template <bool AVG, bool PX = false>
__global__ __launch_bounds__(256, 2) void photo_bwd_tile_kernel(sqd_photo_bwd_args a, PairPass pp, int pass, int TR, int nsx, int nsy, int ntiles, int nblk8) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = (blockIdx.x & 7) * nblk8 + (blockIdx.x >> 3);          // neighbouring tiles (shared rows / columns) on one XCD
    if (tile >= ntiles) return;
    const int H = a.H, W = a.W;
    const unsigned HW = (unsigned)(H * W);
    const int tx = tile % nsx, t2 = tile / nsx, ty = t2 % nsy, b = t2 / nsy;
    const StripX sx = strip_x(tx, nsx, W);
    const int y0 = ty * TR;
    const int own_rows = min(TR, H - y0);
    const int x = sx.x0 + lane;
    const bool in_col = x >= 0 && x < W;
    const bool own_col = x >= sx.own0 && x < sx.own1 && in_col;
    const unsigned ncoef = AVG ? 9u * (unsigned)pp.S : 9u;
    const __amdgpu_buffer_rsrc_t coef_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.coef + (size_t)b * ncoef * HW), 0, ncoef * HW * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t idx_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.idx + (size_t)b * HW), 0, HW, 0x00020000);
    BwdPix k;
    k.wm1 = (float)(W - 1); k.hm1 = (float)(H - 1); k.gscale = a.gscale; k.W = W; k.H = H; k.HW = HW;
    k.l1w = ((a.loss_flags & SQD_LOSS_NO_SSIM) ? 1.f / 3.f : 0.15f / 3.f) * (AVG ? 1.f / (float)pp.S : 1.f);
    float *gdep = a.g_depth + (size_t)b * a.g_depth_img_stride + (size_t)pass * HW;
    float gP0[12], gP1[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) gP0[j] = gP1[j] = 0.f;
    const int lane0 = W <= 64 ? -sx.x0 : -1;                                  // >= 0: both borders inside the wavefront (generic border path)
    bwd_rows<AVG, PX>(lane0 >= 0 ? (int)INTERIOR : sx.kind, a, pp, k, coef_r, idx_r, lane0, b, wave, y0, own_rows, x, in_col, own_col, lane, gdep, gP0, gP1);
    // per-wavefront partials of g_P: [B][S][tiles per image * 4][12]
    const int tpi = nsx * nsy * 4, slot = (ty * nsx + tx) * 4 + wave;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const float v0 = wave_sum_to_lane63(gP0[j]), v1 = wave_sum_to_lane63(gP1[j]);
        if (lane == 63) {
            a.g_P_part[(((size_t)b * pp.S + pp.s0) * tpi + slot) * 12 + j] = v0;
            if (pp.s1 != pp.s0) a.g_P_part[(((size_t)b * pp.S + pp.s1) * tpi + slot) * 12 + j] = v1;
        }
    }
}

// rows a workgroup tile owns: the caller's rows_per_task (even, clamped), or the default of the kernel family.  Measured at config B
// after a proper warm-up (profiles/r03m_photo_tile_heights.md): every uniform shape of the fused forward within 1.5 % of the others
// (4 x 16: 51.2 us, 8 x 28: 50.7), identity / coefficient / backward kernels flat in the tile height — what does count is a whole number
// of tiles per CU (make_tiling's balanced cut: 49.6 us)
enum { FAMILY_FWD = 1, FAMILY_ROWS = 0, FAMILY_BWD = 2 };
int pick_tr(int rows, int family, int H) {
    const int cap = family == FAMILY_FWD && (rows > TR_MAX || (rows <= 0 && H >= 56)) ? 2 * TR_MAX : TR_MAX;
    int tr = rows > 0 ? rows : family == FAMILY_FWD ? (H >= 56 ? 28 : TR_MAX) : family == FAMILY_ROWS ? 12 : TR_MAX;
    tr = tr > cap ? cap : tr;
    tr &= ~1;
    return tr < 2 ? 2 : tr;
}

// the tiling of a launch: family FAMILY_FWD with rows_per_task == 0 -> balanced (a whole number of equal-cost tiles per CU, 8 waves),
// when the image leaves tiles of 12..32 rows for that; otherwise uniform tiles of pick_tr rows
Tiling make_tiling(int B, int H, int W, int rows_per_task, int family) {
    Tiling tl = {};
    tl.nsx = strips_x(W);
    const int cols = B * tl.nsx;
    // (only where 28-row tiles leave the last round of tiles more than 6 % empty: at config C — 1728 tiles, 6.75 per CU — the
    //  balanced cut measured no better than the uniform one, within that shape's +-8 % run-to-run spread)
    const int tu = cols * ((H + 27) / 28);
    const double fill = (double)tu / 256.0, imbalance = (double)((tu + 255) / 256) / fill;
    if (family == FAMILY_FWD && rows_per_task <= 0 && H >= 56 && imbalance > 1.06) {
        // 256 CUs x 2 resident 8-wave workgroups: tiles in multiples of 512, as close to 26 rows each as that allows
        const long long rows_total = (long long)cols * H;
        int best = 0;
        for (int t = 512; t <= 512 * 64; t += 512) {
            const double r = (double)rows_total / t;
            if (r < 12.0) break;
            if (r <= 30.0 && best == 0) best = t;            // the fewest tiles (least halo) that fit the 32-row LDS tile
        }
        const int n_lo = best / cols, n_hi_cols = best - n_lo * cols;
        // tallest tile of the launch: a column with n_lo tiles, cut on even rows
        if (best > 0 && n_lo >= 1 && 2 * (((H + 1) / 2 + n_lo - 1) / n_lo) <= 2 * TR_MAX) {
            tl.n_lo = n_lo;
            tl.n_hi_cols = n_hi_cols;
            tl.ntiles = best;
            tl.TR = 2 * (((H + 1) / 2 + n_lo - 1) / n_lo);
            tl.nsy = 0;
            tl.nblk8 = (tl.ntiles + 7) / 8;
            return tl;
        }
    }
    tl.TR = pick_tr(rows_per_task, family, H);
    tl.nsy = (H + tl.TR - 1) / tl.TR;
    tl.ntiles = cols * tl.nsy;
    tl.nblk8 = (tl.ntiles + 7) / 8;
    return tl;
}
}  // namespace

#ifdef SQD_PHOTO_TRACE
extern "C" int sqd_photo_trace(void *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_photo_trace), &buf, sizeof(buf)); }
#endif
namespace sqd {
static int g_fwd_variant = 0;      // 0: default (stream kernel on 8-wave tilings), 1: round 5's kernel, 2: colour-serial phase 2, 4: wide edition
static int g_fwd_skew = 0, g_fwd_resident = 0, g_fwd_pipe = 0, g_fwd_late = 0;
void photo_set_fwd_variant(int v) {
    g_fwd_variant = v & 0x3f;
    g_fwd_late = g_fwd_variant == 6;      // 6: the lean kernel with its late rows warped behind the barrier (warp_tile; measured: no faster)
    if (g_fwd_variant == 6) g_fwd_variant = 0;
    g_fwd_pipe = (v & 0x40) != 0; g_fwd_resident = (v & 0x80) != 0; g_fwd_skew = (v >> 8) * 256;
}
int photo_fwd_waves(int B, int H, int W, int rows_per_task) {      // loss partials per tile
    const Tiling tl = make_tiling(B, H, W, rows_per_task, FAMILY_FWD);
    return tl.TR > TR_MAX ? 16 : 4;
}
// do the kernels of a training step (identity maps, lean forward with every training output, default backward) read [B,H,W,3] sources at this shape?
bool photo_sources_hwc_ok(int B, int S, int H, int W, int rows_per_task, int loss_flags) {
    const Tiling tl = make_tiling(B, H, W, rows_per_task, FAMILY_FWD);
    return S == 2 && (loss_flags & ~SQD_SOURCES_HWC) == 0 && W >= 64 && (size_t)H * W % 4 == 0 && !g_fwd_variant && tl.TR > TR_MAX && tl.TR <= 36;
}
int photo_tile_count(int B, int H, int W, int rows_per_task, int family) { return make_tiling(B, H, W, rows_per_task, family).ntiles; }

// mode 0: identity maps, 1: fused forward, 2: coefficient planes of the backward (a.warped = stored warps, a.idx, a.coef)
int launch_photo_tile(const sqd_photo_args &a, const float *noise, int mode, hipStream_t stream) {
    Tiling tl = make_tiling(a.B, a.H, a.W, a.rows_per_task, mode == 1 ? FAMILY_FWD : FAMILY_ROWS);
    tl.skew = mode == 1 ? g_fwd_skew : 0;
    tl.pipe = g_fwd_pipe;
    tl.late = g_fwd_late && !g_fwd_resident;
    const int NW = mode == 1 && tl.TR > TR_MAX ? 8 : 4;
    const dim3 grid(tl.nblk8 * 8), block(NW * 64);
    // (SQD_SOURCES_HWC travels in loss_flags)
    const bool hwc = (a.loss_flags & SQD_SOURCES_HWC) != 0;
    const int opts = a.loss_flags & ~SQD_SOURCES_HWC;
    const bool lean = mode == 1 && NW == 8 && g_fwd_variant == 0 && a.W >= 64 && a.S == 2 && opts == 0 && !a.reproj && !a.x0y0[0] && !a.x0y0[1] && a.sel && a.idx &&
                      a.sample[0] && a.sample[1] && a.warped[0] && a.warped[1] && a.identity && tl.TR <= 36;
    if (hwc && mode != 2 && !(a.W >= 64 && a.S == 2 && opts == 0 && !g_fwd_variant && !g_fwd_resident && (mode == 0 || lean))) {
        // (the kernels that read [B,H,W,3] sources: the lean forward, the option-free identity maps, the default backward)
        set_error("SQD_SOURCES_HWC: only with two source frames, the default loss options, W >= 64, every output of the forward requested except the tap / "
                  "reprojection dumps, and the default kernel variant");
        return SQD_EINVAL;
    }
    for (int k = 0; 2 * k < a.S; ++k) {                  // one launch per pair of source frames
        const PairPass pp = {2 * k, 2 * k + 1 < a.S ? 2 * k + 1 : 2 * k, a.S, k == 0, 2 * k + 2 >= a.S};
        if (mode == 0)
            // (default loss options, one pair pass: the option-free / reflection-free / hand-scheduled-shuffle paths of the lean forward)
            if (hwc) hipLaunchKernelGGL((photo_tile_kernel<0, 4, false, true, false, true>), grid, block, 0, stream, a, pp, noise, tl);
            else if (opts == 0 && a.S == 2 && !g_fwd_variant) hipLaunchKernelGGL((photo_tile_kernel<0, 4, false, true>), grid, block, 0, stream, a, pp, noise, tl);
            else hipLaunchKernelGGL((photo_tile_kernel<0>), grid, block, 0, stream, a, pp, noise, tl);
        else if (mode == 1 && NW == 8 && g_fwd_variant == 2)
            hipLaunchKernelGGL((photo_fwd_c_kernel<8>), grid, block, (tl.TR + 6) * 3 * 64 * 8, stream, a, pp, tl);
        else if (lean) {
            const int lds = (tl.TR + 6) * 3 * 64 * 12 + 16;      // warped rows (8 bytes per cell and colour), target rows (4), the late-row count
            static int lds_ok = 0;                   // (dynamic LDS beyond 64 KB is an opt-in of the function)
            if (!lds_ok) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_tile_kernel<1, 8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                lds_ok = 1;
            }
            static int lds_ok2 = 0;
            if (!lds_ok2) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_tile_resident_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                lds_ok2 = 1;
            }
            const int per_xcd = tl.nblk8 < 64 ? tl.nblk8 : 64;            // 256 CUs x 2 resident workgroups = 64 per XCD
            if (g_fwd_resident && tl.nblk8 > 64)
                hipLaunchKernelGGL((photo_tile_resident_kernel<8>), dim3(per_xcd * 8), block, lds, stream, a, pp, tl, per_xcd);
            else if (g_fwd_pipe) {
                static int lds_ok3 = 0;
                if (!lds_ok3) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_tile_kernel<1, 8, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_tile_kernel<1, 8, true, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    lds_ok3 = 1;
                }
                if (hwc) hipLaunchKernelGGL((photo_tile_kernel<1, 8, true, true, true, true>), grid, block, lds, stream, a, pp, noise, tl);
                else hipLaunchKernelGGL((photo_tile_kernel<1, 8, true, true, true>), grid, block, lds, stream, a, pp, noise, tl);
            } else if (hwc) {                                     // pixel-interleaved sources: eight gathers per cell instead of twelve
                static int lds_ok4 = 0;
                if (!lds_ok4) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_tile_kernel<1, 8, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    lds_ok4 = 1;
                }
                hipLaunchKernelGGL((photo_tile_kernel<1, 8, true, true, false, true>), grid, block, lds, stream, a, pp, noise, tl);
            } else
                hipLaunchKernelGGL((photo_tile_kernel<1, 8, true, true>), grid, block, lds, stream, a, pp, noise, tl);
        } else if (mode == 1 && NW == 8 && g_fwd_variant == 5 && a.W >= 64) {
            const int lds = (tl.TR + 6) * 3 * 64 * 12 + (int)sizeof(DynCtl);
            static int lds_ok = 0;                   // (dynamic LDS beyond 64 KB is an opt-in of the function)
            if (!lds_ok) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_fwd_s_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                lds_ok = 1;
            }
            hipLaunchKernelGGL((photo_fwd_s_kernel<8, 4>), grid, block, lds, stream, a, pp, tl);
        } else if (mode == 1 && NW == 8 && g_fwd_variant == 4 && a.W >= 64) {
            const int lds = (tl.TR + 6) * 3 * 64 * 12;
            static int lds_ok = 0;                   // (dynamic LDS beyond 64 KB is an opt-in of the function)
            if (lds > lds_ok) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_tile_kernel<1, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                lds_ok = 160 * 1024;
            }
            hipLaunchKernelGGL((photo_tile_kernel<1, 8, true>), grid, block, lds, stream, a, pp, noise, tl);
        }
        else if (mode == 1 && NW == 8) {
#ifdef SQD_PHOTO_TRACE
            if (g_fwd_variant == 3) {                  // developer experiment: ONE workgroup per CU (LDS padded to 100 KB)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&photo_tile_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL((photo_tile_kernel<1, 8>), grid, block, 100 * 1024, stream, a, pp, noise, tl);
                continue;
            }
#endif
            hipLaunchKernelGGL((photo_tile_kernel<1, 8>), grid, block, (tl.TR + 6) * 3 * 64 * 8, stream, a, pp, noise, tl);
        }
        else if (mode == 1)
            hipLaunchKernelGGL((photo_tile_kernel<1>), grid, block, (tl.TR + 6) * 3 * 64 * 8, stream, a, pp, noise, tl);
        else
            if (opts == 0 && a.S == 2 && !g_fwd_variant) hipLaunchKernelGGL((photo_tile_kernel<2, 4, false, true>), grid, block, 0, stream, a, pp, noise, tl);
            else hipLaunchKernelGGL((photo_tile_kernel<2>), grid, block, 0, stream, a, pp, noise, tl);
    }
    return SQD_OK;
}

// backward: one launch per pair of source frames; plane `pass` of g_depth receives the pair's contribution
void launch_photo_bwd_tile(const sqd_photo_bwd_args &a, hipStream_t stream) {
    const int TR = pick_tr(a.rows_per_task, FAMILY_BWD, a.H);
    const int nsx = strips_x(a.W), nsy = (a.H + TR - 1) / TR;
    const int ntiles = a.B * nsx * nsy;
    const int nblk8 = (ntiles + 7) / 8;
    for (int k = 0; 2 * k < a.S; ++k) {
        const PairPass pp = {2 * k, 2 * k + 1 < a.S ? 2 * k + 1 : 2 * k, a.S, k == 0, 2 * k + 2 >= a.S};
        if (a.loss_flags & SQD_SOURCES_HWC)      // (default loss options only: photometric.hip)
            hipLaunchKernelGGL((photo_bwd_tile_kernel<false, true>), dim3(nblk8 * 8), dim3(256), 0, stream, a, pp, k, TR, nsx, nsy, ntiles, nblk8);
        else if (a.loss_flags & SQD_LOSS_AVG_REPROJECTION)
            hipLaunchKernelGGL(photo_bwd_tile_kernel<true>, dim3(nblk8 * 8), dim3(256), 0, stream, a, pp, k, TR, nsx, nsy, ntiles, nblk8);
        else
            hipLaunchKernelGGL(photo_bwd_tile_kernel<false>, dim3(nblk8 * 8), dim3(256), 0, stream, a, pp, k, TR, nsx, nsy, ntiles, nblk8);
    }
}
}  // namespace sqd
