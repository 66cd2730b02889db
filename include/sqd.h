/* sqd.h — C ABI of libsqd.so, the MI355X (gfx950) implementation of the SQLdepth self-supervised
 * training hot path (reference: hisfog/SfMNeXt-Impl).
 *
 * The reference has no FFI seam (pure Python): each entry point below replaces a run
 * of ATen dispatches issued by a Python function of the reference, cited per function.  The
 * reference-side binding a maintainer would add is a ctypes stub — see INTEGRATION.md.
 *
 * Contract (all functions):
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns all memory (inputs, outputs, workspaces); the library never allocates, frees,
 *     retains pointers or synchronises; work is enqueued on `stream` (a hipStream_t, passed as void*);
 *   - tensors are fp32, contiguous, NCHW unless stated; shapes are given as int32;
 *   - return 0 on success, a negative SQD_E* code otherwise; sqd_last_error() returns a
 *     thread-local description of the last failure; nothing throws, nothing calls exit();
 *   - re-entrant; process-wide state is limited to (a) that thread-local error string and (b) configuration the CALLER sets
 *     through explicit entry points — the measured convolution plans (sqd_conv_set_plan, sqd_conv_wgrad_set_plan: a
 *     mutex-guarded geometry -> plan table; without a registered plan a cost model decides) and the convolution precision
 *     mode (sqd_conv_set_precision).  The library reads no environment variables.
 */
#ifndef SQD_H_
#define SQD_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQD_ABI_VERSION 3   /* 2: loss_flags in sqd_photo_args / sqd_photo_bwd_args; 3: SQD_SOURCES_HWC, sqd_pack_pixels */
#define SQD_OK 0
#define SQD_EINVAL (-1)   /* bad shape / null pointer / unsupported configuration */
#define SQD_ELAUNCH (-2)  /* hipGetLastError() after launch */

/* the reference's loss options (options.py --no_ssim / --avg_reprojection / --disable_automasking; trainer.py:447-451, 480-524) */
#define SQD_LOSS_NO_SSIM 1           /* reprojection loss = L1 alone                                                             */
#define SQD_LOSS_AVG_REPROJECTION 2  /* mean over the S >= 2 source frames instead of the per-pixel minimum                       */
#define SQD_LOSS_NO_AUTOMASK 4       /* no identity candidates: the minimum runs over the reprojection losses only                */
/* a LAYOUT bit that travels with the loss options: sources[] point at pixel-interleaved frames [B,H,W,3] (torch: channels_last; sqd_pack_pixels
 * makes them) instead of [B,3,H,W].  The 2 x 2 taps of a warped pixel are then two runs of 24 bytes — eight gathers per pixel and source pair
 * instead of twelve in sqd_photo_fwd and sqd_photo_bwd, whose outputs do not change by a bit.  Served for two source frames,
 * the default loss options, W >= 64, by sqd_identity_fwd_ex, sqd_photo_fwd (all outputs but the tap / reprojection dumps requested) and
 * sqd_photo_bwd; SQD_EINVAL elsewhere.  sqd_photo_coef_ex ignores it (it reads no source frame).                                        */
#define SQD_SOURCES_HWC 256
#define SQD_MAX_SOURCES 4 /* source frames per target (reference frame_ids[1:], default 2) */
#define SQD_STRIP_COLS 58 /* output columns per wavefront strip of the column-march kernels */

int sqd_abi_version(void);
const char *sqd_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * (1) depth upsample + per-image reductions
 * replaces: F.interpolate(disp,[H,W],bilinear,align_corners=False)  reference trainer.py:395-399
 *           inv_depth.mean(3).mean(2)                               reference trainer.py:417-418
 *           disp.mean(2).mean(3)                                    reference trainer.py:535
 * disp_lr [B,1,h,w] -> depth [B,1,H,W]; part [B, nblk, 2] = per-block (sum 1/depth, sum depth)
 * with nblk = sqd_depth_up_nblk(H,W).  Deterministic (no atomics).                                  */
int sqd_depth_up_nblk(int H, int W);
int sqd_depth_up_fwd(const float *disp_lr, float *depth, float *part, int B, int h, int w, int H, int W,
                     void *stream);
/* adjoint: g_depth [B,ng,H,W] = ng gradient planes w.r.t. the full-res depth that are summed on the
 * fly (photometric per source + smoothness), g_mid [B] = dL/d(mean_inv_depth[b]) (may be NULL)
 * -> g_disp_lr [B,1,h,w] (overwritten).                                                              */
int sqd_depth_up_bwd(const float *g_depth, int ng, const float *depth, const float *g_mid, float *g_disp_lr,
                     int B, int h, int w, int H, int W, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (2) pose -> projection matrices
 * replaces: transformation_from_parameters / rot_from_axisangle / get_translation_matrix
 *           (reference layers.py:75-150) called from trainer.py:336-337 and :420-421, and
 *           P = (K @ T)[:, :3] of Project3D.forward (reference layers.py:248).
 * axisangle, translation [B,S,3]; invert_host[S] (1 = frame_id < 0); K [B,4,4];
 * part/nblk from sqd_depth_up_fwd, or part == NULL to use scale 1 (the un-scaled cam_T_cam).
 * outputs: mid [B] mean inverse depth (may be NULL when part == NULL), T [B,S,4,4], P [B,S,3,4].     */
int sqd_pose_mats_fwd(const float *axisangle, const float *translation, const int32_t *invert_host,
                      const float *K, const float *part, int nblk, int HW, float *mid, float *T, float *P,
                      int B, int S, void *stream);
/* adjoint: g_P [B,S,3,4] -> g_axisangle, g_translation [B,S,3], g_mid [B] (g_mid may be NULL).       */
int sqd_pose_mats_bwd(const float *axisangle, const float *translation, const int32_t *invert_host,
                      const float *K, const float *mid, const float *g_P, float *g_axisangle,
                      float *g_translation, float *g_mid, int B, int S, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (3) photometric chain
 * replaces (fused): BackprojectDepth.forward (layers.py:210-215), Project3D.forward (layers.py:247-258),
 *   F.grid_sample(border, align_corners=True) (trainer.py:431-435), compute_reprojection_loss
 *   (trainer.py:441-453) with SSIM (layers.py:13-46), the identity losses + tie-break noise, the
 *   per-pixel min / auto-mask and the mean (trainer.py:474-532).
 */
typedef struct sqd_photo_args {
    /* inputs */
    const float *depth;     /* [B,1,H,W] from sqd_depth_up_fwd                                       */
    const float *inv_K;     /* [B,4,4]                                                               */
    const float *P;         /* [B,S,3,4] from sqd_pose_mats_fwd (or caller-computed)                 */
    const float *target;    /* [B,3,H,W]  inputs[("color",0,0)]                                      */
    const float *sources[SQD_MAX_SOURCES]; /* S x [B,3,H,W]  inputs[("color",f,0)], f in frame_ids[1:] */
    const float *identity;  /* [B,S,H,W] identity reprojection loss + 1e-5*noise (sqd_identity_fwd)  */
    /* outputs (any of sample[s]/warped[s]/sel/x0y0[s]/reproj may be NULL to skip the store) */
    float *sample[SQD_MAX_SOURCES];  /* S x [B,H,W,2] outputs[("sample",f,0)]                        */
    float *warped[SQD_MAX_SOURCES];  /* S x [B,3,H,W] outputs[("color",f,0)]                         */
    float *sel;             /* [B,H,W]    outputs["identity_selection/0"] (0/1)                      */
    uint8_t *idx;           /* [B,H,W]    argmin over [identity_0..S-1, reproj_0..S-1]               */
    int32_t *x0y0[SQD_MAX_SOURCES];  /* S x [B,H,W,2] integer grid_sample taps (x0,y0) — parity instrumentation */
    float *coef;            /* not written by sqd_photo_fwd (the backward recomputes it: sqd_photo_coef); leave NULL */
    float *reproj;          /* [B,S,H,W]  reprojection loss maps (debug/parity; may be NULL)         */
    float *loss_part;       /* [ntasks]   per-wavefront partial sums of to_optimise, ntasks = sqd_photo_ntasks(...) */
    int32_t B, S, H, W;
    int32_t loss_flags;     /* SQD_LOSS_* bits: the reference's loss options (0 = its defaults)                  */
    int32_t rows_per_task;  /* rows a workgroup tile owns (even; <= 32 for sqd_photo_fwd — above 16: 8-wave workgroups —, <= 16 elsewhere);
                               0 = the kernel family's measured default (fused forward 28, identity / coefficient maps 12, backward 16) */
    void *stream;
} sqd_photo_args;
/* the pixel-interleaved frames SQD_SOURCES_HWC announces: planar_host[n] device pointers to [B,3,H,W] -> px_host[n] device pointers to
 * [B,H,W,3] (n <= SQD_MAX_SOURCES frames in one launch; H*W a multiple of 4).  No reference counterpart: the layout a batch is copied
 * into where the photometric kernels are its only readers (trainer.py here: the static source frames of the captured step).     */
/* 1 if sqd_identity_fwd_ex, sqd_photo_fwd (training outputs) and sqd_photo_bwd all serve SQD_SOURCES_HWC at this shape and these loss options
 * (what a caller asks before it lays its source frames out that way), else 0.                                                   */
int sqd_photo_sources_hwc_ok(int B, int S, int H, int W, int rows_per_task, int loss_flags);
int sqd_pack_pixels(const float *const *planar_host, float *const *px_host, int n, int B, int H, int W, void *stream);
int sqd_photo_ntasks(int B, int H, int W, int rows_per_task);
int sqd_photo_fwd(const sqd_photo_args *a);
/* which kernel sqd_photo_fwd launches on 8-wave tilings (process-wide, default 0; every choice writes the same bits — kept for
 * same-box A/B runs and tests/test_gpu_photometric.py::test_forward_kernel_variants_agree).  Bits 0-5: 0 = the lean kernel where the launch
 * qualifies (S = 2, loss_flags 0, no tap / reprojection dumps, W >= 64), round 5's kernel otherwise; 1 = round 5's kernel; 2 = colour-serial
 * phase 2; 4 = wide accesses only; 5 = dynamic wave roles; 6 = lean kernel with the rows that do not fill a round of its waves warped behind the barrier.  Bit 6 (0x40): lean kernel with two rows of a wave in flight in phase 1;
 * bit 7 (0x80): lean kernel as resident workgroups; bits 8..: cycles / 256 the second workgroup of a CU waits at its start.  DESIGN.md 3.1. */
int sqd_photo_set_fwd_variant(int variant);

/* identity reprojection losses (trainer.py:480-487) + tie-break noise (trainer.py:514-517).
 * Depends only on the batch, not on the networks: the trainer enqueues it on a side stream.
 * target [B,3,H,W], sources_host[S] device pointers to [B,3,H,W], noise [B,S,H,W] (may be NULL)
 * -> identity [B,S,H,W].                                                                              */
int sqd_identity_fwd(const float *target, const float *const *sources_host, const float *noise, float *identity,
                     int B, int S, int H, int W, int rows_per_task, void *stream);
/* the same under loss options: SQD_LOSS_NO_SSIM -> L1 maps; SQD_LOSS_AVG_REPROJECTION -> identity and noise are [B,1,H,W]: ONE map per
 * image, the mean over the S sources + 1e-5 * noise (sqd_photo_fwd reads identity in that layout under the same flag)          */
int sqd_identity_fwd_ex(const float *target, const float *const *sources_host, const float *noise, float *identity,
                        int B, int S, int H, int W, int rows_per_task, int loss_flags, void *stream);

/* d(to_optimise)/d(7x7 window sums) of the winning source — the "coefficient planes" the backward box-filters — from the
 * warped images sqd_photo_fwd stored: target [B,3,H,W], warped_host[S] device pointers to [B,3,H,W], idx [B,H,W] the argmin
 * byte -> coef [B,9,H,W] (planes: d/d sum w_c, d/d sum w_c^2, d/d sum w_c t_c for c = r,g,b) where a reprojection
 * candidate won (idx >= S), zeros elsewhere (fully written).  Part of the backward pass: the forward launch carries no
 * training-only traffic.                                                                                                */
int sqd_photo_coef(const float *target, const float *const *warped_host, const uint8_t *idx, float *coef, int B, int S, int H,
                   int W, int rows_per_task, void *stream);
/* under loss options: SQD_LOSS_NO_SSIM -> zeros; SQD_LOSS_AVG_REPROJECTION -> coef [B,9 S,H,W]: planes 9 s .. 9 s + 8 belong to source s,
 * each with weight 1/S where the mean reprojection won (sqd_photo_bwd reads them in that layout under the same flag)            */
int sqd_photo_coef_ex(const float *target, const float *const *warped_host, const uint8_t *idx, float *coef, int B, int S, int H,
                      int W, int rows_per_task, int loss_flags, void *stream);

/* backward of sqd_photo_fwd w.r.t. depth and P.  gscale = dL/d(mean to_optimise) / (B*H*W).
 * Tile kernel (same tiles as the forward), both sources of a pair in one pass, one launch per pair of source frames:
 * plane k of g_depth = contribution of sources 2k and 2k+1 (fully overwritten); image b's planes start at
 * g_depth + b*g_depth_img_stride (elements), so the caller can reserve extra planes (smoothness) behind the
 * ceil(S/2) pair planes for sqd_depth_up_bwd.  coef: the planes sqd_photo_coef wrote (fully written; zero where an identity
 * candidate won).  g_P_part [ntasks_bwd, 12] per-wavefront partials, ntasks_bwd = sqd_photo_bwd_ntasks(...)
 * ordered [B][S][tasks_per_image] (tasks_per_image = 4 wavefronts x tiles per image).                 */
typedef struct sqd_photo_bwd_args {
    const float *depth, *inv_K, *P, *target, *coef;
    const float *sources[SQD_MAX_SOURCES]; /* S x [B,3,H,W] */
    const float *sample[SQD_MAX_SOURCES];  /* S x [B,H,W,2] as written by sqd_photo_fwd */
    const uint8_t *idx;
    float *g_depth;
    float *g_P_part;
    int64_t g_depth_img_stride; /* >= ceil(S/2)*H*W */
    float gscale;
    int32_t B, S, H, W;
    int32_t loss_flags;     /* SQD_LOSS_* bits, as given to sqd_photo_fwd / sqd_photo_coef_ex                                   */
    int32_t rows_per_task;  /* rows a workgroup tile owns (even, <= 16); 0 = library default (16) */
    void *stream;
} sqd_photo_bwd_args;
int sqd_photo_bwd_ntasks(int B, int S, int H, int W, int rows_per_task);
int sqd_photo_bwd(const sqd_photo_bwd_args *a);
/* sums the per-wavefront partials: g_P_part [B,S,tasks_per_image,12] -> g_P [B,S,3,4]                */
int sqd_photo_bwd_reduce(const float *g_P_part, float *g_P, int ntasks, int tasks_per_image, int B, int S,
                         void *stream);

/* ------------------------------------------------------------------------------------------------
 * (4) edge-aware smoothness on the mean-normalised depth
 * replaces: disp / (disp.mean(2).mean(3) + 1e-7)  (trainer.py:535-536) and get_smooth_loss
 *           (reference layers.py:267-280).
 * depth [B,1,H,W], color [B,3,H,W], part from sqd_depth_up_fwd (sum depth) ->
 * sm_part [B, nblk_s, 2] = per-block (sum of x terms, sum of y terms), nblk_s = sqd_smooth_nblk(H,W);
 * loss = sum_b,blk sm_part[..,0] / (B*H*(W-1)) + sum sm_part[..,1] / (B*(H-1)*W).                    */
int sqd_smooth_nblk(int H, int W);
int sqd_smooth_fwd(const float *depth, const float *color, const float *part, int nblk, float *sm_part,
                   int B, int H, int W, void *stream);
/* plane = gout * d(smooth)/d(depth): image b's plane is written at g_depth + b*g_depth_img_stride
 * (elements) — one of the planes summed by sqd_depth_up_bwd.  part == NULL (sm_part ignored): the adjoint of the
 * stand-alone get_smooth_loss (sqd_smooth_fwd with part == NULL: no mean normalisation).               */
int sqd_smooth_bwd(const float *depth, const float *color, const float *part, int nblk, const float *sm_part,
                   float gout, float *g_depth, int64_t g_depth_img_stride, int B, int H, int W, void *stream);
/* the scalars of compute_losses (trainer.py:531-545) from the partial sums of sqd_photo_fwd and sqd_smooth_fwd, in one launch:
 * photo = sum(loss_part[0..n_loss)) * w_photo, smooth = sum(sm_part[.., 0]) * w_x + sum(sm_part[.., 1]) * w_y (n_sm pairs),
 * out[3] = (photo + smooth_weight * smooth, photo, smooth).  Fixed summation order.                                          */
int sqd_chain_loss(const float *loss_part, int n_loss, const float *sm_part, int n_sm, float w_photo, float w_x, float w_y,
                   float smooth_weight, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (5) Self Query Layer
 * replaces: FullQueryLayer.forward (reference networks/layers.py:7-21): y = x^T K^T (:17),
 *           softmax over the N = h*w pixels (:18), summary = softmax(y)^T x^T (:19).
 * x [B,E,N] (the [B,E,h,w] feature map; x_nhwc = 1: its channels-last memory [B,N,E], the layout the producing
 * convolution writes — no layout copy), K [B,Q,E] queries -> y [B,Q,N] energy maps (raw dot
 * products), summary [B,Q,E], lse [B,Q,2] = (max, 1/sum) for the backward.  E in {16,32,48,64}, Q <= 128.
 * part: workspace of sqd_sql_workspace(..).part_floats floats.  fp32 MFMA (v_mfma_f32_16x16x4_f32).  */
int sqd_sql_workspace(int B, int Q, int E, int N, int64_t *part_floats, int64_t *gk_part_floats);
int sqd_sql_fwd(const float *x, const float *K, float *y, float *summary, float *lse, float *part, int B, int Q,
                int E, int N, int x_nhwc, void *stream);
/* adjoint: g_y [B,Q,N] (may be NULL), g_summary [B,Q,E] -> g_x (x's layout), g_K [B,Q,E];
 * gk_part: workspace of gk_part_floats floats.                                                       */
int sqd_sql_bwd(const float *x, const float *K, const float *y, const float *g_y, const float *g_summary,
                const float *summary, const float *lse, float *g_x, float *g_K, float *gk_part, int B, int Q,
                int E, int N, int x_nhwc, void *stream);
/* ... and amax_gx (may be NULL; cleared by the caller): the bit pattern of max |g_x| — g_x is the output gradient of the convolution that produced the
 * features; that node reads it on two-term fp16 operands (section 10b) and needs no pass of its own for the scale */
int sqd_sql_bwd_amax(const float *x, const float *K, const float *y, const float *g_y, const float *g_summary, const float *summary,
                     const float *lse, float *g_x, float *g_K, float *gk_part, int B, int Q, int E, int N, int x_nhwc, float *amax_gx, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (6) BatchNorm2d fused with activation and residual add, channels-last activations
 * replaces: nn.BatchNorm2d + nn.ReLU / nn.LeakyReLU (+ the residual add of the ResNet blocks) —
 *           reference networks/resnet_encoder.py:89-117 (torchvision Bottleneck/BasicBlock, UpSampleBN).
 * x, res, y: [M = N*H*W, C] row-major (NHWC memory), C % 4 == 0 and C/4 a power of two.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.01).  part: workspace of sqd_bn_nblk(M,C) * C * 2 floats.
 * training forward: batch statistics (biased variance for the normalisation, unbiased for the
 * running estimate, momentum as nn.BatchNorm2d), saves mean / rstd for the backward.               */
int sqd_bn_nblk(int M, int C);
/* mask (may be NULL): [M*C/4] bytes, bit j of byte i = element 4i+j was positive before the activation; handing it to
 * the backward replaces two full reads of y by two reads of a 16x smaller array                                   */
/* pre_rows > 0: part already holds pre_rows rows of [C][2] (sum, sum of squares) partials written by the producing
 * convolution (sqd_conv_fwd's stats / sqd_conv_fwd_stats_rows) and the reduction pass over x is skipped; else part is
 * scratch of sqd_bn_nblk(M, C) * C * 2 floats                                                                        */
int sqd_bn_train_fwd(const float *x, const float *res, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, float *y, unsigned char *mask, float *save_mean, float *save_rstd, float *part,
                     int pre_rows, int M, int C, float eps, float momentum, int act, void *stream);
/* ... and, for a BatchNorm that feeds a squeeze-and-excite gate (B images of M / B pixels, res == NULL): pool_part
 * [B][sqd_se_chunks(M / B)][C] = the per-image channel sums of y, bit-identical to sqd_se_pool(y) — the gate need not read y again */
int sqd_bn_train_fwd_pool(const float *x, const float *res, const float *gamma, const float *beta, float *running_mean,
                          float *running_var, float *y, unsigned char *mask, float *save_mean, float *save_rstd, float *part,
                          int pre_rows, int M, int C, float eps, float momentum, int act, float *pool_part, int B, void *stream);
/* ... and amax_y (may be NULL; a device scalar the caller cleared on the stream): the element-wise pass records the bit pattern of max |y| —
 * the operand scale of a convolution that reads y on two-term fp16 operands (section 10b).  Not together with pool_part. */
int sqd_bn_train_fwd_amax(const float *x, const float *res, const float *gamma, const float *beta, float *running_mean,
                          float *running_var, float *y, unsigned char *mask, float *save_mean, float *save_rstd, float *part,
                          int pre_rows, int M, int C, float eps, float momentum, int act, float *pool_part, int B, float *amax_y,
                          void *stream);
int sqd_bn_eval_fwd(const float *x, const float *res, const float *gamma, const float *beta, const float *running_mean,
                    const float *running_var, float *y, int M, int C, float eps, int act, void *stream);
/* dy, x, (y | mask: the activation's derivative; neither is read when act = 0) -> dx, dres (may be NULL), dgamma [C], dbeta [C].
 * act 3 (swish, the EfficientNet trunk): the derivative needs the pre-activation, recomputed from x with gamma and beta (beta may
 * be NULL for the other activations); no residual.                                                                          */
int sqd_bn_train_bwd(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                     const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                     float *dbeta, float *part, int M, int C, int act, void *stream);
/* the same with the reduction pass replaced by partials the producing data gradient wrote (sqd_conv_dgrad_bn): part holds pre_rows rows of
 * [C][2] = (sum dz, sum dz * xhat); pre_rows = 0: as sqd_bn_train_bwd                                                                  */
int sqd_bn_train_bwd_pre(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                         const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                         float *dbeta, float *part, int pre_rows, int M, int C, int act, void *stream);
/* ... and red_out[i] = sum_{s < red_splits} red_part[s * red_n + i] (sqd_split_reduce's arithmetic) as extra workgroups of the
 * finalize launch; red_part == NULL: none                                                                                        */
int sqd_bn_train_bwd_pre_red(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                             const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                             float *dbeta, float *part, int pre_rows, int M, int C, int act, const float *red_part, float *red_out,
                             int64_t red_n, int red_splits, void *stream);
/* ... and amax_dx / amax_dres (may be NULL; cleared by the caller): the bit patterns of max |dx| / max |dres| (section 10b) */
/* sqd_bn_train_bwd_amax that also takes the backward sums of the BatchNorm behind the residual branch (a bottleneck's down-sample BatchNorm: no
 * activation, this node its only consumer, so dres is its whole incoming gradient): x2 [M,C] its input, mean2 / rstd2 its saved statistics ->
 * part2 [sqd_bn_bwd_res_rows(M, C, pre_rows, act)][C][2] = (sum dres, sum dres * xhat2), precomputed partials for ITS sqd_bn_train_bwd_pre.
 * x2 = mean2 = rstd2 = part2 = NULL: none taken.  dx_colsum_part (may be NULL, independent of x2): [rows][C] partial column sums of dx — their sum
 * over the rows is the bias gradient of the convolution that produced x (for weight-gradient plans that leave none behind).
 * sqd_bn_bwd_res_rows = 0: the shape is not served (C / 4 must divide 256 on the three-launch path).                                                                                                                        */
int sqd_bn_bwd_res_rows(int M, int C, int pre_rows, int act);
int sqd_bn_train_bwd_res(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma, const float *beta,
                         const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma, float *dbeta, float *part,
                         int pre_rows, int M, int C, int act, const float *red_part, float *red_out, int64_t red_n, int red_splits,
                         float *amax_dx, float *amax_dres, const float *x2, const float *mean2, const float *rstd2, float *part2, float *dx_colsum_part,
                         void *stream);
int sqd_bn_train_bwd_amax(const float *dy, const float *x, const float *y, const unsigned char *mask, const float *gamma,
                          const float *beta, const float *save_mean, const float *save_rstd, float *dx, float *dres, float *dgamma,
                          float *dbeta, float *part, int pre_rows, int M, int C, int act, const float *red_part, float *red_out,
                          int64_t red_n, int red_splits, float *amax_dx, float *amax_dres, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (7) bilinear resize (align_corners=True) + channel concat, channels-last activations
 * replaces: F.interpolate(x, size=skip.shape[2:], bilinear, align_corners=True) + torch.cat([up, skip], 1)
 *           of UpSampleBN.forward (reference networks/resnet_encoder.py:114-117).
 * x [N,Hi,Wi,Cx], skip [N,Ho,Wo,Cs] -> out [N,Ho,Wo,Cx+Cs] (NHWC memory; Cx, Cs multiples of 4).      */
int sqd_upcat_fwd(const float *x, const float *skip, float *out, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs,
                  void *stream);
/* ... and amax_out (may be NULL; cleared by the caller): the bit pattern of max |out| (section 10b) */
int sqd_upcat_fwd_amax(const float *x, const float *skip, float *out, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs,
                       float *amax_out, void *stream);
int sqd_upcat_bwd(const float *g_out, float *g_x, float *g_skip, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs,
                  void *stream);
/* ... and, on the way, the two BatchNorm-backward sums of the node that produced x = act(BatchNorm(xb)) (a decoder stage's output: g_x is its whole
 * incoming gradient): xb [N,Hi,Wi,Cx], maskb its sign bits (NULL with act 0), meanb / rstdb [Cx], act 0 none / 1 ReLU / 2 LeakyReLU(0.01) ->
 * partb [sqd_upcat_bwd_bn_rows(...)][Cx][2] for its sqd_bn_train_bwd_pre (rows 0: shape not served).  xb = NULL: sqd_upcat_bwd.            */
int sqd_upcat_bwd_bn_rows(int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs);
int sqd_upcat_bwd_bn(const float *g_out, float *g_x, float *g_skip, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs, const float *xb,
                     const unsigned char *maskb, const float *meanb, const float *rstdb, int act, float *partb, void *stream);
/* ... and amax_gx (may be NULL; cleared by the caller): the bit pattern of max |g_x|, for the convolution backward that reads g_x on two-term fp16 operands */
int sqd_upcat_bwd_bn_amax(const float *g_out, float *g_x, float *g_skip, int N, int Hi, int Wi, int Cx, int Ho, int Wo, int Cs, const float *xb,
                          const unsigned char *maskb, const float *meanb, const float *rstdb, int act, float *partb, float *amax_gx, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (8) stand-alone forward kernels behind the reference's layer classes (not used by the training path,
 * which runs them fused in sqd_photo_fwd):
 *   BackprojectDepth.forward (layers.py:210-215): depth [B,1,H,W], inv_K [B,4,4] -> cam_points [B,4,HW]
 *   Project3D.forward (layers.py:247-258): points [B,4,HW], K, T [B,4,4] -> grid [B,H,W,2]
 *   SSIM.forward (layers.py:31-46): x, y [planes,H,W] -> clamp((1-SSIM)/2,0,1) [planes,H,W]
 * sqd_smooth_fwd with part == NULL serves get_smooth_loss (layers.py:267-280) on an already
 * normalised disparity.                                                                              */
int sqd_backproject_fwd(const float *depth, const float *inv_K, float *cam_points, int B, int H, int W, void *stream);
int sqd_project3d_fwd(const float *points, const float *K, const float *T, float *grid, int B, int H, int W, float eps,
                      void *stream);
int sqd_ssim_fwd(const float *x, const float *y, float *out, int planes, int H, int W, void *stream);
/* adjoints (the reference's modules are differentiable: layers.py:13-46,186-258).
 *   sqd_ssim_bwd: g [planes,H,W] upstream -> g_x, g_y [planes,H,W] (either may be NULL); coef_ws: 4*planes*H*W floats of workspace
 *   sqd_backproject_bwd: g_points [B,4,HW] -> g_depth [B,1,H,W] (inv_K is data)
 *   sqd_project3d_bwd: g_grid [B,H,W,2] -> g_points [B,4,HW], g_T [B,4,4] (K is data); gP_part: B * sqd_project3d_bwd_nblk(H,W) * 12
 *   floats of workspace (per-workgroup partials, summed in a fixed order)                                                           */
int sqd_ssim_bwd(const float *x, const float *y, const float *g, float *coef_ws, float *g_x, float *g_y, int planes, int H, int W,
                 void *stream);
int sqd_backproject_bwd(const float *g_points, const float *inv_K, float *g_depth, int B, int H, int W, void *stream);
int sqd_project3d_bwd_nblk(int H, int W);
int sqd_project3d_bwd(const float *points, const float *K, const float *T, const float *g_grid, float *g_points, float *gP_part,
                      float *g_T, int B, int H, int W, float eps, void *stream);
/* F.grid_sample(img [B,C,H,W], grid [B,Ho,Wo,2], padding_mode="border", align_corners=True) -> out [B,C,Ho,Wo]
 * (reference trainer.py:431-435); x0y0 (optional, int32 [B,Ho,Wo,2]) receives the integer north-west taps.          */
int sqd_grid_sample_border_fwd(const float *img, const float *grid, float *out, int *x0y0, int B, int C, int H, int W, int Ho,
                               int Wo, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (9) multi-tensor Adam step (one launch per parameter group)
 * replaces: torch.optim.Adam.step() of reference trainer.py:128-135,244 (default betas/eps, no weight
 *           decay, no amsgrad).  recs: device array of {float* p, float* exp_avg, float* exp_avg_sq,
 *           int64 n}; grads: device array of const float*; chunks: device array of int32 pairs
 *           (tensor index, chunk index), chunk = sqd_adam_chunk_elems() elements; step >= 1.         */
int sqd_adam_chunk_elems(void);
int sqd_adam_step(const void *recs, const void *grads, const void *chunks, int nchunks, double lr, double beta1,
                  double beta2, double eps, int step, void *stream);
/* graph-capturable form: the two step-dependent scalars live in device memory (hyper_dev[0] = lr / (1 - beta1^step),
 * hyper_dev[1] = 1 / sqrt(1 - beta2^step)); sqd_adam_hyper computes them on the host exactly as sqd_adam_step does. */
int sqd_adam_hyper(double lr, double beta1, double beta2, int step, float *hyper_host);
int sqd_adam_step_dev(const void *recs, const void *grads, const void *chunks, int nchunks, const float *hyper_dev,
                      double beta1, double beta2, double eps, void *stream);
/* ... that also leave max |p| of the updated tensors behind: amax_recs (may be NULL) = device array [ntensors] of pointers to
 * SQD_AMAX_RECORD_FLOATS-float records (section 10b), NULL per tensor that needs none; the records must have been cleared on the stream.
 * The filters' operand scales of the next step's two-term fp16 convolutions then need no pass over the weights (sqd_amax_multi).      */
int sqd_adam_step_amax(const void *recs, const void *grads, const void *chunks, int nchunks, double lr, double beta1,
                       double beta2, double eps, int step, const void *amax_recs, void *stream);
int sqd_adam_step_dev_amax(const void *recs, const void *grads, const void *chunks, int nchunks, const float *hyper_dev,
                           double beta1, double beta2, double eps, const void *amax_recs, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (10) convolution as implicit GEMM on the fp32 matrix cores, channels-last
 * replaces: nn.Conv2d forward / input gradient / weight+bias gradient of the networks' convolutions
 *           (reference networks/resnet_encoder.py:103-147, pose_cnn.py:14-29, depth_decoder_QTR.py:11-17
 *           and the torchvision ResNet trunk).
 * x [N,H,W,C], w [K,R,S,C] (KRSC = torch channels_last weight memory), y [N,Ho,Wo,K];
 * Ho = (H + 2*pad - R)/stride + 1.  forward needs C % 16 == 0; dgrad K % 16 == 0 and C % 4 == 0;
 * wgrad C % 4 == 0 and K % 4 == 0.  act (forward epilogue): 0 none, 1 ReLU.                          */
int sqd_conv_supported(int C, int K);
/* ws: split-K workspace of sqd_conv_plan(mode 0 = fwd / 1 = dgrad, ...) floats; NULL when the plan says 0 */
/* measured plans: see csrc/conv.hip — the library picks tile and split-K by a cost model unless the caller registers
 * a plan it has timed (bm x bn tile, z split-K factor, bk = 16 | 32 channels per reduction slice, + 256: 8-wave workgroups,
 * + 512: single LDS buffer, 32 + 1024: three-term bf16 operands on the bf16 matrix cores — every fp32 operand as the exact sum of
 * three bf16 terms, 6 of the 9 partial products (down to 2^-24 relative), fp32 accumulation, single LDS buffer;
 * 32 + 1024 + 2048: the input-patch kernel of that arithmetic for R = S = 3, stride 1, pad 1 (and the forward of R = S = 4, stride 1,
 * pad 2: the space-to-depth stems) — bm = 128 | 64 pixels of a patch
 * (8x16 | 4x16), bn = 128 | 64 | 32 output channels per workgroup, z = split over 32-channel chunks; + 256 on the three-term plans:
 * 8-wave workgroups (128x128, 64x128, forward 128x64 GEMM tiles; 3x3 input-patch tiles of >= 64 channels); bm = 0 clears) */
int sqd_conv_set_plan(int mode, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo, int bm,
                      int bn, int z, int bk);
/* arithmetic of sqd_conv_fwd / sqd_conv_dgrad: 0 = fp32 MFMA (default, the reference's arithmetic); 1 = split-precision bf16
 * MFMA — every fp32 operand as the exact sum of three bf16 terms, 6 of the 9 partial products (down to 2^-24 relative), fp32
 * accumulation; 2 = bf16 training arithmetic (BASELINE.json configs[3]): operands rounded to nearest-even bf16 as they are staged,
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 tensors in memory, BatchNorm / losses / weight gradients / Adam in fp32. */
int sqd_conv_set_precision(int prec);
int sqd_conv_precision(void);
int sqd_conv_plan(int mode, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                  int64_t *ws_floats);
/* stats (may be NULL): [sqd_conv_fwd_stats_rows(...)][K][2] per-channel (sum, sum of squares) partials of y for the BatchNorm that
 * follows (a plan that splits the reduction takes them in the sum over its splits: ceil(N*Ho*Wo / 64) rows; never more than
 * max(ceil(N*Ho*Wo / 64), N * ceil(Ho/4) * ceil(Wo/16)) rows — the second term: one row per patch of the input-patch plans)   */
int sqd_conv_fwd_stats_rows(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo);
int sqd_conv_fwd(const float *x, const float *w, const float *bias, float *y, float *ws, float *stats, int N, int H, int W,
                 int C, int K, int R, int S, int stride, int pad, int Ho, int Wo, int act, void *stream);
/* addend [N,H,W,C] or NULL: dx = dgrad + addend (the gradient arriving over a second path, e.g. the residual branch) */
/* gradient through the activation the forward applied in its epilogue, from the OUTPUT y: out = g * act'(y) (out may alias
 * g); act 1 ReLU, 2 LeakyReLU(0.01).  replaces the relu / leaky_relu backward nodes of autograd for these layers.           */
int sqd_act_bwd(const float *g, const float *y, float *out, int64_t n, int act, void *stream);
int sqd_conv_dgrad(const float *dy, const float *w, const float *addend, float *dx, float *ws, int N, int H, int W, int C,
                   int K, int R, int S, int stride, int pad, int Ho, int Wo, void *stream);
/* data gradient + the partial sums of the BatchNorm backward whose output act(BN(bn_x)) this convolution differentiates — replaces the
 * reduction pass of torch's batch_norm_backward over dy and x (dx = dgrad + addend must be the complete gradient of that output).
 * bn_x [N,H,W,C], bn_mask [N*H*W*C/4] sign bytes of the forward (NULL: no activation), bn_mean / bn_rstd [C], bn_act 0 | 1 | 2
 * -> stats [sqd_conv_dgrad_stats_rows(...)][C][2] for sqd_bn_train_bwd_pre (split plans: ceil(N*H*W / 64) rows, written by the sum over the splits). */
int sqd_conv_dgrad_stats_rows(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo);
int sqd_conv_dgrad_bn(const float *dy, const float *w, const float *addend, float *dx, float *ws, const float *bn_x,
                      const unsigned char *bn_mask, const float *bn_mean, const float *bn_rstd, int bn_act, float *stats, int N, int H,
                      int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo, void *stream);
/* workspace of the weight gradient: `part_floats` floats (+ max(ceil(N*Ho*Wo/1024), splits) * K more when dbias is wanted) */
int sqd_conv_wgrad_plan(int N, int Ho, int Wo, int C, int K, int R, int S, int *splits, int64_t *part_floats);
/* measured plan for the weight gradient: impl 1 (+ 16 kt + 256 ct: register tile) = direct-operand kernel, 0 = LDS-tiled kernel,
 * 2 / 3 + 16 * v = shared-operand kernel in fp32 MFMA / three-term bf16 arithmetic on blocks v = 0..3 of 128x128, 64x128, 128x64,
 * 64x64 filters x channels (K, C divisible by the block), 4 = row-window kernel (stride 1; K in {16,32,64} filters x C in {16,32,96}
 * channels x 3x3 / 4x4 taps as instantiated — others are refused; `splits` = workgroups, each holding the whole filter bank),
 * 6 + 16 * v = three-term bf16 operands straight from memory (the wave's halves take the even / odd pixel of a pair; register tiles
 * v = 0..5 of 64x64, 64x32, 32x64, 128x64, 64x128, 128x128 filters x channels; even Wo unless R = S = 1),
 * -1 = clear.  Plans are keyed by (N, Ho, Wo, C, K, R, S); a strided convolution of that key under an impl-4 / impl-6 plan it cannot run
 * takes impl 1 with the same splits. */
int sqd_conv_wgrad_set_plan(int N, int Ho, int Wo, int C, int K, int R, int S, int impl, int splits);
int sqd_conv_wgrad(const float *dy, const float *x, float *dw, float *dbias, float *part, int N, int H, int W, int C, int K,
                   int R, int S, int stride, int pad, int Ho, int Wo, void *stream);
/* impl & 15 of the kernel sqd_conv_wgrad launches for THIS convolution under the registered plan (a strided / padded layer that shares
 * the plan's output-geometry key may fall back to impl 1, see sqd_conv_wgrad_set_plan) — the plan timing discards such candidates */
int sqd_conv_wgrad_effective_impl(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo);
/* src [rows][cols] -> dst [cols][rows] (fp32); colsum (may be NULL): [ceil(rows/64)][cols] column sums of each 64-row tile.  The wide
 * 1x1 layers' weight gradient dW[k][c] = sum_m dY[m][k] X[m][c] as a FORWARD problem on transposed operands:
 * sqd_conv_fwd(x = dY^T as [1,1,K,M], w = X^T [C][M]) -> dW [K][C] on the three-term kernels; dbias from colsum. */
int sqd_transpose2d(const float *src, float *dst, int rows, int cols, float *colsum, void *stream);
/* sqd_conv_wgrad without the final sum over the pixel splits: part[0 .. *splits)[K*R*S*C] holds the partial filter gradients and dw
 * is NOT written (dbias is).  The caller adds them later on the same stream: sqd_split_reduce(part, dw, K*R*S*C, *splits), or as extra
 * workgroups of the next BatchNorm-backward finalize launch (sqd_bn_train_bwd_pre_red) — one launch less per layer, the same bits. */
int sqd_conv_wgrad_partials(const float *dy, const float *x, float *dw, float *dbias, float *part, int N, int H, int W, int C, int K,
                            int R, int S, int stride, int pad, int Ho, int Wo, int *splits, void *stream);
/* out[i] = sum_{s < splits} part[s * n + i], i < n (n a multiple of 4); fixed summation order                                      */
int sqd_split_reduce(const float *part, float *out, int64_t n, int splits, void *stream);

/* ---- (10b) two-term fp16 operands ("f16x2", round 5): plans bk = 32 + 1024 + 4096 (+ 2048, + 256) of sqd_conv_set_plan and impl
 * 7 + 16 * v of sqd_conv_wgrad_set_plan are the three-term plans / impl 6 with every fp32 operand staged as s^-1 (h + l): s a power of
 * two that puts the tensor's largest magnitude into [2^14, 2^15), h and l fp16 (round-to-nearest), products h h + h l + l h on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation — half the matrix and conversion instructions of the three-term bf16 split at the
 * accuracy of the fp32 MFMA chain (csrc/conv.hip; tests/test_gpu_conv.py).  Such a plan needs max |.| of both operand tensors:
 * `amax_*` = device RECORDS of SQD_AMAX_RECORD_FLOATS floats whose words 0, 16, 32, .. hold bit patterns of non-negative floats, the tensor's
 * max |.| being their maximum (any upper bound is safe), written before the call on the same stream — by sqd_amax / sqd_amax_multi or by the `amax` outputs of the producing kernels.  The *_scaled entry points are the
 * plain ones + those scalars; plans of the other arithmetics ignore them (NULL allowed), a two-term plan without them is SQD_EINVAL.
 * replaces: nothing in the reference — its convolutions multiply in fp32 (networks/resnet_encoder.py:89-147). */
#define SQD_AMAX_WAYS 64            /* a max |.| record: 64 words, one per 64-byte line — the recording kernels spread their atomics over them  */
#define SQD_AMAX_RECORD_FLOATS 1024 /* ... = 4 KB; the value of the record is the maximum of words 0, 16, 32, ..., 1008 (csrc/sqd_common.h)        */
int sqd_amax(const float *x, int64_t n, float *amax, void *stream);
/* one record per parameter tensor of an optimiser table (recs / chunks of sqd_adam_step): amax + t * SQD_AMAX_RECORD_FLOATS = max |p_t| */
int sqd_amax_multi(const void *recs, const void *chunks, int nchunks, int ntensors, float *amax, void *stream);
/* amax_y / amax_dx (every plan; may be NULL; a scalar the caller cleared on the stream): the epilogue (or the sum over the splits) records the bit
 * pattern of max |output| there — the operand scale of the next convolution in a chain without BatchNorm (PoseCNN, the decoder's plain layers) */
int sqd_conv_fwd_scaled(const float *x, const float *w, const float *bias, float *y, float *ws, float *stats, const float *amax_x,
                        const float *amax_w, float *amax_y, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho,
                        int Wo, int act, void *stream);
/* sqd_conv_dgrad_bn (stats == NULL: sqd_conv_dgrad) + the operand scalars */
int sqd_conv_dgrad_scaled(const float *dy, const float *w, const float *addend, float *dx, float *ws, const float *bn_x,
                          const unsigned char *bn_mask, const float *bn_mean, const float *bn_rstd, int bn_act, float *stats,
                          const float *amax_dy, const float *amax_w, float *amax_dx, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int Ho, int Wo, void *stream);
/* sqd_act_bwd + amax_out (may be NULL; cleared by the caller): the bit pattern of max |out| */
int sqd_act_bwd_amax(const float *g, const float *y, float *out, int64_t n, int act, float *amax_out, void *stream);
/* sqd_conv_wgrad (splits == NULL) / sqd_conv_wgrad_partials (splits != NULL) + the operand scalars */
int sqd_conv_wgrad_scaled(const float *dy, const float *x, float *dw, float *dbias, float *part, const float *amax_dy,
                          const float *amax_x, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int Ho, int Wo,
                          int *splits, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * (11) adaptive-bins depth head.  replaces reference networks/depth_decoder_QTR.py:61-70 (convert_to_prob =
 * Conv2d(Q, D, 1) + Softmax(dim=1), then pred = sum_d out[d] * centers[b, d]).
 * energy [B,Q,N] (the planar energy maps the Self Query Layer writes, N = h*w), weight [D,Q] (the 1x1 filter),
 * bias [D], centers [B,D] -> pred [B,N].  1 <= Q, D <= 128.
 * bwd: g_pred [B,N] -> g_energy [B,Q,N], g_weight [D,Q], g_bias [D], g_centers [B,D];
 *      part: workspace of sqd_bins_workspace floats.  Deterministic (fixed-order partial sums).
 * Arithmetic (sqd_bins_set_arith, process-wide, default 1): 1 = the logits and the g_energy product on two-term fp16 operands
 * (v_mfma_f32_32x32x16_f16, three instructions per product, fp32 accumulation: csrc/sqd_f16x2.h — error against float64 at the level of
 * the fp32 instruction's), the g_weight product on v_mfma_f32_32x32x2_f32; 0 = the fp32 matrix instruction throughout (rounds 1-4). */
int sqd_bins_supported(int Q, int D);
int sqd_bins_set_arith(int arith);
int sqd_bins_workspace(int B, int Q, int D, int N, int64_t *part_floats);
int sqd_bins_fwd(const float *energy, const float *weight, const float *bias, const float *centers, float *pred, int B, int Q,
                 int D, int N, void *stream);
int sqd_bins_bwd(const float *energy, const float *weight, const float *bias, const float *centers, const float *g_pred,
                 float *g_energy, float *g_weight, float *g_bias, float *g_centers, float *part, int B, int Q, int D, int N,
                 void *stream);
/* bin centres from the regressor's raw outputs (norm "linear"; reference networks/depth_decoder_QTR.py:56-66 + the
 * pad / cumsum / mid-point arithmetic of :62-66): v = relu(y) + 0.1, w = v / sum(v), edges = cumsum([vmin, (vmax-vmin) w]),
 * centers = mid-points.  y, centers, g_* [B,D], sums [B] (saved for the backward); D <= 128. */
int sqd_bin_centers_fwd(const float *y, float *centers, float *sums, int B, int D, float vmin, float vmax, void *stream);
int sqd_bin_centers_bwd(const float *y, const float *sums, const float *g_centers, float *g_y, int B, int D, float vmin, float vmax,
                        void *stream);

/* ---------------------------------------------------------------------------------------------------
 * (12) MaxPool2d(3, stride 2, padding 1), channels-last.  replaces: the ResNet stem's maxpool (reference
 * networks/resnet_encoder.py:96).  x [N,H,W,C] -> y [N,Ho,Wo,C] and idx [N,Ho,Wo,C] (1 byte: window position of the
 * maximum, ATen's tie rule); Ho = (H-1)/2+1, Wo = (W-1)/2+1; C multiple of 4.  The backward is a gather (no atomics); addend
 * [N,H,W,C] (may be NULL) is added to it: the gradient arriving from the input's other consumer (the decoder's skip). */
int sqd_maxpool3x3s2_fwd(const float *x, float *y, unsigned char *idx, int N, int H, int W, int C, void *stream);
int sqd_maxpool3x3s2_bwd(const float *dy, const unsigned char *idx, const float *addend, float *dx, int N, int H, int W, int C,
                         void *stream);
/* ... and, on the way, the two BatchNorm-backward sums of the node that produced the pooled tensor x = act(BatchNorm(xb)) (the stem's bn1 + ReLU:
 * dx is its whole incoming gradient): xb [N,H,W,C], maskb the sign bits its forward stored (NULL with act 0), meanb / rstdb [C], act 0 none /
 * 1 ReLU / 2 LeakyReLU(0.01) -> partb [sqd_maxpool3x3s2_bwd_bn_rows(N,H,W,C)][C][2] for its sqd_bn_train_bwd_pre (rows 0: shape not served). */
int sqd_maxpool3x3s2_bwd_bn_rows(int N, int H, int W, int C);
int sqd_maxpool3x3s2_bwd_bn(const float *dy, const unsigned char *idx, const float *addend, float *dx, int N, int H, int W, int C,
                            const float *xb, const unsigned char *maskb, const float *meanb, const float *rstdb, int act, float *partb,
                            void *stream);
/* space-to-depth(2), channels-last, channels zero-padded to Cp: y[n,h2,w2,c*4+dy*2+dx] = x[n,2h2+dy,2w2+dx,c].  The 7x7/2 stems
 * (reference networks/resnet_encoder.py:94, pose_cnn.py:17) run as 4x4/1 convolutions on this layout (sqd_conv_*).       */
int sqd_space_to_depth2(const float *x, float *y, int N, int H, int W, int C, int Cp, void *stream);
/* the same from planar sources x0 [N,C0,H,W], x1 [N,C1,H,W] (or NULL), with the frame staging of the two stems folded in: channel
 * concatenation (x0, x1) (the pose network's torch.cat of a frame pair, reference trainer.py:319-326), every value
 * (v - sub) * (1.0f / div) — the encoder's (x - 0.45) / 0.225 (reference networks/resnet_encoder.py:93) as ATen evaluates a tensor
 * divided by a scalar on the device: bit-identical to the reference's normalised frame; sample n is written at y + n * y_stride floats */
int sqd_space_to_depth2_planar(const float *x0, const float *x1, float *y, int N, int H, int W, int C0, int C1, int Cp,
                               int64_t y_stride, float sub, float div, void *stream);
/* ... and amax_y (may be NULL; cleared by the caller, once for all launches that fill one batch): the bit pattern of max |y| (section 10b) */
int sqd_space_to_depth2_planar_amax(const float *x0, const float *x1, float *y, int N, int H, int W, int C0, int C1, int Cp,
                                    int64_t y_stride, float sub, float div, float *amax_y, void *stream);
/* the matching filter regrouping w [K,C,7,7] -> ws [K,4,4,Cp] (tap u = 2r' + dy - 1; adjoint = 1: g_ws -> g_w, fully overwritten) */
int sqd_stem_regroup(const float *src, float *dst, int K, int C, int Cp, int adjoint, void *stream);
/* ... with the 7x7 filter (adjoint: its gradient) in channels-last memory [K,7,7,C] when w_channels_last != 0 — no layout copy on either side */
int sqd_stem_regroup_ex(const float *src, float *dst, int K, int C, int Cp, int adjoint, int w_channels_last, void *stream);

/* patch tokens + positional encodings (reference networks/depth_decoder_QTR.py:49-51): emb [B,T,E] (the embedding convolution's channels-last
 * output), pos [Tmax,E] -> out [T,B,E] = emb + pos[:T]; backward: g [T,B,E] -> g_emb [B,T,E], g_pos [Tmax,E] (rows >= T zero) */
int sqd_tokens_pos_fwd(const float *emb, const float *pos, float *out, int B, int T, int E, void *stream);
int sqd_tokens_pos_bwd(const float *g, float *g_emb, float *g_pos, int B, int T, int E, int Tmax, void *stream);
/* the first Q tokens as the query matrix (reference networks/depth_decoder_QTR.py:52): src [T,B,E] -> dst [B,Q,E]; adjoint != 0: src [B,Q,E] ->
 * dst [T,B,E] with zero rows for t >= Q */
int sqd_first_queries(const float *src, float *dst, int T, int B, int Q, int E, int adjoint, void *stream);
/* out [n] = (add ? add[n] : 0) + sum_k parts[k][n] in index order (n a multiple of 4) */
int sqd_sum_parts(const float *parts, const float *add, float *out, int nparts, int64_t n, void *stream);
/* ---------------------------------------------------------------------------------------------------
 * (13) token-wise blocks of the post-norm TransformerEncoderLayer over the patch tokens.  replaces: the feed-forward and the
 * two add+dropout+LayerNorm steps of nn.TransformerEncoderLayer as the reference builds it at
 * networks/depth_decoder_QTR.py:31-32 (d_model E in {16,32}, 4 heads, dim_feedforward F = 1024 | 512, ReLU, dropout 0.1)
 * and runs it at :47.  Tokens are the rows of [rows = S*B, E] matrices.  Dropout masks are bytes (1 = keep) drawn by the
 * caller, NULL = no dropout; scale = 1/(1-p).
 *   sqd_addln_fwd: out = LayerNorm_E(x + mask*scale*(sum_p y[p] + ybias)) * gamma + beta; y [nparts][rows,E], ybias [E] or
 *                  NULL; saves xhat [rows,E], rstd [rows]
 *   sqd_addln_bwd: g = g_out + sum_p g_extra[p] (g_extra [nextra][rows,E]) -> g_x, g_y [rows,E];
 *                  part: sqd_addln_nblk(rows) x [2][E] partials of (g_gamma, g_beta) for sqd_colsum_multi
 *   sqd_ffn_fwd:   ypart [G][rows,E], G = sqd_ffn_groups(F) partials over groups of 128 hidden units of
 *                  W2 . (mask*scale*relu(W1 . x + b1)); W1 [F,E], W2 [E,F], mask [rows,F] (4-byte aligned).  The consumer
 *                  (sqd_addln_fwd with nparts = G, ybias = b2) adds them.
 *   sqd_ffn_bwd:   g_y -> gxpart [G][rows,E] (g_x = sum over G; sqd_addln_bwd's g_extra) and per-token-tile partials,
 *                  T = sqd_ffn_tiles(rows): pW1 [T][F,E], pW2T [T][F,E] (g_W2 transposed), pb1 [T][F], pb2 [T][E].
 *                  Hidden activations are recomputed.
 *   sqd_colsum_multi: up to 12 column sums in one launch: dst[s][c] = sum_{r<nrows[s]} src[s][r*ncols[s] + c], ncols % 4 == 0;
 *                  tr[s] > 0 writes the [ncols/tr][tr] result transposed (pW2T -> g_W2 with tr = E).  The arrays are host arrays.
 * fp32 MFMA (v_mfma_f32_32x32x2_f32); deterministic (fixed-order partial sums). */
int sqd_vit_supported(int E, int F);
int sqd_addln_fwd(const float *x, const float *y, int nparts, const float *ybias, const unsigned char *mask, const float *gamma,
                  const float *beta, float *out, float *xhat, float *rstd, int rows, int E, float scale, float eps, void *stream);
int sqd_addln_nblk(int rows);
int sqd_addln_bwd(const float *g_out, const float *g_extra, int nextra, const float *xhat, const float *rstd,
                  const unsigned char *mask, const float *gamma, float *g_x, float *g_y, float *part, int rows, int E, float scale,
                  void *stream);
int sqd_ffn_groups(int F);
int sqd_ffn_tiles(int rows);
int sqd_ffn_fwd(const float *x, const float *W1, const float *b1, const float *W2, const unsigned char *mask, float *ypart, int rows,
                int E, int F, float scale, void *stream);
int sqd_ffn_bwd(const float *x, const float *g_y, const float *W1, const float *b1, const float *W2, const unsigned char *mask,
                float *gxpart, float *pW1, float *pb1, float *pW2T, float *pb2, int rows, int E, int F, float scale, void *stream);
int sqd_colsum_multi(const float *const *src, float *const *dst, const int *nrows, const int *ncols, const int *tr, int nseg,
                     void *stream);

/* ---------------------------------------------------------------------------------------------------
 * (14) multi-head self-attention of the encoder layer for short sequences.  replaces: nn.MultiheadAttention (packed in_proj,
 * batch_first = False, need_weights = False) inside the nn.TransformerEncoderLayer of reference
 * networks/depth_decoder_QTR.py:31-32 — in-projection, softmax(q k^T / sqrt(hd)), attention dropout, P v, out-projection.
 * S <= 512 tokens, E in {16,32}, head dimension E/H in {4,8}.  x [S*B,E], token (s,b) = row s*B+b; Win [3E,E], bin [3E],
 * Wo [E,E]; mask: keep bytes [B][H][S][SP], SP = S rounded up to 4 (4-byte aligned), or NULL; dscale = 1/(1-p).
 *   fwd: ypart [H][S*B,E] per-head partials of the block's output without the out-projection bias (the consumer,
 *        sqd_addln_fwd with nparts = H and ybias = out_proj.bias, adds them); o_save [B][H][S][E/H], ml_save [B][H][S][2]
 *   bwd: g_sa [S*B,E] -> gxpart [H][S*B,E] (g_x = sum over heads) and per-batch-element partials pWin [B][3E,E], pbin [B][3E],
 *        pWo [B][E,E], pbo [B][E] for sqd_colsum_multi.  Deterministic. */
int sqd_mha_supported(int S, int E, int H);
int sqd_mha_fwd(const float *x, const float *Win, const float *bin, const float *Wo, const unsigned char *mask, float *ypart,
                float *o_save, float *ml_save, int S, int B, int E, int H, float dscale, void *stream);
int sqd_mha_bwd(const float *x, const float *g_sa, const float *Win, const float *bin, const float *Wo, const unsigned char *mask,
                const float *o_save, const float *ml_save, float *gxpart, float *pWin, float *pbin, float *pWo, float *pbo, int S,
                int B, int E, int H, float dscale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * EfficientNet-b5 trunk operators (reference networks/base_encoder.py:76-107 loads tf_efficientnet_b5_ap from torch.hub —
 * third-party; the operators below are what its MBConv blocks add to the ResNet path)
 * depthwise k x k convolution, k in {3,5}, stride in {1,2}, explicit top / left padding (TensorFlow "SAME": pad_total =
 * max((ceil(H/stride) - 1) * stride + k - H, 0), top = pad_total / 2), channels-last, C % 4 == 0.  Filters travel tap-major
 * [k*k][C] (sqd_dw_weight_layout converts from / to torch's [C,1,k,k]).  wgrad leaves partials [chunks][k*k][C]
 * (chunks = sqd_dw_conv_wgrad_chunks) for sqd_colsum_multi.                                                            */
int sqd_dw_weight_layout(const float *src, float *dst, int C, int k, int to_taps, void *stream);
int sqd_dw_conv_fwd(const float *x, const float *w_taps, float *y, int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                    int Ho, int Wo, void *stream);
int sqd_dw_conv_dgrad(const float *dy, const float *w_taps, float *dx, int N, int H, int W, int C, int k, int stride, int pad_t,
                      int pad_l, int Ho, int Wo, void *stream);
/* ... + addend [N,H,W,C] (NULL: none; stride 1): the gradient arriving over the input's other path (a block's shortcut) added in the kernel */
int sqd_dw_conv_dgrad_add(const float *dy, const float *w_taps, const float *addend, float *dx, int N, int H, int W, int C, int k, int stride,
                          int pad_t, int pad_l, int Ho, int Wo, void *stream);
int sqd_dw_conv_wgrad_chunks(int N, int Ho, int Wo);
int sqd_dw_conv_wgrad(const float *dy, const float *x, float *part, int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                      int Ho, int Wo, void *stream);
/* squeeze-and-excite: gate[b,c] = sigmoid(W2 . swish(W1 . mean_hw(x[b]) + b1) + b2), y = x * gate.
 * sqd_se_pool: per-chunk channel sums of a (or a * b) over the pixels -> part [B][sqd_se_chunks(HW)][C];
 * sqd_se_gate_fwd: part, W1 [R,C], b1 [R], W2T [R,C] (the [C,R] expansion filter transposed), b2 [C] -> s [B,C] (pooled mean),
 * pre1 [B,R], gate [B,C];
 * sqd_se_gate_bwd: dgpart = sqd_se_pool(dy, x) -> per-image partials dW1part [B,R,C], db1part [B,R rounded up to 4], dW2part [B,C,R], db2part [B,C]
 * and ds [B,C] (gradient w.r.t. the pooled mean, already divided by HW); sqd_se_scale: y = x * gate (+ ds: the backward).          */
int sqd_se_chunks(int HW);
int sqd_se_pool(const float *a, const float *b, float *part, int B, int HW, int C, void *stream);
int sqd_se_gate_fwd(const float *part, const float *W1, const float *b1, const float *W2T, const float *b2, float *s, float *pre1,
                    float *gate, int B, int HW, int C, int R, void *stream);
int sqd_se_gate_bwd(const float *dgpart, const float *W1, const float *W2T, const float *s, const float *pre1, const float *gate,
                    float *dW1part, float *db1part, float *dW2part, float *db2part, float *ds, int B, int HW, int C, int R, void *stream);
int sqd_se_scale(const float *x, const float *gate, const float *ds, float *y, int B, int HW, int C, void *stream);

/* ------------------------------------------------------------------------------------------------
 * PoseCNN tail
 * replaces: out = self.pose_conv(out); out = out.mean(3).mean(2); out = 0.01 * out.view(-1, F, 1, 6); axisangle = out[..., :3];
 *           translation = out[..., 3:]                                                  reference networks/pose_cnn.py:40-45
 * x [B,h,w,C] channels-last, W [J,C] (the 1x1 filter), bias [J], J <= 16; mean [B,C] is kept for the backward.
 * out2 == NULL: out [B,J] = scale * (W . mean_hw(x) + bias).  out2 != NULL (J = 6 F): the same numbers split the reference's way into
 * two dense tensors, out = axisangle [B,F,3] and out2 = translation [B,F,3] (what sqd_pose_mats_fwd reads).
 * backward: g (and g2, the forward's layouts) -> dx [B,h,w,C], per-image partials dWpart [B,J,C], dbpart [B,J rounded up to a
 * multiple of 4] (summed over B by sqd_colsum_multi; needs C % 4 == 0).                                                   */
int sqd_pose_head_fwd(const float *x, const float *W, const float *bias, float *out, float *out2, float *mean, int B, int h, int w,
                      int C, int J, float scale, void *stream);
int sqd_pose_head_bwd(const float *g, const float *g2, const float *W, const float *mean, float *dx, float *dWpart, float *dbpart,
                      int B, int P, int C, int J, float scale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Depth evaluation (not on the training step)
 * replaces: batch_post_process_disparity (reference evaluate_depth_config.py:50-59) and the per-image body of evaluate()
 *           (:225-261: cv2.resize to the ground truth's size, Garg / Eigen crop, scale factor, median scaling, clamp,
 *           compute_errors :30-47), in double precision on the device.
 * sqd_disp_post_process: disp [2N,h,w] fp32 = outputs of N images followed by those of their flipped copies -> out [N,h,w] fp64.
 * sqd_depth_eval: pred [h,w] fp64 (the head's depth output), gt [Hg,Wg] fp32 -> out [9] fp64: abs_rel, sq_rel, rmse, rmse_log,
 *   a1, a2, a3, median-scaling ratio (NaN when disabled), valid pixels (0: the metrics are NaN).  eigen_crop 1: valid =
 *   min_depth < gt < max_depth inside the crop; 0: gt > 0.  One launch per image (ground-truth sizes differ between images).    */
int sqd_disp_post_process(const float *disp, double *out, int N, int h, int w, void *stream);
int sqd_depth_eval(const double *pred, int h, int w, const float *gt, int Hg, int Wg, int eigen_crop, double min_depth,
                   double max_depth, double pred_scale, int median_scaling, double *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side input pipeline (not on the training step proper)
 * replaces: MonoDataset.__getitem__ / preprocess per frame (reference datasets/mono_dataset.py:90-201) — PIL's
 *           Image.resize(..., ANTIALIAS) (:57,73-78), the left-right flip (:163), torchvision's ColorJitter on PIL images
 *           (:64-71,179-183) and ToTensor (:107-108) — byte-exact with Pillow's C arithmetic (Resample.c, Blend.c, Convert.c).
 * Frames are [n,H,W,3] bytes (HWC, as decoded).  Resize = sqd_resample_h_u8 then sqd_resample_v_u8 with the caller's
 * coefficient tables (Pillow's precompute_coeffs + normalize_coeffs_8bpc for the Lanczos filter: bounds [out,2] int32 = first
 * source index and tap count, coef [out,ksize] int32 in 22-bit fixed point).  ColorJitter = up to four sqd_color_jitter_step_u8
 * launches (op[f]: 0 brightness, 1 contrast, 2 saturation, 3 hue, other: copy), the contrast step after sqd_luma_sum_u8 on its
 * input.  sqd_u8_to_chw_f32: [n,H,W,3] bytes -> [n,3,H,W] float = v / 255.                                                  */
int sqd_resample_h_u8(const unsigned char *in, unsigned char *out, const int *bounds, const int *coef, int ksize, int n, int H0, int W0,
                      int W, const unsigned char *flip, void *stream);
int sqd_resample_v_u8(const unsigned char *in, unsigned char *out, const int *bounds, const int *coef, int ksize, int n, int H0, int H,
                      int W, void *stream);
int sqd_luma_sum_u8(const unsigned char *img, unsigned long long *sums, int n, int HW, void *stream);
int sqd_color_jitter_step_u8(const unsigned char *in, unsigned char *out, const int *op, const float *factor, const int *hshift,
                             const unsigned long long *lsum, int n, int HW, void *stream);
int sqd_u8_to_chw_f32(const unsigned char *in, float *out, int n, int HW, void *stream);

/* ------------------------------------------------------------------------------------------------
 * ConvNeXt-L trunk + U-Net decoder operators (config E; reference networks/Unet.py:9-312 builds the encoder with
 * timm.create_model('convnext_large', features_only=True)).  Channels-last activations as rows [M = N*H*W, C], C % 4 == 0.
 * sqd_ln_rows_*: LayerNorm over the channels of every row (timm LayerNorm2d / the block's nn.LayerNorm, eps 1e-6); pre_bias [C] or
 *   NULL is added to x first (the bias of the depthwise convolution in front of it).  C <= 2048.  mean, rstd [M] are saved for the
 *   backward, whose part [sqd_ln_rows_nblk(M)][3][C] holds per-block column sums of dy*xhat, dy, dx (-> dgamma, dbeta, d pre_bias).
 * sqd_gelu_*: exact (erf) GELU.  sqd_scale_residual_*: out = res + gamma[c] * z (layer scale + shortcut); backward dz = gamma * dy
 *   and part [sqd_scale_residual_nblk(M)][C] = per-block column sums of dy * z (-> dgamma).
 * sqd_upsample2x_*: F.interpolate(scale_factor=2, mode='bilinear') (align_corners False) of Unet.py:250, [N,H,W,C] -> [N,2H,2W,C].
 * The 7x7 depthwise convolution of the block is sqd_dw_conv_* (k in {3,5,7}).                                              */
int sqd_ln_rows_fwd(const float *x, const float *pre_bias, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                    int M, int C, float eps, void *stream);
int sqd_ln_rows_nblk(int M);
int sqd_ln_rows_bwd(const float *dy, const float *x, const float *pre_bias, const float *gamma, const float *mean, const float *rstd,
                    float *dx, float *part, int M, int C, void *stream);
int sqd_gelu_fwd(const float *x, float *y, int64_t n, void *stream);
int sqd_gelu_bwd(const float *x, const float *dy, float *dx, int64_t n, void *stream);
int sqd_scale_residual_fwd(const float *res, const float *z, const float *gamma, float *out, int M, int C, void *stream);
int sqd_scale_residual_nblk(int M);
int sqd_scale_residual_bwd(const float *dy, const float *z, const float *gamma, float *dz, float *part, int M, int C, void *stream);
int sqd_upsample2x_fwd(const float *x, float *y, int N, int H, int W, int C, void *stream);
int sqd_upsample2x_bwd(const float *dy, float *dx, int N, int H, int W, int C, void *stream);
/* The same with the bit pattern of max |output| recorded by the pass that writes the tensor (amax_* may be NULL; a record of section 10b,
 * cleared by the caller on the stream before the call): the operand scales of the block's two Linear layers when they run on two-term fp16
 * operands (sqd_conv_fwd_scaled / _dgrad_scaled / _wgrad_scaled) — LayerNorm output and GELU output forward, GELU's dx and the layer
 * scale's dz backward — without a sqd_amax pass over each tensor.                                                            */
int sqd_ln_rows_fwd_amax(const float *x, const float *pre_bias, const float *gamma, const float *beta, float *y, float *mean, float *rstd,
                         int M, int C, float eps, float *amax_y, void *stream);
int sqd_gelu_fwd_amax(const float *x, float *y, int64_t n, float *amax_y, void *stream);
int sqd_gelu_bwd_amax(const float *x, const float *dy, float *dx, int64_t n, float *amax_dx, void *stream);
int sqd_scale_residual_bwd_amax(const float *dy, const float *z, const float *gamma, float *dz, float *part, int M, int C, float *amax_dz,
                                void *stream);
/* ... and the bias gradients of the block's two Linear layers from the passes that write their output gradients: part2 / part
 * [sqd_scale_residual_nblk(M)][C] = per-block column sums of dz (layer scale backward; part2 may be NULL) and of dx (GELU backward on rows
 * [M,C]); summed over the blocks (sqd_colsum_multi) they are dbias, and the Linear layer's weight-gradient call takes dbias = NULL. */
int sqd_scale_residual_bwd_sums(const float *dy, const float *z, const float *gamma, float *dz, float *part, float *part2, int M, int C,
                                float *amax_dz, void *stream);
int sqd_gelu_bwd_rows(const float *x, const float *dy, float *dx, float *part, int M, int C, float *amax_dx, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Supervised metric-depth finetune step (config E; not part of the self-supervised step)
 * replaces: reference finetune/train_ft_SQLdepth.py:219-285 per step — interpolate(pred, depth size, bilinear, align_corners=True)
 *           (:233), the per-sample median rescale computed with numpy on the host (:234-266), SILogLoss (finetune/loss.py:24-42),
 *           nn.utils.clip_grad_norm_(params, 0.1) (:281) and optim.AdamW.step() (:184,282).
 * sqd_resize_ac_*: x [B,1,h,w] -> y [B,1,H,W] and its adjoint (rowscale [B] or NULL multiplies each sample's gradient).
 * sqd_median_ratio: pred, depth [B,H,W] -> ratio [nscale] = median(depth[valid]) / median(pred[valid]) for the first nscale samples,
 *   valid = min_eval < depth < max_eval inside the crop (0 none, 1 Garg, 2 Eigen-KITTI); exact medians, float32 as numpy; 1 when
 *   nothing is valid.
 * sqd_silog_*: g = log(scale[b] * pred) - log(depth) where depth > min_depth; loss = 10 sqrt(var(g) + 0.15 mean(g)^2), unbiased
 *   variance; part: 3 * sqd_silog_nblk(B*HW) doubles; stats [4] = (valid pixels, mean, Dg, loss); backward -> d loss / d pred.
 * sqd_grad_sumsq: part [nchunks] = per-chunk sums of squares of the gradients of an Adam table; sqd_clip_coef: part [n] (of one or
 *   several tables) -> coef_norm [2] = (min(1, max_norm / (L2 norm + 1e-6)), norm).
 * sqd_adamw_step: torch.optim.AdamW on the tables of sqd_adam_step; gscale (device float or NULL) multiplies every gradient.   */
int sqd_resize_ac_fwd(const float *x, float *y, int B, int h, int w, int H, int W, void *stream);
int sqd_resize_ac_bwd(const float *dy, const float *rowscale, float *dx, int B, int h, int w, int H, int W, void *stream);
int sqd_median_ratio(const float *pred, const float *depth, float *ratio, int nscale, int H, int W, float min_eval, float max_eval,
                     int crop, void *stream);
/* pred, depth [B,H,W] float32, the prediction already at the ground truth's size -> out [B][11] doubles: a1, a2, a3, abs_rel, rmse, log_10,
 * rmse_log, silog, sq_rel, median ratio, valid pixels (metrics NaN when 0): the per-image body of the reference's validation loop
 * (finetune/train_ft_SQLdepth.py:347-375: valid = min_eval < depth < max_eval inside the crop, pred *= median(gt) / median(pred) with exact
 * float32 medians, clamp to [min_eval, max_eval]; finetune/utils.py:76-96 compute_errors — per-pixel terms in float32 as numpy forms them,
 * sums in float64).  crop: 0 none, 1 Garg, 2 Eigen (KITTI), 3 Eigen (NYU, rows 45..470 x columns 41..600).  median_scaling = 0: the
 * body of finetune/evaluate_metric_depth.py:65-141 instead (no scaling, no clamps, inf -> max_eval, nan -> min_eval, ratio = 1).  */
int sqd_metric_depth_eval(const float *pred, const float *depth, double *out, int B, int H, int W, float min_eval, float max_eval,
                          int crop, int median_scaling, void *stream);
int sqd_silog_nblk(int64_t total);
int sqd_silog_fwd(const float *pred, const float *depth, const float *scale, double *part, float *stats, int B, int HW, float min_depth,
                  void *stream);
int sqd_silog_bwd(const float *pred, const float *depth, const float *scale, const float *stats, const float *g_loss, float *dpred, int B,
                  int HW, float min_depth, void *stream);
int sqd_grad_sumsq(const void *recs, const void *grads, const void *chunks, int nchunks, float *part, void *stream);
int sqd_clip_coef(const float *part, int n, double max_norm, float *coef_norm, void *stream);
int sqd_adamw_step(const void *recs, const void *grads, const void *chunks, int nchunks, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int step, const float *gscale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * (19) gradient exchange across the GPUs of a node: a RCCL communicator owned by the library
 * replaces: nn.DataParallel's gather of the replicas' gradients onto GPU 0       reference trainer.py:73-74,92-93
 *           (one process per GPU, bucketed all-reduce overlapped with backward)
 * The collectives are plain stream operations on the caller's stream: they order like kernels, they can be
 * captured into a hipGraph like kernels, and no helper thread ever touches the caller's events.
 * librccl is bound at run time (dlopen): sqd_comm_load(path) picks the instance — a PyTorch process passes
 * torch/lib/librccl.so so that RCCL and the caller share ONE HIP runtime; NULL means "librccl.so" on the
 * loader's search path.  Everything else in this header works without RCCL present.
 * Rendezvous stays with the caller: rank 0 calls sqd_comm_unique_id, ships the SQD_COMM_ID_BYTES to the
 * other ranks over whatever channel the job already has (torchrun's store here), then every rank calls
 * sqd_comm_init (collective; the calling thread's current HIP device is the rank's GPU).
 * dtype: 0 = f32, 1 = f64, 2 = i32, 3 = u8;  op: 0 = sum, 1 = average, 2 = max, 3 = min.                 */
#define SQD_COMM_ID_BYTES 128
#define SQD_ECOMM (-3)    /* RCCL returned an error (text in sqd_last_error) or librccl could not be bound */
typedef struct sqd_comm sqd_comm;
int sqd_comm_load(const char *librccl_path_host);
int sqd_comm_unique_id(void *id_host);
int sqd_comm_init(const void *id_host, int rank, int world, sqd_comm **comm_host);
int sqd_comm_rank(const sqd_comm *comm);
int sqd_comm_world(const sqd_comm *comm);
/* reporting only (bench line): NCCL-style version code of the bound librccl (22703 = 2.27.3; 0 = not exported by it), and the
 * number of ranks RCCL itself counts in the communicator (ncclCommCount)                                                     */
int sqd_comm_rccl_version(int *version_host);
int sqd_comm_joined(const sqd_comm *comm, int *count_host);
int sqd_comm_allreduce(sqd_comm *comm, void *buf, int64_t count, int dtype, int op, void *stream);
int sqd_comm_broadcast(sqd_comm *comm, void *buf, int64_t count, int dtype, int root, void *stream);
int sqd_comm_destroy(sqd_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* SQD_H_ */
