// photo_launch.h — what photometric.hip (the C ABI entry points) calls in photo_tile.hip (the tile kernels).  Kept out of sqd_common.h: the pinned
// convolution plans are keyed to the hash of conv.hip + sqd_common.h, and work on the photometric kernels must not stale them.
#pragma once
#include "sqd_common.h"

namespace sqd {
// photo_tile.hip: fused warp+SSIM forward (mode 1) / identity maps (mode 0) / coefficient planes for the backward (mode 2)
int launch_photo_tile(const sqd_photo_args &a, const float *noise, int mode, hipStream_t stream);      // SQD_OK, or SQD_EINVAL (error text set) for a launch no kernel serves
int photo_tile_count(int B, int H, int W, int rows_per_task, int family);      // family: 1 fused forward, 0 identity / coefficients, 2 backward
int photo_fwd_waves(int B, int H, int W, int rows_per_task);
bool photo_sources_hwc_ok(int B, int S, int H, int W, int rows_per_task, int loss_flags);
void launch_photo_bwd_tile(const sqd_photo_bwd_args &a, hipStream_t stream);
}  // namespace sqd
