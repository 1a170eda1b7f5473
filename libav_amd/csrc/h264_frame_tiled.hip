/*
 * h264_frame_tiled.hip — the inter reconstruction kernel for batches whose surfaces are macroblock-tiled throughout
 * (mi355_h264_recon_inter_layouts_dev with MI355_LAYOUTS_TILED: what the bench, sessions and bridges of tiled sequences launch).
 * A wave walks a run of MI355_RECON_RUN consecutive macroblocks; the plain P macroblock (16x16, list 0, no weights, 4x4 transforms)
 * goes through h264_recon_fast.h (raw LDS-DMA windows, the 6-tap filters as v_mfma_i32_16x16x32_i8 products, the next record in flight
 * under the prediction), every other type through h264_recon_dev.h's code — 16x16 path and the per-4x4-block general path, no
 * partition loop.  Compiled with the lane id PLAIN: the compiler may keep what it derives from the lane number in registers over the run.
 */
#define MI355_PLAIN_LANE 1
#include "h264_recon_fast.h"

#ifndef MI355_RECON_RUN
#define MI355_RECON_RUN 8
#endif

namespace {
__attribute__((amdgpu_waves_per_eu(MI355_RECON_WAVES, MI355_RECON_WAVES)))
__global__ void __launch_bounds__(64)
k_recon_inter_tiled(const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, unsigned long long inv_w, unsigned long long inv_h, int nblocks, int per_xcd)
{
    __shared__ MbLds s;
    recon_inter_run<MI355_RECON_RUN>(s, frames, max_w, max_h, inv_w, inv_h, nblocks, per_xcd);
}
}  // namespace

namespace mi355 {
void recon_inter_tiled_launch(const mi355_h264_frame *d_frames, int max_w, int max_h, unsigned long long inv_w, unsigned long long inv_h, int nblocks, int /*per_xcd*/, hipStream_t stream)
{
    const int waves = (nblocks + MI355_RECON_RUN - 1) / MI355_RECON_RUN, per_xcd = (waves + 7) / 8;
    hipLaunchKernelGGL(k_recon_inter_tiled, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, stream, d_frames, max_w, max_h, inv_w, inv_h, nblocks, per_xcd);
}
}  // namespace mi355
