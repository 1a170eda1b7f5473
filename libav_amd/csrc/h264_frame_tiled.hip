/*
 * h264_frame_tiled.hip — the inter reconstruction kernel for batches whose surfaces are macroblock-tiled throughout
 * (mi355_h264_recon_inter_layouts_dev with MI355_LAYOUTS_TILED: what the bench, sessions and bridges of tiled sequences launch).
 * The same device code as h264_frame.hip's kernels (h264_recon_dev.h), instantiated for the tiled form alone — 16x16 fast path and
 * the per-4x4-block general path, no partition loop — and compiled with the lane id PLAIN: this instance needs 39 vector registers
 * with the opaque lane id (h264_dev.h), so the compiler may keep the address arithmetic it derives from the lane number in registers
 * (59 of the 64 that eight waves per SIMD allow) instead of recomputing it in every phase.
 */
#define MI355_PLAIN_LANE 1
#include "h264_recon_dev.h"

namespace {
__attribute__((amdgpu_waves_per_eu(MI355_RECON_WAVES, MI355_RECON_WAVES)))
__global__ void __launch_bounds__(64)
k_recon_inter_tiled(const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, unsigned long long inv_w, unsigned long long inv_h, int nblocks, int per_xcd)
{
    __shared__ MbLds s;
    recon_inter_wave<false, MI355_LAYOUTS_TILED>(s, frames, max_w, max_h, inv_w, inv_h, nblocks, per_xcd);
}
}  // namespace

namespace mi355 {
void recon_inter_tiled_launch(const mi355_h264_frame *d_frames, int max_w, int max_h, unsigned long long inv_w, unsigned long long inv_h, int nblocks, int per_xcd, hipStream_t stream)
{
    hipLaunchKernelGGL(k_recon_inter_tiled, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, stream, d_frames, max_w, max_h, inv_w, inv_h, nblocks, per_xcd);
}
}  // namespace mi355
