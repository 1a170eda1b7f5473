/*
 * h264_frame_rest.hip — the second launch of the inter reconstruction on tiled surfaces: the inter macroblocks h264_frame_tiled.hip's runs left
 * (two lists, weights, partitions, the 8x8 transform), through h264_recon_dev.h's code — 16x16 path and the per-4x4-block general path, no
 * partition loop.  A wave takes the macroblocks of FQ_REST (8) runs one after the other; the lane id is OPAQUE here (h264_dev.h): what the general
 * code derives from the lane number is recomputed in each of its phases instead of living in registers over the wave's loop.
 */
#include "h264_recon_fast.h"

namespace {
__attribute__((amdgpu_waves_per_eu(MI355_RECON_WAVES, MI355_RECON_WAVES)))
__global__ void __launch_bounds__(64)
k_recon_inter_rest(const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, int run, int runs_row, unsigned long long inv_runs, unsigned long long inv_h, int nruns, int ngroups, int per_xcd,
                   const uint32_t *__restrict__ rest)
{
    __shared__ MbLds s;
    recon_inter_rest(s, frames, max_w, max_h, run, runs_row, inv_runs, inv_h, nruns, ngroups, per_xcd, rest);
}
}  // namespace

namespace mi355 {
void recon_inter_rest_launch(const mi355_h264_frame *d_frames, int max_w, int max_h, int run, int runs_row, unsigned long long inv_runs, unsigned long long inv_h, int nruns,
                             const uint32_t *rest, hipStream_t stream)
{
    const int ngroups = (nruns + FQ_REST - 1) / FQ_REST, per_xcd = (ngroups + 7) / 8;
    hipLaunchKernelGGL(k_recon_inter_rest, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, stream, d_frames, max_w, max_h, run, runs_row, inv_runs, inv_h, nruns, ngroups, per_xcd, rest);
}
}  // namespace mi355
