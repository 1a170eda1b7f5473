/*
 * h264_frame_tiled.hip — the inter reconstruction kernel for batches whose surfaces are macroblock-tiled throughout
 * (mi355_h264_recon_inter_layouts_dev with MI355_LAYOUTS_TILED: what the bench, sessions and bridges of tiled sequences launch).
 * Two launches.  k_recon_inter_tiled: a wave walks a run of consecutive macroblocks of one row; the plain P macroblock (16x16, list 0, no
 * weights, 4x4 transforms) goes through h264_recon_fast.h (raw LDS-DMA windows, the 6-tap filters as v_mfma_i32_16x16x32_i8 products, the
 * next macroblock's windows and coefficients in flight under the prediction), every other inter macroblock is noted in the run's word of
 * a scratch buffer.  k_recon_inter_rest (h264_frame_rest.hip): a wave per eight runs takes the noted macroblocks through
 * h264_recon_dev.h's code.  This file is compiled with the lane id PLAIN: the compiler may keep what it derives from the lane number in
 * registers over the run.
 */
#define MI355_PLAIN_LANE 1
#include "h264_recon_fast.h"

#include <cstdlib>

namespace {
__attribute__((amdgpu_waves_per_eu(MI355_RECON_WAVES, MI355_RECON_WAVES)))
__global__ void __launch_bounds__(64)
k_recon_inter_tiled(const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, int run, int runs_row, unsigned long long inv_runs, unsigned long long inv_h, int nwaves, int per_xcd,
                    uint32_t *__restrict__ rest)
{
    __shared__ MbLds s;
    recon_inter_run(s, frames, max_w, max_h, run, runs_row, inv_runs, inv_h, nwaves, per_xcd, rest);
}
}  // namespace

namespace mi355 {
/* nframes pictures of a max_w x max_h grid.  The run length: long runs spread the per-wave set-up (lane constants, the run's description) over more
 * macroblocks, short ones keep a small batch's waves many enough to fill the device (256 CUs x 32 waves) */
bool recon_inter_tiled_launch(const mi355_h264_frame *d_frames, int nframes, int max_w, int max_h, hipStream_t stream)
{
    static const int forced = std::getenv("MI355_RECON_RUN") ? std::atoi(std::getenv("MI355_RECON_RUN")) : 0;
    /* some forty rounds of the device's 8192 wave slots keep the last round's idle slots a few per cent of the launch; beyond that, longer runs */
    const long long mbs = (long long)nframes * max_w * max_h;
    int run = forced > 0 ? forced : (int)(mbs / (40ll * 8192));
    run = run < 4 ? 4 : (run > 15 ? 15 : run);
    if (run > max_w) run = max_w;
    const int runs_row = (max_w + run - 1) / run;
    run = (max_w + runs_row - 1) / runs_row;                      /* a row's runs of equal length */
    const unsigned long long one = 1ull << 40;
    const long long total = (long long)nframes * max_h * runs_row;
    const int nwaves = (int)total, per_xcd = (nwaves + 7) / 8;
    /* a word per run: written by every run, read by the launch behind it (the stream's scratch words: the loop filter zeroes what it uses of them on the same stream) */
    uint32_t *rest = sync_words(stream, (size_t)nwaves);
    if (!rest) return false;
    const unsigned long long inv_runs = (one + runs_row - 1) / runs_row, inv_h = (one + max_h - 1) / max_h;
    hipLaunchKernelGGL(k_recon_inter_tiled, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, stream, d_frames, max_w, max_h, run, runs_row, inv_runs, inv_h, nwaves, per_xcd, rest);
    recon_inter_rest_launch(d_frames, max_w, max_h, run, runs_row, inv_runs, inv_h, nwaves, rest, stream);
    return true;
}
}  // namespace mi355
