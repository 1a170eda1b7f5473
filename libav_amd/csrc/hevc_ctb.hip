/*
 * hevc_ctb.hip — mi355_hevc_recon_ctbs_dev (include/mi355_hevc_batch.h): the inter reconstruction of a coding tree block as ONE workgroup —
 * hls_coding_quadtree's prediction units (hevcdec.c:1695-1885: luma_mc / chroma_mc, put_unweighted_pred / weighted_pred) and transform units
 * (hls_transform_unit :1238-1260: idct*, add_residual) of the block, its samples kept in LDS from the first prediction to the last residual
 * and stored to the picture once, in whole lines.
 *
 * A workgroup = CTB_WAVES wavefronts; the block's prediction jobs are dealt to the waves, a workgroup barrier, its transform units likewise, a
 * barrier, the store.  A wave works on its job alone, on scratch of its own: the wave-level bodies of hevc_dev.h / hevc_batch_dev.h are
 * instantiated here with a WAVE-scope rendezvous between their LDS phases (MI355_HEVC_SYNC, a namespace of their own) — the waves of a workgroup
 * run different jobs of different lengths, a workgroup barrier inside a job would not be reached by all of them.
 * Hot shapes go through the matrix unit (hevc_ctb_fast.h); everything else through the bodies of the batch kernels, writing to the tile.
 */
#define MI355_HEVC_NS mi355_ctb
#define MI355_HEVC_SYNC() MI355_WAVE_SYNC()
#include <type_traits>
#include "mi355_rt.h"
#include "hevc_dev.h"
#include "../../include/mi355_hevc_batch.h"

using namespace mi355;
using namespace mi355_ctb;

namespace {

#include "hevc_batch_dev.h"
#include "hevc_ctb_fast.h"

/* waves per workgroup: the matrix-path kernel deals a block's jobs to more, shorter-lived waves (a 64x64 block of four 32x32 prediction units is 4 luma + 4 chroma jobs,
 * then 6 transform units: with 8 waves a round each); the general kernel keeps 4 (its per-wave scratch is larger) */
#ifndef MI355_CTB_WAVES
#define MI355_CTB_WAVES 8
#endif
constexpr int CTB_WAVES_FAST = MI355_CTB_WAVES, CTB_WAVES_GENERAL = 4;
/* the block's samples in LDS: rows 16 bytes longer than a full row, so that the rows a matrix product's sixteen lanes write (8 bytes each, one row per
 * lane) spread over the banks two by two instead of all sixteen meeting in two banks; still 16-byte aligned for the way out */
constexpr int CTB_PITCH_Y = 64 * 2 + 16, CTB_PITCH_C = 32 * 2 + 16;
struct __attribute__((aligned(16))) CtbTile {
    uint8_t y[64 * CTB_PITCH_Y];
    uint8_t c[2][32 * CTB_PITCH_C];
};
union __attribute__((aligned(16))) CtbScratch {
    HevcMcScratch mc;
    IdctScratch tu;
    CfWin win;
};

/* a / b for a < 2^22, b > 0 (an offset inside a block over its picture's stride), both wave-uniform */
__device__ __forceinline__ int ctb_div(unsigned a, unsigned b)
{
#ifdef MI355_HIP_EMU_H
    return (int)(a / b);
#else
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    if ((unsigned)q * b > a) q--;
    if ((unsigned)(q + 1) * b <= a) q++;
    return q;
#endif
}

/* the block's three planes as scalars (no array indexed at run time: that would live in scratch memory) */
struct CtbGeom {
    uint8_t *d0, *d1, *d2;
    int s0, s1, s2;
    int w, h, px;
    __device__ __forceinline__ uint8_t *dst(int pl) const { return pl == 0 ? d0 : (pl == 1 ? d1 : d2); }
    __device__ __forceinline__ int stride(int pl) const { return pl == 0 ? s0 : (pl == 1 ? s1 : s2); }
    /* which plane a pointer of a job lies in (first_plane or later), and where: the planes are separate ranges of rows */
    __device__ __forceinline__ bool locate(const uint8_t *p, int first_plane, int &pl, int &x, int &y) const
    {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const ptrdiff_t o = p - dst(q);
            const int rows = q ? h >> 1 : h, st = stride(q);
            if (q >= first_plane && o >= 0 && o < (ptrdiff_t)rows * st) {
                pl = q;
                y = ctb_div((unsigned)o, (unsigned)st);
                x = ((int)o - y * st) >> (px - 1);
                return true;
            }
        }
        return false;
    }
};
__device__ __forceinline__ uint8_t *tile_at(CtbTile &t, int pl, int x, int y, int px)
{
    return pl ? t.c[pl - 1] + y * CTB_PITCH_C + x * px : t.y + y * CTB_PITCH_Y + x * px;
}

/* the tile <-> the picture, V bytes per thread and access (rows of `rb` bytes, V divides rb, the plane's pointer and its stride) */
template <int V, bool LOAD>
__device__ __forceinline__ void tile_move(uint8_t *t, int tpitch, uint8_t *g, int gstride, int rb, int rows, int tid, int nthreads)
{
    const int per = rb / V, n = per * rows, inv = mi355_inv20(per);
    for (int i = tid; i < n; i += nthreads) {
        const int r = mi355_div20(i, inv), k = i - r * per;
        uint8_t *tp = t + r * tpitch + V * k, *gp = g + (ptrdiff_t)r * gstride + V * k;
        if (LOAD) __builtin_memcpy(tp, gp, V); else __builtin_memcpy(gp, tp, V);
    }
}
template <bool LOAD>
__device__ __forceinline__ void tile_all(CtbTile &t, const CtbGeom &G, int tid, int nthreads)
{
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        const int rb = (pl ? G.w >> 1 : G.w) * G.px, rows = pl ? G.h >> 1 : G.h, st = G.stride(pl);
        uint8_t *tp = pl ? t.c[pl - 1] : t.y, *g = mi355_global(G.dst(pl));
        const int tpitch = pl ? CTB_PITCH_C : CTB_PITCH_Y;
        const unsigned al = (unsigned)(uintptr_t)g | (unsigned)st | (unsigned)rb;
        if ((al & 15) == 0) tile_move<16, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else if ((al & 7) == 0) tile_move<8, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else if ((al & 3) == 0) tile_move<4, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else if ((al & 1) == 0) tile_move<2, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else tile_move<1, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
    }
}

/* what the matrix path takes (include/mi355_hevc_batch.h) */
template <bool WIDE> __device__ __forceinline__ bool mc_is_fast(const mi355_hevc_mcpred_job &j)
{
    return j.kind == MI355_HEVC_PRED_PUT && (j.width & 15) == 0 && (j.height & 15) == 0 && (j.src0_stride & (WIDE ? 15 : 7)) == 0;
}
__device__ __forceinline__ bool tu_is_fast(const mi355_hevc_tu_job &j)
{
    return j.dst && j.kind == MI355_HEVC_TU_IDCT && (j.log2_size == 4 || j.log2_size == 5) && ((uintptr_t)j.coeffs & 15) == 0;
}

/* one prediction job of the block -> the tile.  GENERAL: the bodies of the batch kernels are compiled in; without them a job the matrix path does not take
 * is reported (false) and nothing is done. */
template <bool WIDE, bool GENERAL, class Scratch>
__device__ __forceinline__ bool ctb_predict(CtbTile &tile, Scratch &s, const CtbGeom &G, mi355_hevc_mcpred_job j, int bd, int lane)
{
    int pl = 0, x = 0, y = 0, plb = 0, xb = 0, yb = 0;
    if (!G.locate(j.dst, j.chroma ? 1 : 0, pl, x, y)) return true;
    if (j.chroma == 2 && !G.locate(j.dst_b, 1, plb, xb, yb)) return true;
    const int pitch = pl ? CTB_PITCH_C : CTB_PITCH_Y, px = G.px;
    uint8_t *t0 = tile_at(tile, pl, x, y, px), *t1 = j.chroma == 2 ? tile_at(tile, plb, xb, yb, px) : nullptr;
    if (mc_is_fast<WIDE>(j)) {
        const int before = j.chroma ? 1 : 3, bx = j.mx0 ? before : 0, by = j.my0 ? before : 0;
        CfPass ph, pv;
        if (j.chroma) {
            ph = j.mx0 ? cf_pass(k_epel[j.mx0], 4, bd - 8) : cf_pass_one(1, 0);
            pv = j.my0 ? cf_pass(k_epel[j.my0], 4, j.mx0 ? 6 : bd - 8) : cf_pass_one(j.mx0 ? 1 : 1 << (14 - bd), 0);
        } else {
            ph = j.mx0 ? cf_pass(k_qpel[j.mx0], 8, bd - 8) : cf_pass_one(1, 0);
            pv = j.my0 ? cf_pass(k_qpel[j.my0], 8, j.mx0 ? 6 : bd - 8) : cf_pass_one(j.mx0 ? 1 : 1 << (14 - bd), 0);
        }
        for (int plane = 0; plane < (j.chroma == 2 ? 2 : 1); plane++) {
            const uint8_t *src = mi355_global(plane ? j.src0_b : j.src0);
            uint8_t *tp = plane ? t1 : t0;
            for (int ty = 0; ty < j.height; ty += 32)
            for (int tx = 0; tx < j.width; tx += 32) {
                const int tw = j.width - tx < 32 ? j.width - tx : 32, th = j.height - ty < 32 ? j.height - ty : 32;
                cf_mc_tile<WIDE>(s.win, src + (ptrdiff_t)(ty - by) * j.src0_stride + (ptrdiff_t)(tx - bx) * px, j.src0_stride, tw, th, ph, pv, bd,
                                 tp + ty * pitch + tx * px, pitch, lane);
            }
        }
        return true;
    }
    if constexpr (GENERAL) {
        /* the body of k_hevc_mcpred_batch with the tile as its picture */
        j.dst = t0; j.dst_b = t1; j.dst_stride = pitch;
        int16_t *const keep = s.mc.tmp + HEVC_MC_BI_ROWS * HEVC_MC_TPITCH;
        switch ((j.chroma ? 4 : 0) + (j.kind & 3)) {
        case 0: hevc_mcpred_taps<8, 0, false>(j, bd, s.mc, keep); break;   case 1: hevc_mcpred_taps<8, 1, false>(j, bd, s.mc, keep); break;
        case 2: hevc_mcpred_taps<8, 2, false>(j, bd, s.mc, keep); break;   case 3: hevc_mcpred_taps<8, 3, false>(j, bd, s.mc, keep); break;
        case 4: hevc_mcpred_taps<4, 0, false>(j, bd, s.mc, keep); break;   case 5: hevc_mcpred_taps<4, 1, false>(j, bd, s.mc, keep); break;
        case 6: hevc_mcpred_taps<4, 2, false>(j, bd, s.mc, keep); break;   default: hevc_mcpred_taps<4, 3, false>(j, bd, s.mc, keep); break;
        }
        MI355_WAVE_SYNC();
        return true;
    }
    return false;
}

/* one transform unit of the block -> added to the tile; `pre`: its coefficients were requested before (cf_idct_load) and wait in `raw` */
template <bool WIDE, bool GENERAL, class Scratch>
__device__ __forceinline__ bool ctb_residual(CtbTile &tile, Scratch &s, const CtbGeom &G, mi355_hevc_tu_job j, int bd, int lane, bool pre, CfRaw &raw)
{
    int pl = 0, x = 0, y = 0;
    if (!j.dst) return !GENERAL ? false : true;
    if (!G.locate(j.dst, 0, pl, x, y)) return true;
    const int pitch = pl ? CTB_PITCH_C : CTB_PITCH_Y;
    uint8_t *tp = tile_at(tile, pl, x, y, G.px);
    if (tu_is_fast(j)) {
        const uint8_t *c = reinterpret_cast<const uint8_t *>(mi355_global(j.coeffs));
        if (j.log2_size == 5) { if (!pre) cf_idct_load<5>(raw, c, j.col_limit, lane); cf_idct_run<5, WIDE>(raw, j.col_limit, bd, tp, pitch, lane); }
        else { if (!pre) cf_idct_load<4>(raw, c, j.col_limit, lane); cf_idct_run<4, WIDE>(raw, j.col_limit, bd, tp, pitch, lane); }
        MI355_WAVE_SYNC();
        return true;
    }
    if constexpr (GENERAL) {
        j.dst = tp; j.dst_stride = pitch;
        hevc_residual_run<false>(s.tu, j, lane < 32, lane >> 5, lane & 31, bd);
        MI355_WAVE_SYNC();
        return true;
    }
    return false;
}

struct FastScratch { CfWin win; };

/* GENERAL = false: the matrix-path kernel — it takes the blocks ALL of whose jobs are of the matrix path's shapes and leaves every other block untouched
 * (a wave that meets such a job says so in LDS; nothing is stored);
 * GENERAL = true: the kernel with every body compiled in.  `only_rest`: it follows the matrix-path kernel on the same list and takes exactly the blocks that one
 * left (the same test, made by the lanes on the job records before anything else). */
template <bool WIDE, bool GENERAL, int NW>
__global__ void __launch_bounds__(64 * NW) k_hevc_recon_ctbs(const mi355_hevc_ctb_job *ctbs, const mi355_hevc_mcpred_job *mc, const mi355_hevc_tu_job *tus, int bd,
                                                             int only_rest, uint32_t *error_word)
{
    typedef typename std::conditional<GENERAL, CtbScratch, FastScratch>::type Scratch;
    __shared__ CtbTile tile;
    __shared__ Scratch scratch[NW];
    __shared__ int s_flag;
    const int tid = (int)threadIdx.x, wave = uniform(tid >> 6), lane = lane_id();
    const mi355_hevc_ctb_job &cj = ctbs[blockIdx.x];
    CtbGeom G;
    G.d0 = cj.dst[0]; G.d1 = cj.dst[1]; G.d2 = cj.dst[2];
    G.s0 = uniform(cj.stride[0]); G.s1 = uniform(cj.stride[1]); G.s2 = uniform(cj.stride[2]);
    G.w = uniform(cj.width); G.h = uniform(cj.height); G.px = WIDE ? 2 : 1;
    const int n_mc = uniform((int)cj.n_mc), n_tu = uniform((int)cj.n_tu);
    const mi355_hevc_mcpred_job *my_mc = mc + uniform((int)cj.first_mc);
    const mi355_hevc_tu_job *my_tu = tus + uniform((int)cj.first_tu);
    if (tid == 0) s_flag = 0;
    if (GENERAL && only_rest) {
        /* is any job of the block outside the matrix path?  (a job record per lane) */
        __syncthreads();
        bool odd = false;
        for (int i = tid; i < n_mc + n_tu; i += 64 * NW)
            odd = odd || (i < n_mc ? !mc_is_fast<WIDE>(mi355_global_v(my_mc)[i]) : !tu_is_fast(mi355_global_v(my_tu)[i - n_mc]));
        if (odd) s_flag = 1;
        __syncthreads();
        if (!s_flag) return;
        __syncthreads();
        if (tid == 0) s_flag = 0;
    }
    /* the first transform unit of this wave: its coefficients are requested now and arrive while the wave predicts */
    CfRaw raw;
    bool pre = false;
    if (wave < n_tu) {
        const mi355_hevc_tu_job &j = my_tu[wave];
        if (tu_is_fast(j)) {
            const uint8_t *c = reinterpret_cast<const uint8_t *>(mi355_global(j.coeffs));
            if (j.log2_size == 5) cf_idct_load<5>(raw, c, j.col_limit, lane); else cf_idct_load<4>(raw, c, j.col_limit, lane);
            pre = true;
        }
    }
    if (uniform(cj.flags) & MI355_HEVC_CTB_PARTIAL) tile_all<true>(tile, G, tid, 64 * NW);
    __syncthreads();
    Scratch &s = scratch[wave];
    bool ok = true;
    for (int i = wave; i < n_mc; i += NW) ok = ctb_predict<WIDE, GENERAL>(tile, s, G, my_mc[i], bd, lane) && ok;
    if (!GENERAL) {
        /* a unit the matrix path does not take: known before any residual is added */
        for (int i = wave; i < n_tu; i += NW) ok = ok && tu_is_fast(my_tu[i]);
        if (!ok) s_flag = 1;
    }
    __syncthreads();
    if (!GENERAL && s_flag) {
        /* left to the general kernel; a caller that promised there are no such blocks (MI355_HEVC_RECON_UNIFORM) finds out */
        if (error_word && tid == 0) atomicOr(error_word, (uint32_t)MI355_ERR_CTB_NOT_UNIFORM);
        return;
    }
    for (int i = wave; i < n_tu; i += NW) { ctb_residual<WIDE, GENERAL>(tile, s, G, my_tu[i], bd, lane, pre && i == wave, raw); }
    __syncthreads();
    tile_all<false>(tile, G, tid, 64 * NW);
}

}  // namespace

extern "C" int mi355_hevc_recon_ctbs_dev(const mi355_hevc_ctb_job *d_ctbs, int n_ctbs, const mi355_hevc_mcpred_job *d_mc, const mi355_hevc_tu_job *d_tus,
                                         int bit_depth, unsigned flags, void *stream)
{
    if (!bind()) { std::fprintf(stderr, "mi355dsp: HEVC batch entry point without mi355_init(); no CPU fallback\n"); std::abort(); }
    if (!d_ctbs || n_ctbs <= 0 || !(bit_depth == 8 || bit_depth == 9 || bit_depth == 10)) return -1;
    const hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n_ctbs);
    static const bool general_only = std::getenv("MI355_CTB_GENERAL_ONLY") != nullptr;      /* developer switch: every block through the general kernel */
    const bool uniform_promised = (flags & MI355_HEVC_RECON_UNIFORM) != 0;
    uint32_t *err = uniform_promised ? mi355::error_word() : nullptr;
    if (uniform_promised && !err) return -4;
    static const int waves = std::getenv("MI355_CTB_WAVES") ? std::atoi(std::getenv("MI355_CTB_WAVES")) : CTB_WAVES_FAST;      /* developer switch: 4 */
    if (!general_only) {
        if (waves == 4) {
            if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_recon_ctbs<true, false, 4>), grid, dim3(256), 0, st, d_ctbs, d_mc, d_tus, bit_depth, 0, err);
            else hipLaunchKernelGGL((k_hevc_recon_ctbs<false, false, 4>), grid, dim3(256), 0, st, d_ctbs, d_mc, d_tus, bit_depth, 0, err);
        } else {
            if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_recon_ctbs<true, false, CTB_WAVES_FAST>), grid, dim3(64 * CTB_WAVES_FAST), 0, st, d_ctbs, d_mc, d_tus, bit_depth, 0, err);
            else hipLaunchKernelGGL((k_hevc_recon_ctbs<false, false, CTB_WAVES_FAST>), grid, dim3(64 * CTB_WAVES_FAST), 0, st, d_ctbs, d_mc, d_tus, bit_depth, 0, err);
        }
    }
    if (general_only || !uniform_promised) {
        const int only_rest = general_only ? 0 : 1;
        if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_recon_ctbs<true, true, CTB_WAVES_GENERAL>), grid, dim3(64 * CTB_WAVES_GENERAL), 0, st, d_ctbs, d_mc, d_tus, bit_depth, only_rest, (uint32_t *)nullptr);
        else hipLaunchKernelGGL((k_hevc_recon_ctbs<false, true, CTB_WAVES_GENERAL>), grid, dim3(64 * CTB_WAVES_GENERAL), 0, st, d_ctbs, d_mc, d_tus, bit_depth, only_rest, (uint32_t *)nullptr);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
