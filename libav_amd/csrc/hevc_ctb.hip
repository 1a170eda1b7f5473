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
#include "mi355_rt.h"
#include "hevc_dev.h"
#include "../../include/mi355_hevc_batch.h"

using namespace mi355;
using namespace mi355_ctb;

namespace {

#include "hevc_batch_dev.h"
#include "hevc_ctb_fast.h"

#ifndef MI355_CTB_WAVES
#define MI355_CTB_WAVES 4
#endif
constexpr int CTB_WAVES = MI355_CTB_WAVES;
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

struct CtbGeom {
    uint8_t *dst[3];
    int stride[3];
    int w, h, px;
    /* which plane a pointer of a job lies in, and where: the planes are separate ranges of rows */
    __device__ __forceinline__ bool locate(const uint8_t *p, int first_plane, int &pl, int &x, int &y) const
    {
        for (pl = first_plane; pl < 3; pl++) {
            const ptrdiff_t o = p - dst[pl];
            const int rows = pl ? h >> 1 : h;
            if (o >= 0 && o < (ptrdiff_t)rows * stride[pl]) {
                y = ctb_div((unsigned)o, (unsigned)stride[pl]);
                x = ((int)o - y * stride[pl]) / px;
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
    for (int pl = 0; pl < 3; pl++) {
        const int rb = (pl ? G.w >> 1 : G.w) * G.px, rows = pl ? G.h >> 1 : G.h;
        uint8_t *tp = pl ? t.c[pl - 1] : t.y, *g = mi355_global(G.dst[pl]);
        const int tpitch = pl ? CTB_PITCH_C : CTB_PITCH_Y;
        const unsigned al = (unsigned)(uintptr_t)G.dst[pl] | (unsigned)G.stride[pl] | (unsigned)rb;
        if ((al & 15) == 0) tile_move<16, LOAD>(tp, tpitch, g, G.stride[pl], rb, rows, tid, nthreads);
        else if ((al & 7) == 0) tile_move<8, LOAD>(tp, tpitch, g, G.stride[pl], rb, rows, tid, nthreads);
        else if ((al & 3) == 0) tile_move<4, LOAD>(tp, tpitch, g, G.stride[pl], rb, rows, tid, nthreads);
        else if ((al & 1) == 0) tile_move<2, LOAD>(tp, tpitch, g, G.stride[pl], rb, rows, tid, nthreads);
        else tile_move<1, LOAD>(tp, tpitch, g, G.stride[pl], rb, rows, tid, nthreads);
    }
}

/* one prediction job of the block -> the tile */
template <bool WIDE>
__device__ __forceinline__ void ctb_predict(CtbTile &tile, CtbScratch &s, const CtbGeom &G, mi355_hevc_mcpred_job j, int bd, int lane)
{
    int pl, x, y, plb = 0, xb = 0, yb = 0;
    if (!G.locate(j.dst, j.chroma ? 1 : 0, pl, x, y)) return;
    if (j.chroma == 2 && !G.locate(j.dst_b, 1, plb, xb, yb)) return;
    const int pitch = pl ? CTB_PITCH_C : CTB_PITCH_Y, px = G.px;
    uint8_t *t0 = tile_at(tile, pl, x, y, px), *t1 = j.chroma == 2 ? tile_at(tile, plb, xb, yb, px) : nullptr;
    constexpr int PB = WIDE ? 16 : 8;
    const bool fast = j.kind == MI355_HEVC_PRED_PUT && (j.width & 15) == 0 && (j.height & 15) == 0 && (j.src0_stride & (PB - 1)) == 0;
    if (fast) {
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
        return;
    }
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
}

/* one transform unit of the block -> added to the tile */
template <bool WIDE>
__device__ __forceinline__ void ctb_residual(CtbTile &tile, CtbScratch &s, const CtbGeom &G, mi355_hevc_tu_job j, int bd, int lane)
{
    int pl, x, y;
    if (!j.dst || !G.locate(j.dst, 0, pl, x, y)) return;
    const int pitch = pl ? CTB_PITCH_C : CTB_PITCH_Y;
    uint8_t *tp = tile_at(tile, pl, x, y, G.px);
    if (j.kind == MI355_HEVC_TU_IDCT && (j.log2_size == 4 || j.log2_size == 5) && ((uintptr_t)j.coeffs & 15) == 0) {
        const uint8_t *c = reinterpret_cast<const uint8_t *>(mi355_global(j.coeffs));
        CfRaw raw;
        if (j.log2_size == 5) { cf_idct_load<5>(raw, c, j.col_limit, lane); cf_idct_run<5, WIDE>(raw, j.col_limit, bd, tp, pitch, lane); }
        else { cf_idct_load<4>(raw, c, j.col_limit, lane); cf_idct_run<4, WIDE>(raw, j.col_limit, bd, tp, pitch, lane); }
        MI355_WAVE_SYNC();
        return;
    }
    j.dst = tp; j.dst_stride = pitch;
    hevc_residual_run<false>(s.tu, j, lane < 32, lane >> 5, lane & 31, bd);
    MI355_WAVE_SYNC();
}

template <bool WIDE>
__global__ void __launch_bounds__(64 * CTB_WAVES) k_hevc_recon_ctbs(const mi355_hevc_ctb_job *ctbs, const mi355_hevc_mcpred_job *mc, const mi355_hevc_tu_job *tus, int bd)
{
    __shared__ CtbTile tile;
    __shared__ CtbScratch scratch[CTB_WAVES];
    const int tid = (int)threadIdx.x, wave = uniform(tid >> 6), lane = lane_id();
    const mi355_hevc_ctb_job &cj = ctbs[blockIdx.x];
    CtbGeom G;
    for (int p = 0; p < 3; p++) { G.dst[p] = cj.dst[p]; G.stride[p] = uniform(cj.stride[p]); }
    G.w = uniform(cj.width); G.h = uniform(cj.height); G.px = WIDE ? 2 : 1;
    const int n_mc = uniform((int)cj.n_mc), n_tu = uniform((int)cj.n_tu);
    const mi355_hevc_mcpred_job *my_mc = mc + uniform((int)cj.first_mc);
    const mi355_hevc_tu_job *my_tu = tus + uniform((int)cj.first_tu);
    if (uniform(cj.flags) & MI355_HEVC_CTB_PARTIAL) {
        tile_all<true>(tile, G, tid, 64 * CTB_WAVES);
        __syncthreads();
    }
    CtbScratch &s = scratch[wave];
    for (int i = wave; i < n_mc; i += CTB_WAVES) ctb_predict<WIDE>(tile, s, G, my_mc[i], bd, lane);
    __syncthreads();
    for (int i = wave; i < n_tu; i += CTB_WAVES) ctb_residual<WIDE>(tile, s, G, my_tu[i], bd, lane);
    __syncthreads();
    tile_all<false>(tile, G, tid, 64 * CTB_WAVES);
}

}  // namespace

extern "C" int mi355_hevc_recon_ctbs_dev(const mi355_hevc_ctb_job *d_ctbs, int n_ctbs, const mi355_hevc_mcpred_job *d_mc, const mi355_hevc_tu_job *d_tus,
                                         int bit_depth, void *stream)
{
    if (!bind()) { std::fprintf(stderr, "mi355dsp: HEVC batch entry point without mi355_init(); no CPU fallback\n"); std::abort(); }
    if (!d_ctbs || n_ctbs <= 0 || !(bit_depth == 8 || bit_depth == 9 || bit_depth == 10)) return -1;
    if (bit_depth > 8) hipLaunchKernelGGL(k_hevc_recon_ctbs<true>, dim3((unsigned)n_ctbs), dim3(64 * CTB_WAVES), 0, (hipStream_t)stream, d_ctbs, d_mc, d_tus, bit_depth);
    else hipLaunchKernelGGL(k_hevc_recon_ctbs<false>, dim3((unsigned)n_ctbs), dim3(64 * CTB_WAVES), 0, (hipStream_t)stream, d_ctbs, d_mc, d_tus, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
