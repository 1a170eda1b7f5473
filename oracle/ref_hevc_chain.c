/* TEST INFRASTRUCTURE (oracle/): BASELINE config 3 — the synthetic HEVC 10-bit 2160p chain of tools/hevc_chain.py — decoded
 * by the REFERENCE's own functions, compiled where they lie into _ref/libhevcfilterref.so: what the reference's decoder
 * would call for such a picture, in its order.
 *   prediction   luma_mc / chroma_mc restated (hevcdec.c:1528-1640): emulated_edge_mc for windows that reach over a picture
 *                border, put_hevc_qpel / put_hevc_epel into the 14-bit intermediate, put_unweighted_pred (:1837-1870)
 *   residual     idct[3](coeffs, col_limit) + add_residual[3] per 32x32 transform unit (hls_transform_unit, :1238-1260);
 *                the coefficients are copied first, as the residual decoder writes them for every unit
 *   in-loop      ff_hevc_hls_filters() per CTB in raster order (hevc_filter.c:735-746: deblocking_filter_CTB + sao_filter_CTB
 *                with the decoder's one-CTB lag) on a zeroed HEVCContext that holds exactly the fields those functions read
 * Used (1) as the checker of the measured chain: tests compare the device pictures with this driver's (tests/test_hevc_chain_*),
 * (2) as bench.py's cpu_baseline of the config-3 point (kind "reference": N pinned threads, one picture each). */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "libavutil/mem.h"
#include "libavutil/frame.h"
#include "libavcodec/hevcdec.h"
#include "libavcodec/videodsp.h"

typedef struct ref_hevc_chain {
    int32_t W, H, bd, reserved;
    uint8_t *ref[3], *cur[3], *out[3];       /* reference picture, reconstruction (deblocked in place), SAO output */
    int32_t stride[3], reserved1;            /* bytes; the same for the three surfaces */
    const int32_t *mv;                       /* [H / 32][W / 32][2] quarter samples */
    const int16_t *coef;                     /* [n_tu][1024]: luma units (raster, 32x32), then Cb, then Cr units (raster, one per 64x64) */
    const uint8_t *col_limit;                /* [n_tu] */
    const uint8_t *vertical_bs, *horizontal_bs;
    const int8_t *qp_y_tab;
    const uint8_t *sao_type;                 /* [3][ctb_h][ctb_w]: 0 off, 1 band, 2 edge */
    const int32_t *sao_offset;               /* [3][ctb_h][ctb_w][4] */
    const uint8_t *sao_eo, *sao_band;        /* [3][ctb_h][ctb_w] */
} ref_hevc_chain;

int ref_hevc_chain_run(const ref_hevc_chain *c)
{
    const int W = c->W, H = c->H, bd = c->bd, ps = bd > 8, px = 1 << ps;
    HEVCDSPContext dsp;
    VideoDSPContext vdsp;
    ff_hevc_dsp_init(&dsp, bd);
    ff_videodsp_init(&vdsp, bd);
    int16_t *tmp = av_malloc(2 * 64 * 64), *mcbuf = av_malloc(2 * (64 + 24) * 64), *cf = av_malloc(2 * 1024);
    uint8_t *emu = av_malloc((64 + 8) * (EDGE_EMU_BUFFER_STRIDE << 1));
    if (!tmp || !mcbuf || !cf || !emu) return -1;
    const ptrdiff_t es = EDGE_EMU_BUFFER_STRIDE << ps;
    const int nbx = W / 32, nby = H / 32;
    /* ---- prediction ---- */
    for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++) {
            const int mvx = c->mv[(by * nbx + bx) * 2], mvy = c->mv[(by * nbx + bx) * 2 + 1];
            {
                const int mx = mvx & 3, my = mvy & 3, el = ff_hevc_qpel_extra_before[mx], et = ff_hevc_qpel_extra_before[my];
                const int x = bx * 32 + (mvx >> 2), y = by * 32 + (mvy >> 2);
                ptrdiff_t ss = c->stride[0];
                const uint8_t *src = c->ref[0] + (ptrdiff_t)y * ss + x * px;
                if (x < el || y < et || x >= W - 32 - ff_hevc_qpel_extra_after[mx] || y >= H - 32 - ff_hevc_qpel_extra_after[my]) {
                    vdsp.emulated_edge_mc(emu, src - (et * ss + (el << ps)), es, ss, 32 + ff_hevc_qpel_extra[mx], 32 + ff_hevc_qpel_extra[my], x - el, y - et, W, H);
                    src = emu + et * es + (el << ps);
                    ss = es;
                }
                dsp.put_hevc_qpel[!!my][!!mx][5](tmp, 64, (uint8_t *)src, ss, 32, mx, my, mcbuf);
                dsp.put_unweighted_pred[5](c->cur[0] + (ptrdiff_t)by * 32 * c->stride[0] + bx * 32 * px, c->stride[0], tmp, 64, 32);
            }
            for (int pl = 1; pl < 3; pl++) {
                const int mx = mvx & 7, my = mvy & 7, pw = W >> 1, ph = H >> 1;
                const int x = bx * 16 + (mvx >> 3), y = by * 16 + (mvy >> 3);
                ptrdiff_t ss = c->stride[pl];
                const uint8_t *src = c->ref[pl] + (ptrdiff_t)y * ss + x * px;
                if (x < EPEL_EXTRA_BEFORE || y < EPEL_EXTRA_AFTER || x >= pw - 16 - EPEL_EXTRA_AFTER || y >= ph - 16 - EPEL_EXTRA_AFTER) {
                    vdsp.emulated_edge_mc(emu, src - EPEL_EXTRA_BEFORE * (ss + px), es, ss, 16 + EPEL_EXTRA, 16 + EPEL_EXTRA, x - EPEL_EXTRA_BEFORE, y - EPEL_EXTRA_BEFORE, pw, ph);
                    src = emu + EPEL_EXTRA_BEFORE * (es + px);
                    ss = es;
                }
                dsp.put_hevc_epel[!!my][!!mx][5](tmp, 32, (uint8_t *)src, ss, 16, mx, my, mcbuf);
                dsp.put_unweighted_pred_chroma[5](c->cur[pl] + (ptrdiff_t)by * 16 * c->stride[pl] + bx * 16 * px, c->stride[pl], tmp, 32, 16);
            }
        }
    /* ---- residual ---- */
    const int n32 = nbx * nby, ncx = W / 64, ncy = H / 64, n64 = ncx * ncy;
    for (int k = 0; k < n32 + 2 * n64; k++) {
        uint8_t *dst;
        ptrdiff_t st;
        if (k < n32) { st = c->stride[0]; dst = c->cur[0] + (ptrdiff_t)(k / nbx) * 32 * st + (k % nbx) * 32 * px; }
        else {
            const int pl = 1 + (k - n32) / n64, t = (k - n32) % n64;
            st = c->stride[pl]; dst = c->cur[pl] + (ptrdiff_t)(t / ncx) * 32 * st + (t % ncx) * 32 * px;
        }
        memcpy(cf, c->coef + (size_t)k * 1024, 2048);
        dsp.idct[3](cf, c->col_limit[k]);
        dsp.add_residual[3](dst, cf, st);
    }
    av_free(tmp); av_free(mcbuf); av_free(cf); av_free(emu);
    /* ---- deblocking + SAO: the decoder's own per-CTB calls ---- */
    HEVCContext *s = av_mallocz(sizeof(*s));
    HEVCSPS *sps = av_mallocz(sizeof(*sps));
    HEVCPPS *pps = av_mallocz(sizeof(*pps));
    AVFrame *fr = av_frame_alloc(), *sf = av_frame_alloc();
    HEVCFrame *hf = av_mallocz(sizeof(*hf));
    const int cw = (W + 63) >> 6, chh = (H + 63) >> 6, nctb = cw * chh;
    int *ident = av_malloc(sizeof(int) * nctb), *zeros = av_mallocz(sizeof(int) * nctb);
    SAOParams *sao = av_mallocz(sizeof(*sao) * nctb);
    DBParams *db = av_mallocz(sizeof(*db) * nctb);
    uint8_t *fse = av_malloc(nctb), *pcm = av_mallocz((size_t)(W >> 2) * (H >> 2));
    if (!s || !sps || !pps || !fr || !sf || !hf || !ident || !zeros || !sao || !db || !fse || !pcm) return -1;
    for (int i = 0; i < nctb; i++) { ident[i] = i; fse[i] = 1; }
    for (int i = 0; i < nctb; i++)
        for (int k = 0; k < 3; k++) {
            const int t = c->sao_type[k * nctb + i];
            sao[i].type_idx[k] = t == 2 ? SAO_EDGE : t == 1 ? SAO_BAND : SAO_NOT_APPLIED;
            sao[i].eo_class[k] = c->sao_eo[k * nctb + i];
            sao[i].band_position[k] = c->sao_band[k * nctb + i];
            for (int e = 0; e < 4; e++) sao[i].offset_val[k][e + 1] = c->sao_offset[(k * nctb + i) * 4 + e];
        }
    sps->log2_ctb_size = 6; sps->ctb_width = cw; sps->ctb_height = chh; sps->width = W; sps->height = H; sps->pixel_shift = ps;
    sps->log2_min_cb_size = 3; sps->min_cb_width = W >> 3; sps->log2_min_pu_size = 2; sps->min_pu_width = W >> 2; sps->min_pu_height = H >> 2;
    sps->log2_min_tb_size = 2;
    sps->hshift[1] = sps->hshift[2] = sps->vshift[1] = sps->vshift[2] = 1;
    sps->sao_enabled = 1;
    sps->bit_depth = bd;
    pps->ctb_addr_rs_to_ts = ident; pps->tile_id = zeros;
    s->ps.sps = sps; s->ps.pps = pps;
    s->deblock = db; s->sao = sao;
    s->vertical_bs = (uint8_t *)c->vertical_bs; s->horizontal_bs = (uint8_t *)c->horizontal_bs;
    s->bs_width = W >> 3; s->bs_height = H >> 3;
    s->qp_y_tab = (int8_t *)c->qp_y_tab;
    s->is_pcm = pcm;
    s->tab_slice_address = zeros; s->filter_slice_edges = fse;
    for (int i = 0; i < 3; i++) { fr->data[i] = c->cur[i]; fr->linesize[i] = c->stride[i]; sf->data[i] = c->out[i]; sf->linesize[i] = c->stride[i]; }
    s->frame = fr; s->sao_frame = sf;
    hf->frame = fr; s->ref = hf;
    ff_hevc_dsp_init(&s->hevcdsp, bd);
    for (int y = 0; y < H; y += 64)
        for (int x = 0; x < W; x += 64)
            ff_hevc_hls_filters(s, x, y, 64);
    /* the decoder's last call: the bottom-right CTB (hls_decode_entry, hevcdec.c:2420-2421) */
    ff_hevc_hls_filter(s, (cw - 1) * 64, (chh - 1) * 64);
    for (int i = 0; i < 3; i++) fr->data[i] = sf->data[i] = NULL;
    av_frame_free(&fr); av_frame_free(&sf);
    av_free(hf); av_free(ident); av_free(zeros); av_free(sao); av_free(db); av_free(fse); av_free(pcm); av_free(pps); av_free(sps); av_free(s);
    return 0;
}

/* ---- N pinned threads, each decoding its own picture repeatedly for ~`seconds`; returns the pictures decoded ------------ */
typedef struct Worker { const ref_hevc_chain *c; int cpu; double seconds; long done; } Worker;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void *worker(void *vp)
{
    Worker *w = vp;
    if (w->cpu >= 0) { cpu_set_t set; CPU_ZERO(&set); CPU_SET(w->cpu, &set); pthread_setaffinity_np(pthread_self(), sizeof(set), &set); }
    const double end = now_s() + w->seconds;
    do { if (ref_hevc_chain_run(w->c) != 0) break; w->done++; } while (now_s() < end);
    return NULL;
}
long ref_hevc_chain_bench_threads(const ref_hevc_chain *chains, int nthreads, const int *cpus, double seconds, double *wall)
{
    pthread_t *th = calloc((size_t)nthreads, sizeof(*th));
    Worker *w = calloc((size_t)nthreads, sizeof(*w));
    if (!th || !w) return -1;
    const double t0 = now_s();
    for (int t = 0; t < nthreads; t++) { w[t].c = &chains[t]; w[t].cpu = cpus ? cpus[t] : -1; w[t].seconds = seconds; pthread_create(&th[t], NULL, worker, &w[t]); }
    long total = 0;
    for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); total += w[t].done; }
    if (wall) *wall = now_s() - t0;
    free(th); free(w);
    return total;
}
