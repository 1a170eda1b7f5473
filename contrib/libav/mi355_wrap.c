/*
 * mi355_wrap.c — the Tier-1 binding without a patch: PRODUCT glue that lives beside the reference's libavcodec.
 *
 * The reference fills its per-codec pointer tables in ff_*_init() and lets an architecture overwrite entries afterwards
 * (h264dsp.c:139-142: `if (ARCH_X86) ff_h264dsp_init_x86(c, bit_depth, chroma_format_idc);`).  An in-tree port adds one such line
 * per table.  Without touching the tree, link the decoder with
 *     -Wl,--wrap=ff_h264dsp_init,--wrap=ff_h264qpel_init,--wrap=ff_h264chroma_init,--wrap=ff_h264_pred_init,--wrap=ff_videodsp_init
 *     -Wl,--wrap=ff_hevc_dsp_init,--wrap=ff_hevc_pred_init
 * and this file: every table is filled by the reference's own init, then by the ff_*_init_mi355x hook of include/mi355dsp.h
 * (the reference's C function stays wherever the hook leaves an entry alone).  The H.264 and the HEVC half are independent
 * (compile with -DMI355_WRAP_NO_HEVC / -DMI355_WRAP_NO_H264 when only one decoder is linked).
 *
 * MI355_TIER1_PLAIN=1 leaves every table as the reference filled it (the comparison run); MI355_DEVICE=n names the GPU.
 * mi355_wrap_stats(): how many table initialisations were hooked and how many pointer-sized entries the hooks replaced — a
 * hook that silently filled nothing would otherwise go unnoticed.
 * Tested as shipped: oracle/_ref/h264_tier1_{emu,gpu}, hevc_tier1_{emu,gpu} link this file (tests/test_tier1_decoder*.py,
 * test_hevc_decoder.py, test_synth_streams*.py).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavcodec/avcodec.h"
#ifndef MI355_WRAP_NO_H264
#include "libavcodec/h264dsp.h"
#include "libavcodec/h264qpel.h"
#include "libavcodec/h264chroma.h"
#include "libavcodec/h264pred.h"
#endif
#include "libavcodec/videodsp.h"
#ifndef MI355_WRAP_NO_HEVC
#include "libavcodec/hevcdsp.h"
#include "libavcodec/hevcdec.h"
#endif
#include "mi355dsp.h"              /* the table structs are skipped: the reference's headers came first */

static unsigned long n_hooks, n_replaced;
static int state;                  /* 0 not looked yet, 1 hooks on, -1 plain (asked for, or no usable GPU) */

static int hooks_on(void)
{
    if (!state) {
        const char *dev = getenv("MI355_DEVICE");
        if (getenv("MI355_TIER1_PLAIN")) state = -1;
        else if (mi355_get_device() < 0 && mi355_init(dev ? atoi(dev) : 0) != 0) {
            fprintf(stderr, "mi355 wrap: no usable MI355X — the reference's C functions stay\n");
            state = -1;
        } else state = 1;
    }
    return state > 0;
}
/* MI355_TIER1_SKIP=h264dsp,qpel,chroma,pred,videodsp,hevcdsp,hevcpred: leave the named tables to the reference (finding which table a
 * difference comes from) */
static int skipped(const char *table)
{
    const char *e = getenv("MI355_TIER1_SKIP"), *p = e ? strstr(e, table) : NULL;
    const size_t n = strlen(table);
    while (p) {
        if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return 1;
        p = strstr(p + 1, table);
    }
    return 0;
}
static void count(const void *before, const void *after, size_t bytes)
{
    const void *const *a = before, *const *b = after;
    for (size_t i = 0; i < bytes / sizeof(void *); i++) n_replaced += a[i] != b[i];
    n_hooks++;
}
void mi355_wrap_stats(unsigned long *tables_hooked, unsigned long *entries_replaced)
{
    if (tables_hooked) *tables_hooked = n_hooks;
    if (entries_replaced) *entries_replaced = n_replaced;
}

#ifndef MI355_WRAP_NO_H264
void __real_ff_h264dsp_init(H264DSPContext *c, const int bit_depth, const int chroma_format_idc);
void __wrap_ff_h264dsp_init(H264DSPContext *c, const int bit_depth, const int chroma_format_idc)
{
    __real_ff_h264dsp_init(c, bit_depth, chroma_format_idc);
    const H264DSPContext was = *c;
    if (hooks_on() && !skipped("h264dsp")) ff_h264dsp_init_mi355x(c, bit_depth, chroma_format_idc);
    count(&was, c, sizeof(was));
}
void __real_ff_h264qpel_init(H264QpelContext *c, int bit_depth);
void __wrap_ff_h264qpel_init(H264QpelContext *c, int bit_depth)
{
    __real_ff_h264qpel_init(c, bit_depth);
    const H264QpelContext was = *c;
    if (hooks_on() && !skipped("qpel")) ff_h264qpel_init_mi355x(c, bit_depth);
    count(&was, c, sizeof(was));
}
void __real_ff_h264chroma_init(H264ChromaContext *c, int bit_depth);
void __wrap_ff_h264chroma_init(H264ChromaContext *c, int bit_depth)
{
    __real_ff_h264chroma_init(c, bit_depth);
    const H264ChromaContext was = *c;
    if (hooks_on() && !skipped("chroma")) ff_h264chroma_init_mi355x(c, bit_depth);
    count(&was, c, sizeof(was));
}
void __real_ff_h264_pred_init(H264PredContext *h, int codec_id, const int bit_depth, const int chroma_format_idc);
void __wrap_ff_h264_pred_init(H264PredContext *h, int codec_id, const int bit_depth, const int chroma_format_idc)
{
    __real_ff_h264_pred_init(h, codec_id, bit_depth, chroma_format_idc);
    const H264PredContext was = *h;
    if (hooks_on() && !skipped("pred")) ff_h264_pred_init_mi355x(h, codec_id, bit_depth, chroma_format_idc);
    count(&was, h, sizeof(was));
}
#endif

#ifndef MI355_WRAP_NO_VIDEODSP
void __real_ff_videodsp_init(VideoDSPContext *ctx, int bpc);
void __wrap_ff_videodsp_init(VideoDSPContext *ctx, int bpc)
{
    __real_ff_videodsp_init(ctx, bpc);
    const VideoDSPContext was = *ctx;
    if (hooks_on() && !skipped("videodsp")) ff_videodsp_init_mi355x(ctx, bpc);
    count(&was, ctx, sizeof(was));
}
#endif

#ifndef MI355_WRAP_NO_HEVC
void __real_ff_hevc_dsp_init(HEVCDSPContext *c, int bit_depth);
void __wrap_ff_hevc_dsp_init(HEVCDSPContext *c, int bit_depth)
{
    __real_ff_hevc_dsp_init(c, bit_depth);
    const HEVCDSPContext was = *c;
    if (hooks_on() && !skipped("hevcdsp")) ff_hevc_dsp_init_mi355x(c, bit_depth);
    count(&was, c, sizeof(was));
}
#ifndef MI355_WRAP_HEVC_NO_PRED
void __real_ff_hevc_pred_init(HEVCPredContext *c, int bit_depth);
void __wrap_ff_hevc_pred_init(HEVCPredContext *c, int bit_depth)
{
    __real_ff_hevc_pred_init(c, bit_depth);
    const HEVCPredContext was = *c;
    if (hooks_on() && !skipped("hevcpred")) ff_hevc_pred_init_mi355x(c, bit_depth);
    count(&was, c, sizeof(was));
}
#endif
#endif
