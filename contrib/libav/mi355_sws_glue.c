/*
 * mi355_sws_glue.c — libswscale binding: PRODUCT glue that lives beside the reference's libswscale.
 *
 * Compiled against the reference's own (private) swscale_internal.h; nothing in the reference tree changes.  Link with
 *   -Wl,--wrap=ff_getSwsFunc,--wrap=ff_yuv2rgb_get_func_ptr
 * (the two selectors of SwsContext.swscale, swscale_internal.h:263: ff_getSwsFunc, swscale.c:773, is the last step of
 * sws_init_context, utils.c:1336; ff_yuv2rgb_get_func_ptr, yuv2rgb.c:570, is what ff_get_unscaled_swscale,
 * swscale_unscaled.c:1054, takes for unscaled yuv -> rgb).
 *
 *   whole pictures (default)  both selectors return mi355_swsfunc for yuv420p -> rgb24 contexts: a call that hands over the
 *                          whole picture (srcSliceY == 0, srcSliceH == srcH — what sws_scale() callers such as vf_scale
 *                          without slices do) is ONE device pass (mi355_sws_scale: horizontal scaling, vertical scaling and
 *                          the table look-ups fused, no int16 lines in memory); a sliced call goes to the function the
 *                          reference had chosen.  The device context of a SwsContext is kept in a side table keyed by the
 *                          context's address (no field is added to the private struct) and built at the first call from the
 *                          filter banks and tables the reference built; it is rebuilt when the tables have changed
 *                          (sws_setColorspaceDetails).  Contexts outside the path (other formats, SWS_FULL_CHR_H_INT,
 *                          filters longer than the device tiles hold) keep the reference's function.
 *   inner loops (MI355_SWS_LINES=1)  the arch-hook form (cf. ff_sws_init_swscale_x86, swscale.c:773-790): after the
 *                          reference has chosen its C inner loops, c->hyScale / c->hcScale (hScale8To15_c, swscale.c:133)
 *                          and c->yuv2packed1 / 2 / X (yuv2rgb24_{1,2,X}_c, output.c:937-1110) are pointed at shims that
 *                          forward each call to the device (mi355_sws_hscale8to15, mi355_sws_yuv2rgb24_*).  swscale()'s
 *                          line-pull loop, ring buffers and filter banks stay the reference's.  ff_sws_init_mi355x().
 *   mi355_sws_describe()   fills the plain descriptor of include/mi355_sws.h from a live SwsContext (filter banks and
 *                          yuv->rgb tables exactly as the reference built them): what mi355_sws_create() takes for the
 *                          whole-picture and batched entry points.
 * The device is MI355_DEVICE (default 0), as for the decoder bridges.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"
#include "mi355_sws.h"
#include "mi355dsp.h"

/* -DMI355_SWS_DESCRIBE_ONLY: only mi355_sws_describe() (a build that wants the descriptor of a context without binding the
 * library: the plain reference library the tests compare against) */
#ifndef MI355_SWS_DESCRIBE_ONLY
static int mi355_swsfunc(SwsContext *c, const uint8_t *src[], int srcStride[], int srcSliceY, int srcSliceH, uint8_t *dst[], int dstStride[]);
#define mi355_swsfunc_entry mi355_swsfunc
static unsigned long n_calls;
unsigned long mi355_sws_glue_calls(void) { return n_calls; }      /* inner-loop calls forwarded so far (diagnostics) */

#endif

static void luts_of(const SwsContext *c, mi355_sws_luts *t)
{
    memcpy(t->y_table, c->yuvTable, 1024);
    for (int i = 0; i < 256; i++) {
        t->rV[i] = c->table_rV[i] - (uint8_t *)c->yuvTable;
        t->gU[i] = c->table_gU[i] - (uint8_t *)c->yuvTable;
        t->gV[i] = c->table_gV[i];
        t->bU[i] = c->table_bU[i] - (uint8_t *)c->yuvTable;
    }
}

/* 0: the context is one this path takes; the chroma lines of SWS_FULL_CHR_H_INT contexts are full width (chrDstHSubSample = 0,
 * utils.c:998-1014; yuv2rgb24_full_X_c, output.c:1392-1452) — not the 2:1 sharing these kernels restate */
static int taken(const SwsContext *c)
{
    return c->srcFormat == AV_PIX_FMT_YUV420P && c->dstFormat == AV_PIX_FMT_RGB24 && !(c->flags & SWS_FULL_CHR_H_INT);
}
static int describe(struct SwsContext *c, mi355_sws_desc *d, int special)
{
    if (!taken(c)) return -1;
    d->srcW = c->srcW; d->srcH = c->srcH; d->dstW = c->dstW; d->dstH = c->dstH;
    d->chrSrcW = c->chrSrcW; d->chrSrcH = c->chrSrcH; d->chrDstW = c->chrDstW;
    d->unscaled_special = special;                               /* yuv2rgb_c_24_rgb was selected (yuv2rgb.c:570) */
    d->hLum = (mi355_sws_filter){ c->hLumFilter, c->hLumFilterPos, c->hLumFilterSize, c->dstW };
    d->hChr = (mi355_sws_filter){ c->hChrFilter, c->hChrFilterPos, c->hChrFilterSize, c->chrDstW };
    d->vLum = (mi355_sws_filter){ c->vLumFilter, c->vLumFilterPos, c->vLumFilterSize, c->dstH };
    d->vChr = (mi355_sws_filter){ c->vChrFilter, c->vChrFilterPos, c->vChrFilterSize, c->dstH };
    luts_of(c, &d->luts);
    return 0;
}
/* the unscaled special converter was selected (ff_get_unscaled_swscale, utils.c:1045: sws_init_context returns before it
 * builds any filter bank) */
int mi355_sws_describe(struct SwsContext *c, mi355_sws_desc *d) { return describe(c, d, c->vLumFilter == NULL); }

#ifndef MI355_SWS_DESCRIBE_ONLY
static void t1_hscale(SwsContext *c, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize)
{
    (void)c; n_calls++;
    mi355_sws_hscale8to15(dst, dstW, src, filter, filterPos, filterSize);
}
static void t1_packedX(SwsContext *c, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize, const int16_t *chrFilter,
                       const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize, const int16_t **alpSrc, uint8_t *dest, int dstW, int y)
{
    mi355_sws_luts t;
    (void)alpSrc; (void)y; n_calls++;
    luts_of(c, &t);
    mi355_sws_yuv2rgb24_X(&t, lumFilter, lumSrc, lumFilterSize, chrFilter, chrUSrc, chrVSrc, chrFilterSize, dest, dstW);
}
static void t1_packed2(SwsContext *c, const int16_t *lumSrc[2], const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], const int16_t *alpSrc[2],
                       uint8_t *dest, int dstW, int yalpha, int uvalpha, int y)
{
    mi355_sws_luts t;
    (void)alpSrc; (void)y; n_calls++;
    luts_of(c, &t);
    mi355_sws_yuv2rgb24_2(&t, lumSrc, chrUSrc, chrVSrc, dest, dstW, yalpha, uvalpha);
}
static void t1_packed1(SwsContext *c, const int16_t *lumSrc, const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], const int16_t *alpSrc,
                       uint8_t *dest, int dstW, int uvalpha, int y)
{
    mi355_sws_luts t;
    (void)alpSrc; (void)y; n_calls++;
    luts_of(c, &t);
    mi355_sws_yuv2rgb24_1(&t, lumSrc, chrUSrc, chrVSrc, dest, dstW, uvalpha);
}

static int device_ready(void)
{
    const char *dev = getenv("MI355_DEVICE");
    return mi355_init(dev ? atoi(dev) : 0) == 0;
}

void ff_sws_init_mi355x(SwsContext *c)
{
    if (!taken(c) || c->hyscale_fast) return;
    if (!device_ready()) return;                                /* no usable MI355X: the reference's functions stay */
    /* only what the reference set to the functions these shims restate (a context whose selectors left one of them empty
     * uses another template) */
    if (!c->hyScale || !c->hcScale || !c->yuv2packedX || !c->yuv2packed2 || !c->yuv2packed1) return;
    c->hyScale = c->hcScale = t1_hscale;
    c->yuv2packedX = t1_packedX;
    c->yuv2packed2 = t1_packed2;
    c->yuv2packed1 = t1_packed1;
}

/* ---- whole pictures: SwsContext.swscale ------------------------------------------------------------------------------ */
#define MI355_SWS_SLOTS 256
typedef struct Bound {
    SwsContext *c;
    SwsFunc real;                 /* what the reference chose */
    int special;                  /* ... through ff_yuv2rgb_get_func_ptr */
    mi355_sws_ctx *dev;
    int failed;                   /* the device side does not take this context: the reference's function from now on */
    int busy;                     /* calls of mi355_sws_scale in flight on `dev` (under bound_mu): the device context is destroyed only at 0 */
    unsigned long stamp;          /* last use (diagnostics) */
    uint8_t y_table[1024];        /* the tables the device context was built from */
    int32_t gv0;
} Bound;
static Bound bound[MI355_SWS_SLOTS];
static pthread_mutex_t bound_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t bound_idle = PTHREAD_COND_INITIALIZER;
static unsigned long n_pictures, n_stamp;
unsigned long mi355_sws_glue_pictures(void) { return n_pictures; }   /* whole pictures converted on the device so far (diagnostics) */
int mi355_sws_glue_live_contexts(void)                                /* contexts that hold a device side right now (diagnostics) */
{
    int n = 0;
    pthread_mutex_lock(&bound_mu);
    for (int i = 0; i < MI355_SWS_SLOTS; i++) n += bound[i].c && bound[i].dev;
    pthread_mutex_unlock(&bound_mu);
    return n;
}

/* under bound_mu: wait until no call runs on the slot's device context, then drop it */
static void slot_release(Bound *b)
{
    while (b->busy) pthread_cond_wait(&bound_idle, &bound_mu);
    if (b->dev) mi355_sws_destroy(b->dev);
    memset(b, 0, sizeof(*b));
}
static Bound *bound_find(SwsContext *c, int create)
{
    Bound *free_slot = NULL;
    for (int i = 0; i < MI355_SWS_SLOTS; i++) {
        if (bound[i].c == c) return &bound[i];
        if (!free_slot && !bound[i].c) free_slot = &bound[i];
    }
    if (!create) return NULL;
    /* Table full: the new context stays UNBOUND (bind() hands back the reference's function).  A slot is never taken from another context: it may be
     * alive — its SwsContext.swscale points here, and without its slot a call could neither reach the device nor the reference's function
     * (an eviction turned such a call into "0 lines, no error").  Slots come back through sws_freeContext (--wrap) or when a new context
     * is built at a freed one's address; a host linked without the wrap that leaks 256 addresses runs further contexts on the CPU, correctly. */
    if (free_slot) { memset(free_slot, 0, sizeof(*free_slot)); free_slot->c = c; }
    return free_slot;
}
/* a selector runs for this context: sws_init_context() of a new context — possibly at the address of one that was freed */
static SwsFunc bind(SwsContext *c, SwsFunc real, int special)
{
    if (!real || !taken(c) || getenv("MI355_SWS_PLAIN")) return real;
    pthread_mutex_lock(&bound_mu);
    Bound *b = bound_find(c, 1);
    if (b) {
        slot_release(b);
        b->c = c; b->real = real; b->special = special; b->stamp = ++n_stamp;
    }
    pthread_mutex_unlock(&bound_mu);
    return b ? mi355_swsfunc_entry : real;
}

static int mi355_swsfunc(SwsContext *c, const uint8_t *src[], int srcStride[], int srcSliceY, int srcSliceH, uint8_t *dst[], int dstStride[])
{
    pthread_mutex_lock(&bound_mu);
    Bound *b = bound_find(c, 0);
    SwsFunc real = b ? b->real : NULL;
    mi355_sws_ctx *dev = NULL;
    /* the device path takes whole pictures with positive strides (a 2-D copy has no negative pitch; sws_scale() itself also takes bottom-up
     * pictures — vf_vflip makes them — and those go to the reference's function) */
    if (b && !b->failed && srcSliceY == 0 && srcSliceH == c->srcH &&
        srcStride[0] > 0 && srcStride[1] > 0 && srcStride[2] > 0 && dstStride[0] > 0) {
        if (b->dev && !b->busy && (memcmp(b->y_table, c->yuvTable, 1024) || b->gv0 != c->table_gV[0])) { mi355_sws_destroy(b->dev); b->dev = NULL; }
        if (!b->dev) {
            mi355_sws_desc d;
            if (device_ready() && describe(c, &d, b->special) == 0) b->dev = mi355_sws_create(&d);
            if (b->dev) { memcpy(b->y_table, c->yuvTable, 1024); b->gv0 = c->table_gV[0]; }
            else b->failed = 1;
        }
        dev = b->dev;
        if (dev) { b->busy++; b->stamp = ++n_stamp; }          /* the context stays until this call is back (a bind() or free for its address waits) */
    }
    pthread_mutex_unlock(&bound_mu);
    if (dev) {
        const int st[3] = { srcStride[0], srcStride[1], srcStride[2] };
        const int n = mi355_sws_scale(dev, src, st, dst[0], dstStride[0]);
        pthread_mutex_lock(&bound_mu);
        b->busy--;
        pthread_cond_broadcast(&bound_idle);
        pthread_mutex_unlock(&bound_mu);
        if (n > 0) { __sync_fetch_and_add(&n_pictures, 1); return n; }
        /* a failed copy / launch / allocation: this picture by the reference's function (mi355_sws_scale leaves the context usable) */
    }
    /* real == NULL: no slot for a context whose swscale points here — only a context used after sws_freeContext gets this far (slots are never taken away) */
    return real ? real(c, src, srcStride, srcSliceY, srcSliceH, dst, dstStride) : 0;
}

/* sws_freeContext: the context's device side goes with it (--wrap=sws_freeContext; without the wrap an entry stays until its address is
 * bound again or the table recycles it) */
void __real_sws_freeContext(SwsContext *c);
void __wrap_sws_freeContext(SwsContext *c)
{
    if (c) {
        pthread_mutex_lock(&bound_mu);
        Bound *b = bound_find(c, 0);
        if (b) slot_release(b);
        pthread_mutex_unlock(&bound_mu);
    }
    __real_sws_freeContext(c);
}

SwsFunc __real_ff_getSwsFunc(SwsContext *c);
SwsFunc __wrap_ff_getSwsFunc(SwsContext *c)
{
    SwsFunc f = __real_ff_getSwsFunc(c);
    if (getenv("MI355_SWS_LINES")) { ff_sws_init_mi355x(c); return f; }
    return bind(c, f, 0);
}
SwsFunc __real_ff_yuv2rgb_get_func_ptr(SwsContext *c);
SwsFunc __wrap_ff_yuv2rgb_get_func_ptr(SwsContext *c)
{
    SwsFunc f = __real_ff_yuv2rgb_get_func_ptr(c);
    if (getenv("MI355_SWS_LINES")) return f;
    return bind(c, f, 1);
}
#endif
