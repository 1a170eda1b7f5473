/*
 * mi355dsp.h — C ABI of libmi355dsp.so, the MI355X (gfx950) backend for libav's
 * H.264 / HEVC DSP pointer tables and libswscale's inner loops.
 *
 * Plain C: pointers and sizes only.  Two tiers behind one library:
 *
 *  Tier 1 — `ff_<table>_init_mi355x()`: per-architecture init hooks in the style of
 *    the reference's ff_h264dsp_init_x86() (libavcodec/x86/h264dsp_init.c), called
 *    right after the C defaults are filled (libavcodec/h264dsp.c:139-142 is where the
 *    reference calls its arch hooks).  They overwrite the table entries this backend
 *    implements with functions of IDENTICAL signature and semantics that run the
 *    work on the GPU synchronously (host pointers in, host pointers out).  This is
 *    the function-pointer surface the reference's decoder already calls; it is the
 *    parity surface, not the fast path (one launch per call).
 *
 *  Tier 2 — `mi355_h264_*` batched frame reconstruction on device-resident data
 *    (declared in mi355_h264_frame.h): the throughput path.
 *
 * There is NO CPU fallback: every entry point aborts with a message if no gfx950
 * device was initialised with mi355_init().
 */
#ifndef MI355DSP_H
#define MI355DSP_H

#include "mi355_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Select and initialise GPU `device` (0-based).  0 on success, <0 if there is no such
 * device or it is not gfx950. */
int mi355_init(int device);
int mi355_device_cus(void);
/* One host process, several GPUs: mi355_init() names the process default; a thread that calls mi355_set_device(d) works on
 * device d from then on (allocations, copies, launches, Tier-1 staging), -1 returns it to the default.  Sessions, groups and
 * swscale contexts remember the device of the thread that created them and switch to it inside their entry points.  0 on
 * success, <0 as mi355_init(). */
int mi355_set_device(int device);
int mi355_get_device(void);            /* the device this thread works on now: its own, or the process default */
int mi355_get_thread_device(void);     /* the thread's OWN setting: -1 while it follows the process default (what a scope that switches
                                          devices temporarily must put back with mi355_set_device) */
int mi355_device_count(void);

/* The device's ERROR WORD.  A kernel that cannot do what it was launched for — a bounded wait on another workgroup that ran out
 * (MI355_ERR_WAIT_EXPIRED: the picture it was working on is NOT valid), a coding tree block outside the shapes its caller promised
 * (MI355_ERR_CTB_NOT_UNIFORM: the block was left untouched), a SAO job outside the forms mi355_hevc_filter_ctbs_dev takes (MI355_ERR_FILTER_CTB_FORM: that component of
 * the block was left unwritten) — ORs its bit into one word per device, in pinned host memory.  mi355_sync(),
 * mi355_event_sync() and mi355_h264_pipelines_sync() read it after their wait and return MI355_E_DEVICE_FAULT (-5) when it is set;
 * mi355_error_word_take() returns the bits and clears them (what a caller does before it repeats the batch another way). */
enum { MI355_ERR_WAIT_EXPIRED = 1, MI355_ERR_CTB_NOT_UNIFORM = 2, MI355_ERR_FILTER_CTB_FORM = 4, MI355_ERR_TEST = 0x40000000 };
enum { MI355_E_DEVICE_FAULT = -5 };
unsigned mi355_error_word_take(void);      /* bits set since the last take (this thread's device); 0: none */
unsigned mi355_error_word_peek(void);
/* test hook: a one-thread kernel on `stream` that ORs `bits` into the word the way a failing kernel does */
int mi355_error_word_inject(unsigned bits, void *stream);

/* replaces ff_h264dsp_init_{x86,arm,...}   libavcodec/h264dsp.h:119-128, call site h264dsp.c:139-142 */
void ff_h264dsp_init_mi355x(H264DSPContext *c, const int bit_depth, const int chroma_format_idc);
/* replaces ff_h264qpel_init_{x86,...}      libavcodec/h264qpel.h:34-37, call site h264qpel.c:102-109 */
void ff_h264qpel_init_mi355x(H264QpelContext *c, int bit_depth);
/* replaces ff_h264chroma_init_{x86,...}    libavcodec/h264chroma.h:36-39, call site h264chroma.c:47-54 */
void ff_h264chroma_init_mi355x(H264ChromaContext *c, int bit_depth);
/* replaces ff_h264_pred_init_{x86,...}     libavcodec/h264pred.h:116-121, call site h264pred.c:573-578 */
void ff_h264_pred_init_mi355x(H264PredContext *h, int codec_id, const int bit_depth, const int chroma_format_idc);
/* replaces ff_videodsp_init_{x86,...}      libavcodec/videodsp.h:68-72, call site videodsp.c:42-49 */
void ff_videodsp_init_mi355x(VideoDSPContext *ctx, int bpc);
/* replaces ff_hevc_dsp_init_{x86,arm}      libavcodec/hevcdsp.h:118-119, call site hevcdsp.c:248-253 */
void ff_hevc_dsp_init_mi355x(HEVCDSPContext *c, const int bit_depth);
/* replaces the tail of ff_hevc_pred_init   libavcodec/hevcpred.c:37-73 (no arch hook exists in the reference) */
void ff_hevc_pred_init_mi355x(HEVCPredContext *hpc, int bit_depth);

#ifdef __cplusplus
}
#endif
#endif /* MI355DSP_H */
