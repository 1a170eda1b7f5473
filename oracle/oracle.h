/*
 * oracle.h — entry points of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Same init-function shapes as the reference's ff_*_init (file:line in each .c)
 * so tests can swap {reference build, oracle, MI355X backend} behind one table.
 */
#ifndef MI355_ORACLE_H
#define MI355_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#include "../include/mi355_abi.h"
#ifdef __cplusplus
extern "C" {
#endif
int  oracle_scan8(int i);
void oracle_h264dsp_init(H264DSPContext *c, int bit_depth, int chroma_format_idc);   /* h264dsp.c:57 */
void oracle_h264qpel_init(H264QpelContext *c, int bit_depth);                         /* h264qpel.c:37 */
void oracle_h264chroma_init(H264ChromaContext *c, int bit_depth);                     /* h264chroma.c:39 */
void oracle_h264_pred_init(H264PredContext *h, int codec_id, int bit_depth, int chroma_format_idc); /* h264pred.c:402 */
void oracle_videodsp_init(VideoDSPContext *c, int bpc);
void oracle_hevc_dsp_init(HEVCDSPContext *c, int bit_depth);                         /* hevcdsp.c:136 */
void oracle_hevc_pred_init(HEVCPredContext *h, int bit_depth);                       /* hevcpred.c:37 */                               /* videodsp.c:35 */
void oracle_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t stride, int size, int mx, int my, int avg);
void oracle_h264_qpel2(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride, int size, int mx, int my, int avg);
void oracle_h264_chroma_mc2(uint8_t *dst, ptrdiff_t dst_stride, const uint8_t *src, ptrdiff_t src_stride, int h, int x, int y, int w, int avg);
/* frame level (oracle_h264frame.c): the mi355_h264_frame pointers are HOST pointers here */
struct mi355_h264_frame;
void oracle_h264_recon_frame(const struct mi355_h264_frame *f);
void oracle_h264_deblock_frame(const struct mi355_h264_frame *f);
/* fill the driver's four tables from other init functions (the reference's, for bench.py's cpu_baseline); NULL = restated */
void oracle_h264frame_bind_tables(void (*dsp_init)(H264DSPContext *, int, int), void (*qpel_init)(H264QpelContext *, int),
                                  void (*chroma_init)(H264ChromaContext *, int), void (*pred_init)(H264PredContext *, int, int, int));
#ifdef __cplusplus
}
#endif
#endif
