/*
 * mi355_hevc_batch.h — Tier 2 for the HEVC rows: the same wave-level code that
 * sits behind HEVCDSPContext (include/mi355dsp.h), launched over DEVICE-RESIDENT batches of independent
 * work items — one wavefront per item (eight edge segments per wavefront for the loop filter).
 *
 * The reference issues these calls one block at a time from hls_transform_unit / hevc_luma_mv_mpred /
 * ff_hevc_deblocking_filter / sao_filter_CTB (libavcodec/hevcdec.c, hevc_filter.c); HEVC's in-loop
 * filters are picture-parallel by construction (all vertical edges, then all horizontal ones; SAO reads
 * the deblocked picture and writes another), and transform / prediction blocks of a picture do not
 * depend on each other once the entropy decoder has produced them.  A bridge therefore collects one
 * job per call site and submits a picture's worth per launch; ordering between dependent stages
 * (residual after prediction, vertical before horizontal edges, SAO after deblocking) is the caller's
 * sequence of launches on one stream.  All pointers are device pointers; strides are in BYTES.
 */
#ifndef MI355_HEVC_BATCH_H
#define MI355_HEVC_BATCH_H

#include <stddef.h>
#include <stdint.h>
#include "mi355dsp.h"      /* the device error word */

#ifdef __cplusplus
extern "C" {
#endif

/* a13/a12: one transform unit: c->idct[]/idct_dc[]/transform_4x4_luma/dequant (hevcdsp.h:46-52), then
 * c->add_residual[] (:45) when `dst` is set (the tail of hls_transform_unit, hevcdec.c:1238-1260) */
enum { MI355_HEVC_TU_IDCT = 0, MI355_HEVC_TU_IDCT_DC = 1, MI355_HEVC_TU_DST4 = 2, MI355_HEVC_TU_SKIP = 3,
       MI355_HEVC_TU_BYPASS = 4,      /* cu_transquant_bypass: the coefficients ARE the residual (hls_residual_coding, hevcdec.c:1236-1260) */
       MI355_HEVC_TU_PCM = 5 };       /* pcm_sample: `coeffs` holds the samples themselves (put_pcm, hevcdsp_template.c:28-41): stored, not added */
typedef struct mi355_hevc_tu_job {
    int16_t *coeffs;          /* size x size, row-major; rewritten in place when dst == NULL */
    uint8_t *dst;             /* picture samples of the block, or NULL */
    int32_t dst_stride;
    uint8_t log2_size;        /* 2..5 */
    uint8_t col_limit;        /* as passed to c->idct[]: hevcdec.c:1178-1196 derives it from the last significant coefficient of the block's scan
                                 (last_x + last_y + 4, capped by size class), so rows col_limit + 4 .. size - 1 of `coeffs` hold zeros — the batched
                                 kernel does not fetch them (the reference's pruned passes read nothing of them but every fourth row, which is zero) */
    uint8_t kind;             /* MI355_HEVC_TU_* */
    uint8_t reserved;
} mi355_hevc_tu_job;
/* Two consecutive jobs share a wavefront: lists binned by (log2_size, kind, col_limit <= size / 2) run fastest (both
 * halves then take the same — pruned or full — transform path); any order is correct. */
int mi355_hevc_residual_batch_dev(const mi355_hevc_tu_job *d_jobs, int n, int bit_depth, void *stream);

/* a14: put_hevc_qpel / put_hevc_epel (hevcdsp.h:64-69): width x height samples at `src` (fractions mx,my)
 * to the 14-bit intermediate */
typedef struct mi355_hevc_mc_job {
    const uint8_t *src;       /* sample (0,0) of the block in the reference picture */
    int16_t *dst;
    int32_t src_stride, dst_stride;
    uint8_t width, height;    /* <= 64 */
    uint8_t mx, my;           /* luma: 0..3, chroma: 0..7 */
    uint8_t chroma;           /* 0: 8-tap qpel, 1: 4-tap epel */
    uint8_t reserved[3];
} mi355_hevc_mc_job;
int mi355_hevc_mc_batch_dev(const mi355_hevc_mc_job *d_jobs, int n, int bit_depth, void *stream);

/* a15: put_unweighted_pred / _avg / weighted_pred / _avg (hevcdsp.h:71-103) */
enum { MI355_HEVC_PRED_PUT = 0, MI355_HEVC_PRED_AVG = 1, MI355_HEVC_PRED_W = 2, MI355_HEVC_PRED_W_AVG = 3 };
typedef struct mi355_hevc_pred_job {
    uint8_t *dst;
    const int16_t *src1, *src2;   /* src2 only for the _avg kinds */
    int32_t dst_stride, src_stride;
    uint8_t width, height, kind, denom;
    int16_t w0, w1, o0, o1;
} mi355_hevc_pred_job;
int mi355_hevc_pred_batch_dev(const mi355_hevc_pred_job *d_jobs, int n, int bit_depth, void *stream);

/* a14 + a15 fused: what luma_mc / chroma_mc / hevc_luma_mv_mpred's callers do per prediction block (hevcdec.c:1395-1560 and
 * the put_unweighted_pred / weighted_pred calls that follow each MC call): interpolate the block from one or two reference
 * pictures and turn the 14-bit result(s) straight into samples.  The intermediate stays in LDS; results are identical to
 * mi355_hevc_mc_batch_dev followed by mi355_hevc_pred_batch_dev.  `kind`: MI355_HEVC_PRED_PUT / _W use reference 0 only,
 * _AVG / _W_AVG combine reference 0 (weight w0, offset o0) with reference 1 (w1, o1). */
typedef struct mi355_hevc_mcpred_job {
    const uint8_t *src0, *src1;   /* sample (0,0) of the block in each reference picture; src1 unused for the one-reference kinds */
    uint8_t *dst;
    int32_t src0_stride, src1_stride, dst_stride;
    uint8_t width, height;        /* <= 64 */
    uint8_t chroma;               /* 0: 8-tap qpel (fractions 0..3), 1: 4-tap epel (0..7), 2: 4-tap epel of BOTH chroma planes of the block: the
                                     second plane (Cr) at src0_b / src1_b / dst_b with the same strides, vector fractions and parameters —
                                     for the unweighted kinds only (MI355_HEVC_PRED_PUT, _AVG: the weighted ones carry per-plane weights).
                                     One job instead of two: the windows of both planes are fetched in one round trip and share every pass */
    uint8_t kind;                 /* MI355_HEVC_PRED_* */
    uint8_t mx0, my0, mx1, my1;
    uint8_t denom;
    uint8_t reserved[3];
    int16_t w0, w1, o0, o1;
    const uint8_t *src0_b, *src1_b;   /* chroma == 2: the second plane */
    uint8_t *dst_b;
} mi355_hevc_mcpred_job;
int mi355_hevc_mcpred_batch_dev(const mi355_hevc_mcpred_job *d_jobs, int n, int bit_depth, void *stream);

/* a16: hevc_{h,v}_loop_filter_{luma,chroma} (hevcdsp.h:104-113): one 8-sample edge segment pair.
 * Jobs of one launch must not touch the same samples (all vertical edges of a picture, or all
 * horizontal ones: ff_hevc_deblocking_filter's two passes). */
typedef struct mi355_hevc_lf_job {
    uint8_t *pix;             /* first sample on the q side */
    int32_t stride;
    int32_t beta;             /* luma only */
    int32_t tc[2];
    uint8_t no_p[2], no_q[2];
    uint8_t horizontal_edge;  /* 1: the edge is horizontal (hevc_h_loop_filter_*) */
    uint8_t chroma;
    uint8_t reserved[2];
} mi355_hevc_lf_job;
int mi355_hevc_deblock_batch_dev(const mi355_hevc_lf_job *d_jobs, int n, int bit_depth, void *stream);

/* a17: sao_band_filter[cls] / sao_edge_filter[cls] (hevcdsp.h:54-62) for one CTB component */
typedef struct mi355_hevc_sao_job {
    uint8_t *dst;
    const uint8_t *src;
    int32_t stride;           /* both pictures */
    int32_t width, height;
    int32_t borders[4];
    int32_t offset_val[5];
    uint8_t cls;              /* which of the four table slots: bit 0 = rows above, bit 1 = columns left */
    uint8_t edge;             /* 0 band, 1 edge */
    uint8_t c_idx, eo_class, band_position;
    uint8_t vert_edge, horiz_edge, diag_edge;
} mi355_hevc_sao_job;
int mi355_hevc_sao_batch_dev(const mi355_hevc_sao_job *d_jobs, int n, int bit_depth, void *stream);

/* a17, CTB level: sample adaptive offset of one component of one CTB's OWN samples in ONE job, so that a picture's SAO is one
 * launch that reads the deblocked picture once and writes every sample of the output picture once, in whole cache lines.
 * sao_filter_CTB (hevc_filter.c:188-314) filters a CTB's samples in up to four calls made while four different CTBs pass
 * through the decoder (the CTB itself: class 0 without its right / bottom strips; the CTB to its right: class 2 = that strip;
 * the CTB below: class 1; the one below-right: class 3), all with the OWNER's parameters.  A job lists those calls as
 * `pieces`: each names the CTB the reference makes the call for (first sample relative to the job's, size, picture-border
 * flags) and carries the flags the reference derives for that CTB and class.  A piece of type 0 (the owner has SAO off) is
 * copied.  The pieces of all jobs partition the picture: jobs are independent. */
typedef struct mi355_hevc_sao_piece {
    int32_t offset_val[5];
    uint8_t cls;              /* bit 0 = rows above, bit 1 = columns left of the piece's CTB (the reference's class) */
    uint8_t type;             /* 0 none (copy), 1 band, 2 edge */
    uint8_t eo_class, band_position;
    uint8_t vert_edge, horiz_edge, diag_edge;
    uint8_t borders;          /* bit e: the piece's CTB lies on the picture's left / top / right / bottom border (e = 0..3) */
    int16_t dx, dy;           /* first sample of the piece's CTB relative to dst / src (0 or the CTB size) */
    int16_t width, height;    /* of the piece's CTB */
} mi355_hevc_sao_piece;
typedef struct mi355_hevc_sao_ctb_job {
    uint8_t *dst;             /* first sample of the owner CTB */
    const uint8_t *src;
    int32_t stride;           /* both pictures, bytes */
    uint8_t c_idx, npieces, reserved[2];
    mi355_hevc_sao_piece piece[4];
} mi355_hevc_sao_ctb_job;
int mi355_hevc_sao_ctbs_dev(const mi355_hevc_sao_ctb_job *d_jobs, int n, int bit_depth, void *stream);

/* a11, batched: emulated_edge_mc (videodsp_template.c:24-96; HEVC callers hevcdec.c:1555, 1613-1630) — a block_w x block_h
 * window at (src_x, src_y) of a w x h plane (`src` = the plane's sample (0, 0)... see below) copied into `dst` with the
 * picture's border samples replicated.  `src` points at the window's first sample INSIDE OR OUTSIDE the plane, exactly as
 * the reference passes it (plane origin + src_y * stride + src_x samples).  The prediction jobs of blocks that reach over a
 * picture border then name `dst` as their source, as the decoder's luma_mc / chroma_mc do. */
typedef struct mi355_edge_emu_job {
    uint8_t *dst;
    const uint8_t *src;
    int32_t dst_stride, src_stride;   /* bytes */
    int32_t block_w, block_h;         /* samples */
    int32_t src_x, src_y, w, h;
} mi355_edge_emu_job;
int mi355_edge_emu_batch_dev(const mi355_edge_emu_job *d_jobs, int n, int bit_depth, void *stream);

/* a18: pred_planar[] / pred_dc / pred_angular[] (hevcdec.h:399-409, hevcpred_template.c:349-516) for one transform
 * block.  `top` / `left` point at element 0 of the neighbour arrays the reference's intra_pred() wrapper builds
 * (hevcpred_template.c:31-334: 2 * size samples each, element -1 = the corner); they live wherever the bridge put
 * them (typically a per-picture edge buffer filled from the reconstruction).  Jobs of one launch must not depend on
 * each other's output: the bridge submits one launch per dependency level of the picture's intra blocks (a block
 * needs its left, top-left, top and top-right neighbours), like mi355_h264_intra_schedule() does for H.264. */
enum { MI355_HEVC_INTRA_PLANAR = 0, MI355_HEVC_INTRA_DC = 1, MI355_HEVC_INTRA_ANGULAR = 2 };
typedef struct mi355_hevc_intra_job {
    uint8_t *dst;             /* sample (0,0) of the block */
    const uint8_t *top, *left;
    int32_t stride;           /* of dst, BYTES (the table's own entry points take samples; this header is bytes throughout) */
    uint8_t log2_size;        /* 2..5 */
    uint8_t kind;             /* MI355_HEVC_INTRA_* */
    uint8_t c_idx;            /* 0 luma: DC / angular edge smoothing applies */
    uint8_t mode;             /* angular: 2..34 */
} mi355_hevc_intra_job;
int mi355_hevc_intra_batch_dev(const mi355_hevc_intra_job *d_jobs, int n, int bit_depth, void *stream);

/* ---- a16, picture level: ff_hevc_hls_filter's deblocking half for a whole picture (deblocking_filter_CTB,
 * hevc_filter.c:337-505, with its tables tctable / betatable :35-45, chroma_tc :47-72, TC_CALC :332, get_qPy :166,
 * get_pcm :316) from the frame-level arrays the reference's slice decoder leaves behind — no per-edge work on the host.
 * All vertical edges of the picture, then all horizontal ones: the reference's per-CTB order (vertical edges of a CTB,
 * then the horizontal ones eight samples to the left) gives the same picture, because no vertical edge reads a sample a
 * horizontal edge filtered earlier wrote.  Edge parameters follow the CTB that contains the edge sample
 * (cur_tc_offset / left_tc_offset, :456-457, :488-490).  All pointers are device pointers. */
typedef struct mi355_hevc_db_params {   /* DBParams, hevcdec.h:359-362 */
    int32_t beta_offset, tc_offset;
} mi355_hevc_db_params;
typedef struct mi355_hevc_lf_picture {
    uint8_t *data[3];                  /* s->frame->data: filtered in place */
    int32_t linesize[3];               /* bytes */
    int32_t width, height;             /* sps->width / height (luma samples) */
    int32_t log2_ctb_size;             /* sps->log2_ctb_size */
    int32_t log2_min_cb_size;          /* sps->log2_min_cb_size, granularity of qp_y_tab */
    int32_t log2_min_pu_size;          /* sps->log2_min_pu_size, granularity of is_pcm / tab_mvf */
    int32_t min_cb_width;              /* sps->min_cb_width */
    int32_t min_pu_width, min_pu_height;
    int32_t ctb_width;                 /* sps->ctb_width */
    int32_t bs_width;                  /* s->bs_width = width >> 3 */
    const uint8_t *vertical_bs;        /* s->vertical_bs [(x >> 3) + (y >> 2) * bs_width] */
    const uint8_t *horizontal_bs;      /* s->horizontal_bs [(x + y * bs_width) >> 2] */
    const int8_t *qp_y_tab;            /* s->qp_y_tab */
    const uint8_t *is_pcm;             /* s->is_pcm (only read when pcmf) */
    const mi355_hevc_db_params *deblock;   /* s->deblock, one per CTB in raster order */
    int32_t pcmf;                      /* (sps->pcm_enabled_flag && sps->pcm.loop_filter_disable_flag) || pps->transquant_bypass_enable_flag */
    int32_t cb_qp_offset, cr_qp_offset;    /* pps->cb_qp_offset / cr_qp_offset */
} mi355_hevc_lf_picture;
/* `d_pics`: device array of `npics` descriptors (pictures of independent streams, or of one stream once their
 * reconstruction is complete).  4:2:0, bit depth 8..10.  Enqueues four launches on `stream` (luma and chroma, vertical
 * then horizontal) and returns.  max_width / max_height: the largest picture of the batch. */
int mi355_hevc_deblock_pictures_dev(const mi355_hevc_lf_picture *d_pics, int npics, int max_width, int max_height, int bit_depth, void *stream);

/* a16 + a17 fused per coding tree block: deblocking_filter_CTB (hevc_filter.c:337-505) and sao_filter_CTB (:188-314) of a block's OWN samples as ONE workgroup,
 * from the unfiltered reconstruction to the output picture: the block and eight samples around it are fetched into LDS, the vertical edges are filtered there,
 * then the horizontal ones (ff_hevc_deblocking_filter's order), then SAO reads the tile — the deblocked picture is never written and read back (what
 * mi355_hevc_deblock_pictures_dev followed by mi355_hevc_sao_ctbs_dev does: 2 x 12 KB written and read per 64x64 block at 10 bit), and there is one launch
 * instead of three.  The output picture is identical to those two entry points' (tests/test_hevc_chain_*.py against the reference's functions).
 *   d_pics[pic]  the picture as for mi355_hevc_deblock_pictures_dev, `data` = the RECONSTRUCTION, which is only read here;
 *   d_sao[sao[c]]  the block's job of component c as for mi355_hevc_sao_ctbs_dev: `dst` = the block's first sample in the OUTPUT picture, `src` is not used.
 * The jobs must be of the whole-region forms (every piece of the owner's type, no restored slice / tile / pcm edge in an edge-offset job, offsets within a signed
 * byte) — pictures whose slices forbid filtering across their edges take the two separate entry points.  A job of another form is refused on the device: that
 * component of the block is left unwritten and MI355_ERR_FILTER_CTB_FORM is set in the device's error word (mi355_sync returns MI355_E_DEVICE_FAULT). */
typedef struct mi355_hevc_filter_ctb_job {
    int32_t pic;              /* index into d_pics */
    uint16_t x0, y0;          /* luma position of the block's first sample */
    uint32_t sao[3];          /* index into d_sao: the block's luma / Cb / Cr job */
} mi355_hevc_filter_ctb_job;
int mi355_hevc_filter_ctbs_dev(const mi355_hevc_lf_picture *d_pics, const mi355_hevc_filter_ctb_job *d_ctbs, int n_ctbs, const mi355_hevc_sao_ctb_job *d_sao,
                               int log2_ctb_size, int bit_depth, void *stream);


/* ---- a16, boundary strengths: ff_hevc_deblocking_boundary_strengths (hevc_filter.c:585-725) + boundary_strength
 * (:507-583) for every 4-sample edge segment of a picture at once, from the motion field and the geometry of the blocks
 * the reference calls that function for (transform-tree leaves, hevcdec.c:1451, and coding units without residual,
 * :2095, :2182).  The bridge hands over, per 4x4 luma cell, what the walk knows about the cell's left and top side:
 *   MI355_HEVC_EDGE_L_BLOCK / _T_BLOCK   the side is the left / top edge of such a block, on the 8x8 grid, and filtered
 *                                         (boundary_left / boundary_upper true: slice and tile boundaries with filtering
 *                                         across them switched off are simply not marked)          -> tu_border = 1
 *   MI355_HEVC_EDGE_L_INNER / _T_INNER   the side lies on the 8x8 grid INSIDE a block whose first prediction unit is
 *                                         not intra (the "TU internal PU boundaries" loops :629-651, :694-722) -> tu_border = 0
 * Reference pictures are compared by identity: ref_poc[list][ref_idx] as RefPicList.list[] holds it (one table per call:
 * pictures whose slices use different lists take one call per group of slices).  Every cell side on the 8x8 grid gets a
 * value (0 where not marked), so vertical_bs / horizontal_bs need no clearing; layout as in mi355_hevc_lf_picture. */
enum { MI355_HEVC_EDGE_L_BLOCK = 1, MI355_HEVC_EDGE_T_BLOCK = 2, MI355_HEVC_EDGE_L_INNER = 4, MI355_HEVC_EDGE_T_INNER = 8 };
typedef struct mi355_hevc_mvfield {     /* MvField, hevcdec.h:326-331: same size and field offsets */
    int16_t mv[2][2];                   /* [list][x, y] quarter samples */
    int8_t ref_idx[2];
    int8_t pred_flag[2];
    uint8_t is_intra;
    uint8_t pad[3];
} mi355_hevc_mvfield;
typedef struct mi355_hevc_bs_picture {
    int32_t width, height;              /* luma samples, multiples of 8 */
    int32_t log2_min_pu_size, log2_min_tb_size;
    int32_t min_pu_width, min_tb_width;
    int32_t bs_width;                   /* width >> 3 */
    int32_t reserved;
    const mi355_hevc_mvfield *tab_mvf;  /* s->ref->tab_mvf, min-PU granularity */
    const uint8_t *cbf_luma;            /* s->cbf_luma, min-TB granularity */
    const uint8_t *edge_flags;          /* MI355_HEVC_EDGE_* per 4x4 luma cell, raster, width / 4 cells per row */
    int32_t ref_poc[2][16];             /* RefPicList[list].list[ref_idx] */
    uint8_t *vertical_bs, *horizontal_bs;   /* 2 * bs_width * ((height >> 3) + 1) bytes each */
} mi355_hevc_bs_picture;
int mi355_hevc_boundary_strengths_dev(const mi355_hevc_bs_picture *d_pics, int npics, int max_width, int max_height, void *stream);

/* ---- a18, wrapper level: HEVCPredContext.intra_pred[log2_size - 2](s, x0, y0, c_idx) (hevcpred_template.c:31-334) for a
 * list of transform blocks: neighbour availability (the caller's lc->na flags narrowed by the z-scan order test :93-97
 * and, with constrained_intra_pred, by the motion field's is_intra :105-151), the gather of the 4 * size + 1 neighbour
 * samples from the picture (:152-170), the constrained-intra substitution (:172-232), the inference of unavailable
 * samples (:233-270), the [1 2 1] / strong smoothing (:272-318) and the prediction itself (:320-333) — what round 1 left
 * to the host.  Blocks of one launch must not depend on each other's output (one launch per dependency level of the
 * picture's intra blocks; a block reads the column left of it from y0 - 1 to y0 + 2 * size - 1 and the row above it
 * likewise).  All pointers are device pointers. */
enum { MI355_HEVC_CAND_BOTTOM_LEFT = 1, MI355_HEVC_CAND_LEFT = 2, MI355_HEVC_CAND_UP_LEFT = 4, MI355_HEVC_CAND_UP = 8,
       MI355_HEVC_CAND_UP_RIGHT = 16 };
typedef struct mi355_hevc_intra_picture {
    uint8_t *data[3];                   /* s->frame->data: reconstruction so far, predicted in place */
    int32_t linesize[3];                /* bytes */
    int32_t width, height;              /* sps->width / height (luma samples) */
    int32_t hshift, vshift;             /* sps->hshift[1] / vshift[1] (4:2:0: 1, 1) */
    int32_t log2_min_pu_size;           /* granularity of tab_mvf */
    int32_t log2_min_tb_size;           /* granularity of min_tb_addr_zs */
    int32_t min_pu_width, min_pu_height;
    int32_t min_tb_width;
    int32_t constrained_intra_pred;     /* pps->constrained_intra_pred_flag */
    int32_t strong_intra_smoothing;     /* sps->sps_strong_intra_smoothing_enable_flag */
    int32_t reserved;
    const mi355_hevc_mvfield *tab_mvf;  /* s->ref->tab_mvf: only is_intra is read, and only with constrained_intra_pred */
    const int32_t *min_tb_addr_zs;      /* pps->min_tb_addr_zs [y_tb * min_tb_width + x_tb] */
} mi355_hevc_intra_picture;
typedef struct mi355_hevc_intra_block {
    int32_t pic;                        /* index into the descriptor array */
    uint16_t x0, y0;                    /* LUMA position of the block, as the reference passes it (chroma too) */
    uint8_t log2_size;                  /* 2..5, in samples of the plane */
    uint8_t c_idx;
    uint8_t mode;                       /* 0 planar, 1 DC, 2..34 angular (lc->tu.cur_intra_pred_mode / lc->pu.intra_pred_mode_c) */
    uint8_t cand;                       /* MI355_HEVC_CAND_*: lc->na as ff_hevc_set_neighbour_available left it */
} mi355_hevc_intra_block;
int mi355_hevc_intra_pred_blocks_dev(const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks, int n,
                                     int bit_depth, void *stream);
/* The same with each block's residual behind its prediction in ONE launch — an intra transform block as hls_transform_unit runs it
 * (hevcdec.c:1002-1030 s->hpc.intra_pred[], then :1238-1260 the transform and add_residual of the same block): d_tus[i] is the unit of
 * d_blocks[i] (`dst` = the block's samples, as for mi355_hevc_residual_batch_dev), `coeffs` NULL for a block that has none.  A caller
 * that walks dependency levels (contrib/libav/mi355_hevc_bridge.c) issues one launch per level instead of two. */
int mi355_hevc_intra_recon_blocks_dev(const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks,
                                      const mi355_hevc_tu_job *d_tus, int n, int bit_depth, void *stream);

/* One dependency level of a batch of pictures in ONE launch: n_mc prediction jobs (as mi355_hevc_mcpred_batch_dev takes them), n_tus transform
 * units (mi355_hevc_residual_batch_dev), n_blocks intra blocks with their units (mi355_hevc_intra_recon_blocks_dev) — the three kinds of a level
 * write disjoint samples and run side by side.  Any of the three may be empty (pointer NULL, count 0); all empty: -1.  What the reference does
 * block by block in hls_coding_unit / hls_transform_unit (hevcdec.c:1002-1030, :1238-1260, :1695-1850) a caller replays level by level. */
int mi355_hevc_recon_level_dev(const mi355_hevc_mcpred_job *d_mc, int n_mc, const mi355_hevc_tu_job *d_tus, int n_tus,
                               const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks,
                               const mi355_hevc_tu_job *d_block_tus, int n_blocks, int bit_depth, void *stream);

/* EVERY dependency level of a batch of pictures in ONE launch.  d_levels[l] names level l's jobs inside the three job arrays (the arrays in level order,
 * as a caller that walks levels has them anyway) and `first_wg`, the number of workgroups of the levels before it: a level takes
 * n_mc + (n_tu + 1) / 2 + n_in workgroups (what mi355_hevc_recon_level_dev launches for it), first_wg of level 0 is 0, n_workgroups is the total.
 * Results are those of mi355_hevc_recon_level_dev called for each level in turn: a workgroup of level l starts its job when every workgroup of the
 * levels before has finished and its samples are visible.  An all-intra picture is hundreds to thousands of levels of a handful of blocks each
 * (contrib/libav/mi355_hevc_bridge.c).  Measured on MI355X: 3.0 us per level for levels of two workgroups, growing with the level's size (7 us at 32 workgroups,
 * 147 us at 1000: every workgroup counts itself into one word); back-to-back calls of mi355_hevc_recon_level_dev are 3.5 - 4 us per level whatever its size when
 * the caller keeps the queue ahead of the device.  For chains of small levels whose caller cannot do that.
 * Should the device ever start workgroups in an order that leaves a level waiting for good, the wait ends after about a second and
 * MI355_ERR_WAIT_EXPIRED is set in the error word (mi355dsp.h): the next mi355_sync / mi355_event_sync returns MI355_E_DEVICE_FAULT and the caller
 * repeats the batch level by level. */
typedef struct mi355_hevc_level {
    uint32_t first_wg;                  /* workgroups of levels 0 .. l - 1 */
    uint32_t mc0, n_mc;                 /* d_mc[mc0 .. mc0 + n_mc) */
    uint32_t tu0, n_tu;                 /* d_tus[tu0 .. tu0 + n_tu) */
    uint32_t in0, n_in;                 /* d_blocks / d_block_tus[in0 .. in0 + n_in) */
    uint32_t reserved;
} mi355_hevc_level;
int mi355_hevc_recon_levels_dev(const mi355_hevc_level *d_levels, int n_levels, int n_workgroups, const mi355_hevc_mcpred_job *d_mc,
                                const mi355_hevc_tu_job *d_tus, const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks,
                                const mi355_hevc_tu_job *d_block_tus, int bit_depth, void *stream);

/* ---- a12-a15 fused per coding tree block: what hls_coding_quadtree (hevcdec.c:2202-2290) does for the INTER coding units of one CTB —
 * every prediction block (hls_prediction_unit :1695-1885: luma_mc / chroma_mc + put_unweighted_pred / weighted_pred), then every transform
 * unit (hls_transform_unit :1238-1260: idct / transform_skip / ..., add_residual) — as ONE workgroup: the CTB's samples are predicted into an
 * LDS tile, the residuals are added there, and the tile leaves for the picture once, in whole lines.  The prediction is never written to
 * the picture and read back (what mi355_hevc_mcpred_batch_dev followed by mi355_hevc_residual_batch_dev does: 2 x 12 KB per 64x64 CTB at
 * 10 bit), and there is one launch instead of two.  Results are identical to those two launches.
 *
 * A CTB job names its prediction jobs d_mc[first_mc .. first_mc + n_mc) and its transform units d_tus[first_tu .. first_tu + n_tu): the
 * SAME records the two batch entry points take, `dst` pointing into the picture — every `dst` of a CTB's jobs must lie inside that CTB
 * (the kernel turns it into tile coordinates through dst[] / stride[] of the CTB job).  Prediction jobs run before transform units, each
 * list in any order among the workgroup's waves: blocks of one list must not overlap.
 * MI355_HEVC_CTB_PARTIAL: the jobs do not cover every sample of the CTB (intra coding units reconstructed by other launches, a picture whose
 * last rows no block covers): the tile is first loaded from the picture, so uncovered samples come back unchanged.  Without the flag the
 * CTB is not read.
 * Fast paths (the matrix unit on byte planes, libav_amd/csrc/hevc_ctb_fast.h): one-reference unweighted prediction blocks whose width and
 * height are multiples of 16 samples of their plane, and 16x16 / 32x32 inverse DCTs (MI355_HEVC_TU_IDCT) with 16-byte aligned
 * coefficients; everything else runs the bodies of the two batch kernels on the tile.  Precondition for the transform units, as for
 * mi355_hevc_residual_batch_dev and for the same reason (hevcdec.c:1249-1256): a unit's coefficients lie in rows AND columns
 * 0 .. col_limit + 3 of its block (the diagonal scan leaves none beyond either). */
enum { MI355_HEVC_CTB_PARTIAL = 1 };
typedef struct mi355_hevc_ctb_job {
    uint8_t *dst[3];              /* first sample of the CTB in the picture's three planes */
    int32_t stride[3];            /* bytes */
    uint16_t width, height;       /* luma samples of the CTB inside the picture: 1 << log2_ctb_size except in the last column / row */
    uint8_t log2_ctb_size;        /* 4..6 */
    uint8_t flags;                /* MI355_HEVC_CTB_* */
    uint8_t reserved[2];
    uint32_t first_mc, n_mc;
    uint32_t first_tu, n_tu;
    uint32_t reserved1;
} mi355_hevc_ctb_job;
/* Two launches: the matrix-path kernel takes the blocks ALL of whose jobs are of its shapes (eight waves per block, 4.6 KB of scratch per wave), the general
 * kernel follows on the same list and takes exactly the others (every body of the batch kernels compiled in: four waves, larger scratch, more registers).  A
 * caller that KNOWS its list holds only matrix-path shapes (it made the jobs) passes MI355_HEVC_RECON_UNIFORM and saves the second launch (a launch of
 * workgroups that look at their block's records and leave: ~40 us per 130 k blocks); the promise is checked on the device — a block that breaks it is left
 * untouched and MI355_ERR_CTB_NOT_UNIFORM is set in the device's error word (include/mi355dsp.h: mi355_sync returns MI355_E_DEVICE_FAULT). */
enum { MI355_HEVC_RECON_UNIFORM = 1 };
int mi355_hevc_recon_ctbs_dev(const mi355_hevc_ctb_job *d_ctbs, int n_ctbs, const mi355_hevc_mcpred_job *d_mc, const mi355_hevc_tu_job *d_tus,
                              int bit_depth, unsigned flags, void *stream);

#ifdef __cplusplus
}
#endif
#endif
