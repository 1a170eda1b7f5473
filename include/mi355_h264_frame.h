/*
 * mi355_h264_frame.h — Tier-2: batched H.264 macroblock reconstruction + deblocking
 * on device-resident data (the throughput path; C ABI of libmi355dsp.so).
 *
 * What it replaces in the reference: the per-macroblock body of the slice decoder
 * after entropy decoding —
 *     ff_h264_hl_decode_mb()   libavcodec/h264_mb.c:798 (hl_decode_mb, h264_mb_template.c:41,
 *                               8-bit "simple" variant: frame or field pictures without MBAFF, 4:2:0, not lossless)
 *     loop_filter() / ff_h264_filter_mb()   libavcodec/h264_slice.c:2198, h264_loopfilter.c:716
 * for ALL macroblocks of a batch of independent pictures at once, in the reference's
 * own "reconstruct the whole picture, then filter it" mode (H264Context.postpone_filter,
 * h264_slice.c:2344-2345, :2570-2586): pass 1 writes unfiltered samples to `recon`,
 * pass 2 filters `recon` into `dst`.  The host decoder (entropy decoding, MV prediction,
 * reference lists) stays the reference's C code and fills the records below from the
 * state it already holds when it would call ff_h264_hl_decode_mb();
 * INTEGRATION.md shows that bridge.
 *
 * All pointers inside mi355_h264_frame are DEVICE pointers.  Pictures are planar 8-bit
 * 4:2:0 with byte strides, exactly like the reference's AVFrame planes, so `dst` can be
 * handed back (or copied out) unchanged.
 */
#ifndef MI355_H264_FRAME_H
#define MI355_H264_FRAME_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mb_type bits: the reference's own encoding (libavcodec/mpegutils.h:51-71,
 * MB_TYPE_8x8DCT libavcodec/h264dec.h), so cur_pic.mb_type[] can be copied as is. */
#define MI355_MB_INTRA4x4    0x00000001u  /* with MI355_MB_8x8DCT: Intra 8x8 */
#define MI355_MB_INTRA16x16  0x00000002u
#define MI355_MB_INTRA_PCM   0x00000004u
#define MI355_MB_16x16       0x00000008u
#define MI355_MB_16x8        0x00000010u
#define MI355_MB_8x16        0x00000020u
#define MI355_MB_8x8         0x00000040u
#define MI355_MB_P0L0        0x00001000u
#define MI355_MB_P1L0        0x00002000u
#define MI355_MB_P0L1        0x00004000u
#define MI355_MB_P1L1        0x00008000u
#define MI355_MB_8x8DCT      0x01000000u
#define MI355_MB_INTRA       (MI355_MB_INTRA4x4 | MI355_MB_INTRA16x16 | MI355_MB_INTRA_PCM)

/* mi355_h264_mb.flags */
#define MI355_MBF_LEFT_EDGE  0x01  /* the left MB edge is filtered (neighbour exists and the slice's
                                      disable_deblocking_filter_idc allows it: sl->left_type != 0) */
#define MI355_MBF_TOP_EDGE   0x02  /* same for the top edge (sl->top_type != 0) */
#define MI355_MBF_NO_DEBLOCK 0x04  /* sl->deblocking_filter == 0 for this MB's slice: copy only */
#define MI355_MBF_WEIGHTED   0x08  /* sl->pwt.use_weight != 0: the slice table must be consulted for MC */
/* transform bypass (read by mi355_h264_decode_frames_wide_dev only — the 8-bit kernels never see such pictures) */
#define MI355_MBF_BYPASS     0x10  /* sl->qscale == 0 && sps->transform_bypass: the coefficients are the residual itself, added without
                                      a transform and without clipping (h264_mb_template.c:51, h264addpx_template.c:30-72); the Intra16x16
                                      DC levels then sit at the reference's dc_mapping[] slots (h264_mb.c:706-722), not at mi355_luma_dc_slot() */
#define MI355_MBF_BYPASS_PRED 0x20 /* ... in a High 4:4:4 Predictive stream (sps->profile_idc == 244): vertical / horizontal intra prediction
                                      becomes the running sum of the pred*_add functions (h264pred_template.c:1127-1354; h264_mb.c:636, :668, :737) */
#define MI355_MBF_BYPASS_X264OLD 0x40 /* ... decoded with h->x264_build < 151: Intra 8x8 sums start from the UNFILTERED edge (pred8x8l_add
                                      instead of pred8x8l_filter_add, h264_mb.c:637-643) */
#define MI355_MBF_FILTER_OWN_SLICE 0x80 /* sl->deblocking_filter == 2: edges towards other slices are not filtered.  The kernels of plain frames and fields
                                      read that from LEFT_EDGE / TOP_EDGE; the loop filter of MBAFF frames decides which macroblock is "left" and "above"
                                      itself (fill_filter_caches, h264_slice.c:2066-2113) and compares slice_id */

/* mi355_h264_mb.sub_mb_type[i] (only for MI355_MB_8x8): shape of 8x8 quadrant i */
#define MI355_SUB_8x8  0
#define MI355_SUB_8x4  1
#define MI355_SUB_4x8  2
#define MI355_SUB_4x4  3
#define MI355_SUB_L0   0x10  /* quadrant predicted from list 0 */
#define MI355_SUB_L1   0x20  /* quadrant predicted from list 1 */

/* nnz_mask bit numbers */
#define MI355_NNZ_LUMA(i)   (i)          /* i = 0..15, the reference's block index (scan8 order) */
#define MI355_NNZ_CB(j)     (16 + (j))   /* j = 0..3 */
#define MI355_NNZ_CR(j)     (20 + (j))
#define MI355_NNZ_LUMA_DC   24           /* non_zero_count_cache[scan8[LUMA_DC_BLOCK_INDEX]] != 0 */
#define MI355_NNZ_CB_DC     25
#define MI355_NNZ_CR_DC     26

/* One macroblock: exactly 64 bytes (offsets in the comments).  Field sources in the reference at the time
 * ff_h264_hl_decode_mb() runs are given on the right (sl = H264SliceContext). */
typedef struct mi355_h264_mb {
    uint32_t mb_type;                    /*  0 h->cur_pic.mb_type[mb_xy] */
    uint32_t nnz_mask;                   /*  4 MI355_NNZ_*: which blocks carry coefficients; for MI355_MB_8x8DCT
                                               the four bits of a quadrant are equal (what the loop filter sees
                                               after fill_filter_caches(), h264_slice.c:2160-2191) */
    uint16_t cbp;                        /*  8 h->cbp_table[mb_xy] (low 6 bits used) */
    int8_t   qp;                         /* 10 h->cur_pic.qscale_table[mb_xy] */
    uint8_t  flags;                      /* 11 MI355_MBF_* */
    int8_t   slice_alpha_c0_offset;      /* 12 sl->slice_alpha_c0_offset */
    int8_t   slice_beta_offset;          /* 13 sl->slice_beta_offset */
    uint8_t  intra16x16_pred_mode;       /* 14 sl->intra16x16_pred_mode (table slot, after availability remap) */
    uint8_t  chroma_pred_mode;           /* 15 sl->chroma_pred_mode */
    uint16_t topleft_samples_available;  /* 16 sl->topleft_samples_available */
    uint16_t topright_samples_available; /* 18 sl->topright_samples_available */
    uint8_t  sub_mb_type[4];             /* 20 MI355_SUB_*, from sl->sub_mb_type[] */
    int8_t   ref_idx[2][4];              /* 24 sl->ref_cache[list][scan8[4*i]] per 8x8 quadrant; <0 = unused */
    uint32_t dc_qmul[3];                 /* 32 pps->dequant4_coeff[{0 | 1,2 intra | 4,5 inter}][qp][0]:
                                               luma DC (Intra16x16), Cb DC, Cr DC (h264_mb.c:706-708,
                                               h264_mb_template.c:239-244) */
    uint8_t  slice_id;                   /* 44 index into mi355_h264_frame.slices */
    uint8_t  intra_level;                /* 45 informational, written by mi355_h264_intra_schedule(): 0 for inter MBs;
                                               for intra MBs 1 + max(level of the intra MBs among left, top-left, top,
                                               top-right), SATURATED at 255 (an all-intra 2160p picture reaches 508).
                                               The device never reads it: the schedule is intra_list / intra_level_start */
    uint8_t  qpc[2];                     /* 46 get_chroma_qp(pps, {0,1}, qp): this MB's Cb / Cr QP; MUST equal
                                               slices[slice_id].chroma_qp_table[p][qp] (I_PCM: qp = 0), the loop filter
                                               reads it instead of the table */
    union {                              /* 48 */
        int8_t intra4x4_pred_mode[16];   /*    intra MBs: sl->intra4x4_pred_mode_cache[scan8[i]]; Intra 8x8 uses
                                               i = 0,4,8,12 */
        struct {
            uint8_t ref_pic[2][4];       /*    inter MBs: picture slot (index into mi355_h264_frame.ref[]) of each
                                               quadrant's reference = slices[slice_id].ref_slot[list][ref_idx],
                                               0xFF when unused; what the loop filter compares (h->ref2frm) */
            int8_t  chroma_dy[2][4];     /*    field pictures: what mc_dir_part adds to the chroma vector's y (eighth samples of a chroma
                                               line) when the reference field has the other parity: 2 * (parity of this field -
                                               parity of the reference), h264_mb.c:287-291; 0 in frame pictures */
        } inter;
    } u;
} mi355_h264_mb;

/* Coefficients: 384 int16 per macroblock = sl->mb with the chroma planes packed
 * (luma block i at [16*i], Cb block j at [256+16*j], Cr block j at [320+16*j]),
 * dequantised and stored transposed exactly as the reference's residual decoder
 * leaves them.  Two conventions on top:
 *  - Intra16x16: the 16 luma DC levels of sl->mb_luma_dc[0] are placed in the DC slots
 *    (index 0 of each 4x4 block): level k goes to coefficient index mi355_luma_dc_slot(k)
 *    — the position ff_h264_luma_dc_dequant_idct() writes its k-th output to, so the
 *    device transforms them in place.
 *  - I_PCM: the 384 raw sample bytes (Y 256, Cb 64, Cr 64) occupy the first 384 bytes.
 * Blocks whose nnz_mask bit is clear must have all AC coefficients zero. */
#define MI355_H264_COEFS_PER_MB 384
static inline int mi355_luma_dc_slot(int k)
{
    static const uint8_t col[4] = { 0, 2, 8, 10 }, row[4] = { 0, 1, 4, 5 };
    return 16 * (col[k >> 2] + row[k & 3]);
}

/* Motion vectors: int16 (x,y) quarter-sample units per 4x4 block, RASTER order inside
 * the MB (index = x4 + 4*y4), one array per list: mv[list][mb*16 + blk][2].  Blocks that
 * do not use a list must hold (0,0) (what mv_cache holds, h264_slice.c:2031-2038). */

#define MI355_H264_MAX_REFS 16   /* per list (a field picture with more than 16 fields in a list is outside the path) */
#define MI355_H264_MAX_SLOTS 32  /* distinct reference pictures per frame */

/* Per-slice constants (sl->pwt, ref lists, PPS chroma QP tables). */
typedef struct mi355_h264_slice {
    uint8_t use_weight;               /* sl->pwt.use_weight: 0 none, 1 explicit, 2 implicit */
    uint8_t use_weight_chroma;
    uint8_t luma_log2_weight_denom;
    uint8_t chroma_log2_weight_denom;
    uint8_t list_count;               /* sl->list_count */
    uint8_t reserved[3];
    uint8_t ref_slot[2][MI355_H264_MAX_REFS];           /* ref_idx -> index into mi355_h264_frame.ref[];
                                                          also the picture identity the loop filter
                                                          compares (h->ref2frm, h264_slice.c:1828-1855) */
    int16_t luma_weight[MI355_H264_MAX_REFS][2][2];      /* [ref][list][{weight,offset}] */
    int16_t chroma_weight[MI355_H264_MAX_REFS][2][2][2]; /* [ref][list][cb/cr][{weight,offset}] */
    int16_t implicit_weight[MI355_H264_MAX_REFS][MI355_H264_MAX_REFS]; /* [ref0][ref1], frame MBs */
    uint8_t chroma_qp_table[2][52];   /* pps->chroma_qp_table[cb/cr][qp] (get_chroma_qp) */
    int16_t implicit_weight_field[2][2 * MI355_H264_MAX_REFS][2 * MI355_H264_MAX_REFS];
                                      /* MBAFF frames with use_weight 2, FIELD macroblocks: [mb_y & 1][ref0][ref1] with the field reference indices
                                         the macroblock codes (0 .. 2 * ref_count - 1) = the reference's implicit_weight[(16 + ref0) ^ p][(16 + ref1) ^ p][p],
                                         p = mb_y & 1 (h264_slice.c:623-682 with field 0 / 1; the index swap: h264_mb_template.c:78-91) */
} mi355_h264_slice;

/* Alignment contract of the picture surfaces (what the reference's own frame pool provides: av_frame_get_buffer
 * aligns planes and line sizes to >= 32 bytes): planes and strides of dst / recon / ref must be multiples of 4
 * bytes; when luma planes and strides are multiples of 16 and chroma ones of 8 the kernels move whole 16-/8-byte
 * row pieces (the fast path the benchmark runs).  coef, mv and mb arrays: 4-byte aligned. */
typedef struct mi355_h264_frame {
    int32_t mb_width, mb_height;
    uint8_t *dst[3];                  /* deblocked output picture */
    int32_t dst_stride[2];            /* luma, chroma */
    uint8_t *recon[3];                /* unfiltered reconstruction (intra prediction reads it) */
    int32_t recon_stride[2];
    const uint8_t *ref[MI355_H264_MAX_SLOTS][3]; /* reference pictures, same strides as dst */
    const mi355_h264_mb *mb;          /* [mb_width*mb_height], raster */
    const int16_t *mv[2];             /* see above; mv[1] may be NULL for P pictures */
    const int16_t *coef;              /* [nmb][384] */
    const mi355_h264_slice *slices;
    int32_t nslices;
    int32_t max_intra_level;          /* number of levels = return value of mi355_h264_intra_schedule() (0: no intra MBs) */
    /* intra schedule (device arrays; see mi355_h264_intra_schedule): the intra MBs of this
     * picture sorted by level; level L (1-based) occupies intra_list[start[L-1] .. start[L]) */
    const uint32_t *intra_list;
    const int32_t *intra_level_start; /* [max_intra_level + 1] */
    int32_t max_level_width;          /* largest number of macroblocks on one level (*max_level_width of
                                         mi355_h264_intra_schedule); 0 = unknown: mi355_h264_decode_frames() then assumes
                                         the widest level a picture of this size can have */
    int32_t field_picture;            /* 1: the picture is ONE FIELD of a frame (PAFF) — its planes are every other line of the frame's
                                         (pointers at the field's first line, strides doubled, mb_height = the field's), every reference
                                         slot is a field addressed the same way; the loop filter then uses the vertical vector limit 2
                                         (h264_loopfilter.c:723) and strength 3 instead of 4 on horizontal macroblock edges of intra
                                         macroblocks (:551-556).  0: a frame picture */
    int32_t surface_layout;           /* MI355_SURFACE_LINEAR (0): dst / recon / ref are planes with byte strides, as above.
                                         MI355_SURFACE_TILED (1): macroblock-tiled surfaces, the layout a decoded picture buffer
                                         keeps while it stays in HBM (see below); frame pictures only */
    int32_t flags;                    /* MI355_FRAME_NO_INTER (1): the picture holds no inter macroblock (an I picture): a wave of the inter pass
                                         leaves after reading this word instead of fetching its record and coefficients to find that out
                                         (960 bytes per macroblock; the launch of the picture's waves itself stays: 0.87 ms per 512 1080p
                                         pictures either way).  A hint: 0 is always right */
} mi355_h264_frame;
#define MI355_FRAME_NO_INTER 1
#define MI355_FRAME_MBAFF    2        /* mi355_h264_decode_frames_wide_dev only: an MBAFF frame (mb_adaptive_frame_field_flag) — macroblock rows 2k, 2k + 1
                                         are the top / bottom macroblocks of pair row k; a macroblock whose mb_type has MB_TYPE_INTERLACED (0x80) is a FIELD
                                         macroblock: its sixteen rows are every other line of the pair (top macroblock: the even lines), it predicts from the
                                         fields of its references (a slot per (frame, parity), the pointer at that field's first line; stride and height are
                                         taken doubled / halved) and its vectors count field lines (h264_mb_template.c:61-98, h264_mb.c:59-101) */

/* Macroblock-tiled surfaces (surface_layout == MI355_SURFACE_TILED).  What the reference keeps in an AVFrame with a line
 * stride (h264_mb.c:239-314 fetches a 21 x 21 window as 21 row pieces of 21 different cache lines, h264_mb_template.c:85-91
 * writes a macroblock as 32 row pieces) is stored here so that the unit every pass moves — a macroblock — is a run of whole
 * 128-byte lines:
 *   plane 0 (dst[0] / recon[0] / ref[s][0]): luma.  Macroblock (x, y) occupies the 256 bytes at y * stride[0] + x * 256:
 *            its 16 rows of 16 samples one after the other.
 *   plane 1 (dst[1] / recon[1] / ref[s][1]): chroma.  Macroblock (x, y) occupies the 128 bytes at y * stride[1] + x * 128:
 *            the 8 rows of 8 Cb samples, then the 8 rows of 8 Cr samples.
 *   plane 2 pointers are not read.
 * dst_stride / recon_stride = bytes per macroblock ROW of tiles (>= 256 * mb_width / 128 * mb_width, multiples of 128); the
 * reference surfaces use dst_stride, as in the linear layout.  Surfaces must be 128-byte aligned.
 * All three kinds of surface of a picture use the same layout; field pictures (field_picture != 0) must be linear.
 * mi355_h264_surface_convert_dev() moves pictures between the two layouts (upload of a reference decoded elsewhere; a
 * picture leaving HBM for display or for a consumer that wants lines). */
#define MI355_SURFACE_LINEAR 0
#define MI355_SURFACE_TILED  1
/* bit mask of the layouts a batch may hold (the *_layouts_dev entry points) */
#define MI355_LAYOUTS_LINEAR 1
#define MI355_LAYOUTS_TILED  2
#define MI355_TILE_LUMA_BYTES   256
#define MI355_TILE_CHROMA_BYTES 128

/* One picture to convert between the layouts.  `lin`: three planes with byte strides lin_stride[0] (luma), [1] (both chroma
 * planes); `tiled`: the two tiled planes with their macroblock-row strides.  Device pointers, or device-visible host memory
 * (mi355_host_alloc) on the linear side. */
typedef struct mi355_surface_job {
    uint8_t *lin[3];
    uint8_t *tiled[2];
    int32_t lin_stride[2];
    int32_t tiled_stride[2];
    int32_t mb_width, mb_height;
    int32_t to_tiled;                 /* 1: lin -> tiled, 0: tiled -> lin */
    int32_t reserved0;
} mi355_surface_job;
/* `n` conversions in ONE launch; `jobs` device-visible; max_mb_width / max_mb_height: the largest picture of the batch.
 * Linear planes and strides must be multiples of 8 bytes (luma 16 for the widest accesses; 4-byte multiples take a dword
 * path).  Returns 0, or -1 / -2 / -3 as mi355_h264_decode_frames. */
int mi355_h264_surface_convert_dev(const mi355_surface_job *jobs, int n, int max_mb_width, int max_mb_height, void *stream);

/* Reconstruct and deblock `nframes` independent pictures described by the HOST array
 * `frames` (its pointers are device pointers).  Work is enqueued on `stream`
 * (a hipStream_t; NULL = the null stream); the call returns without synchronising: the descriptors are copied into a
 * per-thread pinned buffer before it returns (the caller's array may be reused at once) and travel to a per-thread
 * device buffer on `stream`; both buffers are allocated once and grow on demand.  A thread that alternates between
 * streams must not have more than one call's descriptors in flight per stream pair — the second call waits for the
 * first one's descriptor copy, not for its kernels.
 * Returns 0; -1 invalid arguments / library not initialised; -2 launch failure; -3 geometry too large; -4 runtime error. */
int mi355_h264_decode_frames(const mi355_h264_frame *frames, int nframes, void *stream);

/* Same, with the descriptor array already resident on the device (`d_frames`), for callers
 * that keep everything in HBM; geometry limits must then be given explicitly. */
int mi355_h264_decode_frames_dev(const mi355_h264_frame *d_frames, int nframes,
                                 int max_mb_width, int max_mb_height, int max_intra_level,
                                 int max_level_width, void *stream);

/* Same, with the intra passes sized level by level: level_widths[l - 1] (HOST array, max_intra_level entries) = the
 * largest number of macroblocks any picture of the batch has on level l (level_start[l] - level_start[l - 1] as
 * mi355_h264_intra_schedule wrote them).  With the single bound above every level launches nframes * max_level_width
 * workgroups, most of which find nothing to do on the sparse later levels (0.5 ms per 2048 pictures of config 2). */
int mi355_h264_decode_frames_levels_dev(const mi355_h264_frame *d_frames, int nframes,
                                        int max_mb_width, int max_mb_height, int max_intra_level,
                                        const int32_t *level_widths, void *stream);

/* The same for a caller that knows which surface layouts its batch holds (MI355_LAYOUTS_*: sessions, bridges, the bench): the
 * reconstruction and the loop filter then run the kernel instances of those layouts alone (see the *_layouts_dev passes below). */
int mi355_h264_decode_frames_layouts_dev(const mi355_h264_frame *d_frames, int nframes,
                                         int max_mb_width, int max_mb_height, int max_intra_level,
                                         const int32_t *level_widths, int layouts, void *stream);

/* The same three passes for the formats outside the 8-bit kernels above: bit_depth 9 / 10 with chroma_format_idc 1 / 2 (High 10,
 * High 4:2:2 — h264_mb_template.c:27-58 PIXEL_SHIFT, :174-232 CHROMA422) and bit_depth 8 with chroma_format_idc 2.  What changes in
 * the descriptors of such a batch (every picture of it has the same format):
 *   - samples of dst / recon / ref are uint16_t when bit_depth > 8 (`pixel`, bit_depth_template.c:49-87); strides stay in bytes;
 *     surfaces are MI355_SURFACE_LINEAR; with chroma_format_idc 2 the chroma planes have the luma's height;
 *   - coef holds int32_t when bit_depth > 8 (`dctcoef`), int16_t otherwise: 256 luma + 2 * 64 chroma per macroblock, or 2 * 128 with
 *     chroma_format_idc 2 (Cb block j at [256 + 16 * j], Cr at [384 + 16 * j], j = 0..7 in the reference's order: sl->mb + 256 /
 *     + 512, h264idct_template.c:216-238), laid out as MI355_H264_COEFS_PER_MB describes otherwise; an I_PCM macroblock holds its
 *     samples one per coefficient slot (Y, Cb, Cr) — the host unpacks the bit_depth-wide fields of h264_mb_template.c:108-137;
 *   - mi355_h264_mb: qp / qpc are the table values (QpBdOffset included, as h->cur_pic.qscale_table holds them); dc_qmul[1..2] with
 *     chroma_format_idc 2 = dequant4_coeff[..][chroma_qp + 3][0] (h264_mb_template.c:232-236); the nnz_mask bits of chroma BLOCKS
 *     are not read (the kernels look at a block's coefficients: AC present, DC alone, nothing), the DC bits are.
 * passes: bit 0 inter, bit 1 intra, bit 2 loop filter (7 = everything; the separate bits are for measurement); bit 3 (8): the batch holds MBAFF
 * frames (MI355_FRAME_MBAFF in every descriptor) — the loop filter then works on macroblock pairs.
 * bit_depth 8 with chroma_format_idc 1 is taken too: the pictures the kernels above decode, through this kernel set (linear surfaces
 * only; how the two sets are tested against each other and against the oracle).
 * Returns 0, -1 (arguments / a format outside 8..10 bits, 4:2:0 / 4:2:2), -2, -3 as the others. */
int mi355_h264_decode_frames_wide_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                      int max_intra_level, const int32_t *level_widths, int bit_depth, int chroma_format_idc,
                                      int passes, void *stream);

/* Individual passes (same argument meaning), exposed for measurement and tests. */
int mi355_h264_recon_intra_levels_dev(const mi355_h264_frame *d_frames, int nframes, int max_intra_level, const int32_t *level_widths, void *stream);
/* The intra pass for a caller that names the batch's grid (as the decode_frames entry points do): every level of every picture in ONE launch — a macroblock
 * waits for its own intra neighbours (left, above-left, above, above-right: the type words of their records, one flag byte per macroblock in scratch memory of
 * the stream) instead of all pictures waiting for a level's slowest wave at a launch boundary.  The records must hold what mi355_h264_intra_schedule() saw
 * (mb_type) and the list its order.  MI355_INTRA_SINGLE=0 in the environment: the launch per level of mi355_h264_recon_intra_levels_dev. */
int mi355_h264_recon_intra_all_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, int max_intra_level, const int32_t *level_widths, void *stream);
int mi355_h264_recon_inter_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream);
/* The same for a caller that knows which surface layouts occur in the batch (MI355_LAYOUTS_*, below): a batch that is tiled throughout runs the
 * kernel instance that carries the tiled form of the macroblock code alone (fewer registers spilled, half the code); a picture of another layout
 * in such a launch is left untouched.  Any other mask = mi355_h264_recon_inter_dev. */
int mi355_h264_recon_inter_layouts_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, int layouts, void *stream);
/* The same pass for descriptors whose `coef` arrays live in device-visible host memory (mi355_host_alloc): an inter
 * macroblock fetches its coefficient block only when its record's cbp says it has coefficients (a second, dependent round
 * of loads for those; none for the others) — over PCIe the skipped blocks are what counts.  Same results. */
int mi355_h264_recon_inter_sparse_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream);
int mi355_h264_recon_intra_dev(const mi355_h264_frame *d_frames, int nframes, int max_intra_level, int max_level_width, void *stream);
/* The loop filter of the whole batch: bands of four macroblock rows, top to bottom.  The launch shape follows the batch: one
 * launch per band with a wave per picture when there are enough pictures to fill the device, two to six bands per launch
 * pipelined inside a workgroup (one wave each) when there are few — same pictures either way (the environment variable
 * MI355_DEBLOCK_FORM = 1 / 2 / 3 / 4 / 6 pins the number of bands per workgroup: a developer switch). */
int mi355_h264_deblock_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream);
/* The same for a caller that knows which surface layouts occur in the batch (the descriptors live on the device: the entry point
 * above cannot look, and launches the loop filter's kernel for each layout — the waves of the kernel whose layout a picture does
 * not have leave at once, ~1 us per 1000 macroblock-row bands): `layouts` = MI355_LAYOUTS_LINEAR | MI355_LAYOUTS_TILED, or one of them. */
int mi355_h264_deblock_layouts_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, int layouts, void *stream);

/* Host helper (plain CPU bookkeeping, no sample arithmetic): write the intra schedule of one picture:
 * `list` (capacity mb_width*mb_height) receives the intra MB indices sorted by level, `level_start`
 * (capacity mb_width + 2*mb_height + 1) the offsets.  Levels are computed in int (no 8-bit limit: any
 * picture size).  Returns the maximum level (<0: invalid arguments); *max_level_width receives the
 * largest number of MBs on one level. */
int mi355_h264_intra_schedule(mi355_h264_mb *mb, int mb_width, int mb_height,
                              uint32_t *list, int32_t *level_start, int *max_level_width);

/* The same for an MBAFF frame (MI355_FRAME_MBAFF: macroblock rows 2k, 2k + 1 are the pairs of pair row k; mb_height even): a macroblock waits for
 * both macroblocks of the left, above-left, above and above-right PAIRS, the bottom macroblock of a pair for the top one.
 * level_start needs 2 * (mb_width + 2 * mb_height) + 1 entries here (a pair adds two levels). */
int mi355_h264_intra_schedule_mbaff(mi355_h264_mb *mb, int mb_width, int mb_height,
                                    uint32_t *list, int32_t *level_start, int *max_level_width);

/* Device-memory plumbing for callers that do not link a HIP runtime themselves. */
void *mi355_malloc(size_t bytes);
void  mi355_free(void *dptr);
int   mi355_memcpy_h2d(void *dst, const void *src, size_t bytes);
int   mi355_memcpy_d2h(void *dst, const void *src, size_t bytes);
int   mi355_memcpy_d2d(void *dst, const void *src, size_t bytes);
int   mi355_sync(void *stream);
/* before the first mi355_init(): 1 = a thread that waits for the device sleeps instead of spinning (a host with more waiting threads than cores);
 * the environment variable MI355_BLOCKING_SYNC=0|1 overrides */
void  mi355_prefer_blocking_sync(int on);
/* pinned host memory and asynchronous copies on a stream: what a bridge needs to overlap the host's entropy decoding with
 * the device's reconstruction (contrib/libav/mi355_h264_bridge.c).  Host buffers of the async copies must come from
 * mi355_host_alloc(), or the copy is staged and effectively synchronous. */
void *mi355_host_alloc(size_t bytes);
void  mi355_host_free(void *p);
int   mi355_memcpy_h2d_async(void *dst, const void *src, size_t bytes, void *stream);
int   mi355_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream);
int   mi355_memcpy2d_d2h_async(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows, void *stream);
int   mi355_memcpy2d_d2d_async(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows, void *stream);
int   mi355_event_sync(void *event);
/* Memory from mi355_host_alloc() is DEVICE-VISIBLE at the same address: kernels may read records, vectors and
 * coefficients where the host wrote them (every input byte of a picture is read once, so nothing is gained by copying it
 * first), and mi355_copy_batch_dev() may write pictures into it.  `n` independent byte copies in ONE launch (one per
 * finished picture of a batch instead of one runtime copy call each); src / dst / bytes multiples of 16; `jobs` itself
 * device-visible.  max_bytes: the largest job. */
typedef struct mi355_copy_job {
    const void *src;
    void *dst;
    uint64_t bytes;
} mi355_copy_job;
int   mi355_copy_batch_dev(const mi355_copy_job *jobs, int n, size_t max_bytes, void *stream);
/* LARGE batches (hundreds of pictures and more; round 5, libav_amd/csrc/h264_pipelines.hip): the batch as `shares` shares, each through the three passes on a HIP stream
 * of its own, the shares' reconstruction launches taking turns (`turns` != 0) so that a reconstruction — bound by the memory pipeline — runs beside the other shares' loop
 * filters and intra passes — bound by the vector pipe — and never beside another reconstruction.  Three shares: 12.0 - 12.2 ms per 2048 1080p pictures against 13.4 - 13.7
 * for mi355_h264_decode_frames_layouts_dev on one stream.  The object keeps its streams and the turn across calls: consecutive batches run into each other; nothing is
 * finished before mi355_h264_pipelines_sync() (or _collect) returns.  One object per host thread and device.
 * _share(): which pictures of a batch of nframes share i takes (first, count): 2048 as 683 + 683 + 682.
 * _timing(1) + _collect(): sums[0..2] += milliseconds of the reconstruction / intra / loop-filter pass of every (share, call) since the last collect, measured by events on
 * the share's stream (the passes of different shares overlap: the sums exceed the wall time), *launches += their number. */
typedef struct mi355_h264_pipelines mi355_h264_pipelines;
mi355_h264_pipelines *mi355_h264_pipelines_create(int shares, int turns);      /* shares 1..16 */
void mi355_h264_pipelines_destroy(mi355_h264_pipelines *p);                      /* waits first */
int  mi355_h264_pipelines_decode_dev(mi355_h264_pipelines *p, const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                     int max_intra_level, const int32_t *level_widths, int layouts);
int  mi355_h264_pipelines_sync(mi355_h264_pipelines *p);
int  mi355_h264_pipelines_share(const mi355_h264_pipelines *p, int nframes, int share, int *first, int *count);
void mi355_h264_pipelines_timing(mi355_h264_pipelines *p, int on);
/* How consecutive calls are ordered.  Inside a call a share's three passes follow each other on its stream, and share i's pictures of the NEXT call follow them there.
 * A picture that references one decoded by ANOTHER share of the call before needs more: mode 1 (default) — a call whose batch differs from the one before (another
 * d_frames or nframes: pictures move between shares) first waits in every share for every share's loop filter of the call before; calls on the same batch are not
 * joined (a stream's pictures keep their share; they run into each other, which is where the schedule's gain at the call boundary comes from).  mode 2: every call is
 * joined.  mode 0: never (the caller orders what needs ordering).  A caller that moves streams between slots of an unchanged batch uses 2. */
void mi355_h264_pipelines_join(mi355_h264_pipelines *p, int mode);
int  mi355_h264_pipelines_collect(mi355_h264_pipelines *p, double sums[3], int *launches);

/* Streams for callers that pipeline half-batches (reconstruction of one against deblocking of the other);
 * `stream` arguments of every entry point accept these or NULL (the default stream). */
void *mi355_stream_create(void);
void  mi355_stream_destroy(void *stream);
int   mi355_stream_wait_event(void *stream, void *event);
/* HIP events on the caller's stream (kernel timing without a host sync per kernel) */
void *mi355_event_create(void);
void  mi355_event_destroy(void *event);
int   mi355_event_record(void *event, void *stream);
int   mi355_event_query(void *event);                  /* 1 finished, 0 not yet, -1 error; never waits */
float mi355_event_elapsed_ms(void *start, void *end);   /* waits for `end` */

#ifdef __cplusplus
}
#endif
#endif /* MI355_H264_FRAME_H */
