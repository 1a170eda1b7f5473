/* mi355_h264_session.h — whole-frame decoding sessions over the Tier-2 path.
 *
 * The reference's other plugin boundary is AVHWAccel (libavcodec/avcodec.h:3062-3086): per picture the decoder calls
 * start_frame(), decode_slice() once per slice and end_frame(), and the accelerator owns the picture surfaces
 * (call sites: start_frame h264_slice.c:1534, decode_slice h264dec.c:591, end_frame h264_picture.c:166).  This header is
 * that boundary for the batched engine, with ONE difference that follows from the scope of this library (the per-macroblock
 * DSP; entropy decoding stays on the host): decode_slice() takes the slice's PARSED macroblocks — Tier-2 records, vectors
 * and coefficients (mi355_h264_frame.h) — where AVHWAccel.decode_slice takes the slice's bitstream.  A session
 *   - owns the decoded picture buffer in HBM (`num_surfaces` surfaces; the caller names surfaces by index, as VAAPI /
 *     DXVA callers name theirs) plus the unfiltered reconstruction surfaces the intra pass reads,
 *   - copies the slice data it is handed into pinned staging memory (the caller's arrays may be reused when
 *     decode_slice() returns), computes the intra schedule, builds the descriptor and, at end_frame(), enqueues the
 *     copies and the Tier-2 passes on its own HIP stream — end_frame() does not wait for them; pictures complete in
 *     submission order, so a later picture may name any earlier one as a reference at once,
 *   - hands pictures back with get_frame() (waits for that picture, copies it into host planes) or as device pointers
 *     for consumers on the device (mi355_sws_scale_frames_dev: the f2 chain).
 * One session per stream of pictures; a session is used from one thread at a time.  Throughput across MANY streams comes
 * from giving the GPU the pictures of all of them at once — contrib/libav/mi355_h264_bridge.c does that for the reference
 * decoder (one dispatcher for all decoder threads); a session is the simple form: one picture per launch set.
 *
 * Return values: 0, or negative: -1 invalid argument / state, -2 launch or copy failure, -3 out of memory,
 * -4 the picture is incomplete at end_frame() (macroblocks no slice covered: the picture is dropped, nothing is launched). */
#ifndef MI355_H264_SESSION_H
#define MI355_H264_SESSION_H

#include "mi355_h264_frame.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355_h264_session mi355_h264_session;

typedef struct mi355_h264_session_params {
    int32_t mb_width, mb_height;      /* 8-bit 4:2:0 frames of 16 mb_width x 16 mb_height samples, decoded as frame pictures or as field pairs */
    int32_t num_surfaces;             /* decoded picture buffer size + 1 (the picture being decoded); 2 .. 64 */
    int32_t max_slices;               /* per picture; 0 = 64 */
    int32_t surface_layout;           /* MI355_SURFACE_LINEAR (0): the surfaces are planes with line strides.  MI355_SURFACE_TILED (1): the
                                         decoded picture buffer keeps its pictures macroblock-tiled (mi355_h264_frame.h) — the fast
                                         layout; frame pictures only (start_frame() with field != 0 fails), get_frame() / put_frame() /
                                         export_frame_dev() convert on the device */
    int32_t device;                   /* 0: the calling thread's device (mi355_set_device(), else mi355_init()'s); n > 0: device n - 1.  A
                                         session (and everything it owns) lives on one device; its entry points switch the calling thread to
                                         it for the duration of the call.  Sessions of a group must live on the group's device */
} mi355_h264_session_params;

typedef struct mi355_h264_picture_params {
    int32_t surface;                             /* receives the decoded picture (must not be one of its references) */
    int32_t nslots;                              /* entries of ref_surface in use */
    int32_t ref_surface[MI355_H264_MAX_SLOTS];   /* reference slot — what mi355_h264_slice.ref_slot[][] and the records name — -> surface */
    int32_t two_lists;                           /* the slices carry list-1 vectors (B picture) */
    int32_t field;                               /* 0: a frame picture; 1 / 2: the top / bottom FIELD of the frame in `surface` (PAFF; the session's
                                                    mb_height must be even): the slices' macroblock addresses count the field's macroblocks, the
                                                    records carry chroma_dy (mi355_h264_frame.h) */
    int32_t ref_parity[MI355_H264_MAX_SLOTS];    /* field pictures: 0 / 1 = the slot is the top / bottom field of ref_surface[slot] (the second
                                                    field of a frame may name the first field of its own surface) */
} mi355_h264_picture_params;

int  mi355_h264_session_open(mi355_h264_session **out, const mi355_h264_session_params *p);
void mi355_h264_session_close(mi355_h264_session *s);

/* AVHWAccel.start_frame: begin the picture that goes to pp->surface.  Waits only if both staging sets are still in flight. */
int mi355_h264_start_frame(mi355_h264_session *s, const mi355_h264_picture_params *pp);
/* AVHWAccel.decode_slice: `nmbs` macroblocks of one slice, in the order given by `mb_addr` (macroblock addresses in the
 * picture; NULL: first_mb, first_mb + 1, ...).  mb[i], mv0 + 32 i, mv1 + 32 i (NULL unless two_lists), coef + 384 i belong to
 * macroblock i of the call.  The records' slice_id is set by the session. */
int mi355_h264_decode_slice(mi355_h264_session *s, const mi355_h264_slice *hdr, int first_mb, int nmbs, const int32_t *mb_addr,
                            const mi355_h264_mb *mb, const int16_t *mv0, const int16_t *mv1, const int16_t *coef);
/* AVHWAccel.end_frame: enqueue reconstruction and loop filter of the picture; returns without waiting. */
int mi355_h264_end_frame(mi355_h264_session *s);

/* Wait for the picture in `surface` and copy it out (dst planes y, cb, cr with their strides in bytes). */
int mi355_h264_get_frame(mi355_h264_session *s, int surface, uint8_t *const dst[3], const int dst_stride[3]);
/* Load a picture into a surface from host planes (a reference decoded elsewhere: seeking, a stream joined in the middle).
 * Synchronous. */
int mi355_h264_put_frame(mi355_h264_session *s, int surface, const uint8_t *const src[3], const int src_stride[3]);
/* Device address of a surface's plane (valid while the session lives; contents defined once the picture's event has
 * passed: mi355_h264_surface_wait) and its stride — for consumers on the device.  A tiled session hands out its tile planes:
 * plane 0 = luma tiles, plane 1 = chroma tiles (plane 2 = plane 1), stride = bytes per macroblock row of tiles. */
const uint8_t *mi355_h264_surface_dev(mi355_h264_session *s, int surface, int plane, int *stride);
/* A surface as planes with line strides in DEVICE memory of the caller (dst planes y, cb, cr; strides multiples of 8, luma
 * 16 for the widest stores), for a consumer on the device that wants lines (mi355_sws_scale_frames_dev: the f2 chain).
 * Enqueued on `stream` (a hipStream_t; NULL = the session's stream) behind the picture's event; nothing waits.  One launch:
 * a tiled session converts, a linear one copies plane by plane. */
int mi355_h264_export_frame_dev(mi355_h264_session *s, int surface, uint8_t *const dst[3], const int dst_stride[3], void *stream);
int mi355_h264_surface_wait(mi355_h264_session *s, int surface);
/* the session's HIP stream (a hipStream_t): work enqueued on it after end_frame() runs after the picture */
void *mi355_h264_session_stream(mi355_h264_session *s);

/* ---- groups: the throughput form.  Sessions opened in a group share its HIP stream and do not launch at end_frame(): their
 * pictures wait until mi355_h264_group_flush() issues ONE launch set for all of them (one descriptor array, one pass of each
 * Tier-2 kernel over the pictures of all member sessions — picture sizes may differ).  At most one picture per session waits:
 * the next start_frame() of a session whose picture is still waiting flushes the group first, and so do get_frame(),
 * surface_wait(), put_frame() and close.  A group and its sessions are used from one thread at a time (hosts that feed from many
 * threads serialise around it, as contrib/libav/mi355_h264_bridge.c does with its dispatcher).  Close the sessions before
 * destroying the group. */
typedef struct mi355_h264_group mi355_h264_group;
int  mi355_h264_group_create(mi355_h264_group **out);             /* on the calling thread's device */
void mi355_h264_group_destroy(mi355_h264_group *g);
int  mi355_h264_session_open_grouped(mi355_h264_session **out, const mi355_h264_session_params *p, mi355_h264_group *g);
int  mi355_h264_group_flush(mi355_h264_group *g);

#ifdef __cplusplus
}
#endif
#endif
