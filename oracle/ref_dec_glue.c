/*
 * ref_dec_glue.c — symbols the reference decoder objects expect from libavcodec/bitstream_filters.c,
 * which we cannot compile (it includes a configure-generated list).  TEST INFRASTRUCTURE ONLY.
 * The H.264 decoder only ever asks for the pass-through "null" filter (libavcodec/decode.c:169).
 */
#include <string.h>
#include "libavcodec/avcodec.h"
extern const AVBitStreamFilter ff_null_bsf;
const AVBitStreamFilter *av_bsf_get_by_name(const char *name) { return name && !strcmp(name, "null") ? &ff_null_bsf : NULL; }
const AVClass *ff_bsf_child_class_next(const AVClass *prev) { (void)prev; return NULL; }
