/* mi355_wrap_hevc.c — contrib/libav/mi355_wrap.c for a binary that links only the HEVC decoder and wraps ff_hevc_pred_init itself
 * (oracle/ref_hevc_tier1_main.c: the test harness keeps that one wrap for a pin of its own).  Product builds compile
 * mi355_wrap.c as it is. */
#define MI355_WRAP_NO_H264
#define MI355_WRAP_HEVC_NO_PRED
#include "mi355_wrap.c"
