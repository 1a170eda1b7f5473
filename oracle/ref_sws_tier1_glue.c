/*
 * ref_sws_tier1_glue.c — TEST INFRASTRUCTURE.  The reference's libswscale with the inner loops of its generic scaler
 * replaced, per call, by this project's Tier-1 swscale entry points: the replacing is done by the PRODUCT binding
 * contrib/libav/mi355_sws_glue.c (ff_sws_init_mi355x through --wrap=ff_getSwsFunc), which is compiled into
 * _ref/libswsref_tier1.so from there and linked against the emulated build of the product sources, so sws_scale() runs
 * through the product's kernels on a machine without a GPU.  This file only reports how many calls were forwarded.
 */
unsigned long mi355_sws_glue_calls(void);
unsigned long ref_sws_tier1_calls(void) { return mi355_sws_glue_calls(); }
unsigned long mi355_sws_glue_pictures(void);
unsigned long ref_sws_pictures(void) { return mi355_sws_glue_pictures(); }
