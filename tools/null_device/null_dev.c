/* A "device" that does nothing, for timing the HOST side of a binding on a CPU: every kernel entry point returns at once, device
 * memory is host memory, copies are memcpy.  The pictures that come out are garbage (arithmetic decoding does not depend on them);
 * what the clock sees is the decoder's parsing + the bridge's recording, level sorting, staging and copies — everything the GPU
 * cannot hide.  Developer tool (tools/host_side_time.sh); never part of libmi355dsp.so, never used by a test. */
#include <stdlib.h>
#include <string.h>
int mi355_init(int d) { (void)d; return 0; }
void *mi355_malloc(size_t n) { return malloc(n); }
void mi355_free(void *p) { free(p); }
void *mi355_host_alloc(size_t n) { return malloc(n); }
void mi355_host_free(void *p) { free(p); }
int mi355_memcpy_h2d(void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
int mi355_memcpy_d2h(void *d, const void *s, size_t n) { memcpy(d, s, n); return 0; }
int mi355_sync(void *s) { (void)s; return 0; }
#define NOP(name) int name() { return 0; }
NOP(mi355_edge_emu_batch_dev) NOP(mi355_hevc_boundary_strengths_dev) NOP(mi355_hevc_deblock_pictures_dev)
NOP(mi355_hevc_intra_pred_blocks_dev) NOP(mi355_hevc_intra_recon_blocks_dev) NOP(mi355_hevc_mcpred_batch_dev)
NOP(mi355_hevc_residual_batch_dev) NOP(mi355_hevc_sao_ctbs_dev)
