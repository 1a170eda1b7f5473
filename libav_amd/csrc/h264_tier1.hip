/*
 * h264_tier1.hip — Tier-1 entry points: the reference's H.264 DSP pointer tables
 * (H264DSPContext, H264QpelContext, H264ChromaContext, H264PredContext,
 * VideoDSPContext) served by HIP kernels, one synchronous launch per call.
 *
 * Host side = gather the touched sample window + parameters into the staging
 * arena, launch, scatter the written extent back, and reproduce the reference's
 * side effects on the coefficient block (cleared after use, h264idct_template.c:66,
 * :140, :150, :164).  All arithmetic runs in the kernels, which are thin wrappers
 * over the same wave-level building blocks (h264_dev.h) the batched frame
 * pipeline uses, so parity here is parity of the hot path's arithmetic.
 */
#include "mi355_rt.h"
#include "h264_dev.h"
#include "../../include/mi355dsp.h"

using namespace mi355;

/* the 9 / 10-bit instantiations live in h264_tier1_hbd.hip */
namespace mi355 {
void h264dsp_init_hbd(H264DSPContext *c, int bit_depth, int chroma_format_idc);
void h264qpel_init_hbd(H264QpelContext *c, int bit_depth);
void h264chroma_init_hbd(H264ChromaContext *c, int bit_depth);
void h264pred_init_hbd(H264PredContext *h, int bit_depth, int chroma_format_idc);
void videodsp_init_hbd(VideoDSPContext *ctx, int bpc);
}

#define LAUNCH1(kernel, a, ...) \
    hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, (a).stream, __VA_ARGS__)

/* ------------------------------------------------------------------------- */
/* qpel / chroma MC                                                            */
/* ------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64)
k_qpel(const uint8_t *win, int wpitch, uint8_t *dst, int dpitch, int size, int mx, int my, int avg)
{
    __shared__ McScratch s;
    __shared__ uint8_t pred[16 * 16];
    const int lane = lane_id();
    if (avg)
        for (int i = lane; i < size * size; i += 64) {
            int y = i / size, x = i - y * size;
            pred[y * 16 + x] = dst[y * dpitch + x];
        }
    __syncthreads();
    PlaneRef ref{win, wpitch, size + 5, size + 5};
    stage_windows(s, &ref, 2, 2, size, size, nullptr, nullptr, 0, 0, 0, 0);
    mc_luma_compute(s, mx, my, size, size, pred, 16, 0, 0, avg);
    for (int i = lane; i < size * size; i += 64) {
        int y = i / size, x = i - y * size;
        dst[y * dpitch + x] = pred[y * 16 + x];
    }
}

template <int SIZE, int POS, int AVG>
static void qpel_shim(uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    Arena &a = arena();
    constexpr int mx = POS & 3, my = POS >> 2;
    /* rows/cols the reference position actually reads (h264qpel_template.c:380-531):
     * copy the full (SIZE+5)^2 apron only where it exists for this position */
    const int x0 = mx ? -2 : 0, x1 = mx ? SIZE + 3 : SIZE;
    const int y0 = my ? -2 : 0, y1 = my ? SIZE + 3 : SIZE;
    Win w = win_pack(a, nullptr, 0, SIZE + 5, SIZE + 5, 0, 0); /* zero-filled */
    uint8_t *wp = a.h<uint8_t>(w.off);
    if ((mx & 1) && (my & 1)) {
        /* the diagonal quarter positions average one horizontally and one vertically filtered half sample
         * (h264qpel_template.c:429-481: mc11 / mc31 / mc13 / mc33): the reference reads a cross — SIZE rows with the
         * horizontal apron and SIZE columns with the vertical one — and never the four 2 x 2 / 2 x 3 corners */
        const int hrow = my == 3, vcol = mx == 3;          /* mc13 / mc33 filter the rows below, mc31 / mc33 the columns to the right */
        for (int y = hrow; y < hrow + SIZE; y++)
            std::memcpy(wp + (size_t)(y + 2) * w.pitch, src + y * stride - 2, (size_t)(SIZE + 5));
        for (int y = -2; y < SIZE + 3; y++)
            std::memcpy(wp + (size_t)(y + 2) * w.pitch + 2 + vcol, src + y * stride + vcol, (size_t)SIZE);
    } else
    for (int y = y0; y < y1; y++)
        std::memcpy(wp + (size_t)(y + 2) * w.pitch + (x0 + 2), src + y * stride + x0, (size_t)(x1 - x0));
    Win d = win_pack(a, dst, stride, SIZE, SIZE);
    a.upload();
    LAUNCH1(k_qpel, a, a.d<uint8_t>(w.off), w.pitch, a.d<uint8_t>(d.off), d.pitch, SIZE, mx, my, AVG);
    a.download();
    win_unpack(a, d, dst, stride, 0, 0, SIZE, SIZE);
}

__global__ void __launch_bounds__(64)
k_chroma(const uint8_t *win, int wpitch, uint8_t *dst, int dpitch, int w, int h, int fx, int fy, int avg)
{
    __shared__ McScratch s;
    __shared__ uint8_t pred[16 * 8];
    const int lane = lane_id();
    if (avg)
        for (int i = lane; i < w * h; i += 64) {
            int y = i / w, x = i - y * w;
            pred[y * 8 + x] = dst[y * dpitch + x];
        }
    __syncthreads();
    PlaneRef ref{win, wpitch, w + 1, h + 1};
    /* h can be 16 (4:2:2 callers); the wave handles it in two 8-row halves */
    for (int y0 = 0; y0 < h; y0 += 8) {
        int bh = h - y0 < 8 ? h - y0 : 8;
        stage_windows(s, nullptr, 0, 0, 0, 0, &ref, &ref, 0, y0, w, bh);
        mc_chroma_compute(s, 1, fx, fy, w, bh, pred, pred, 8, 0, y0, avg);
    }
    for (int i = lane; i < w * h; i += 64) {
        int y = i / w, x = i - y * w;
        dst[y * dpitch + x] = pred[y * 8 + x];
    }
}

template <int W, int AVG>
static void chroma_shim(uint8_t *dst, uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    Arena &a = arena();
    /* the reference never touches the extra column/row when its weight is zero */
    Win w = win_pack(a, src, stride, W + 1, h + 1, x ? W + 1 : W, y ? h + 1 : h);
    Win d = win_pack(a, dst, stride, W, h);
    a.upload();
    LAUNCH1(k_chroma, a, a.d<uint8_t>(w.off), w.pitch, a.d<uint8_t>(d.off), d.pitch, W, h, x, y, AVG);
    a.download();
    win_unpack(a, d, dst, stride, 0, 0, W, h);
}

/* ------------------------------------------------------------------------- */
/* inverse transforms                                                          */
/* ------------------------------------------------------------------------- */
/* up to 16 4x4 blocks per launch: mode[b] 0 = skip, 1 = dc only, 2 = full; block b
 * is added at (bx[b], by[b]) of the window; coefficients coef[b*16 ..] */
struct Idct4Job {
    int16_t coef[16 * 16];
    uint8_t mode[16], bx[16], by[16];
};
__global__ void __launch_bounds__(64)
k_idct4(uint8_t *win, int pitch, const Idct4Job *job)
{
    const int lane = lane_id(), b = lane >> 2, q = lane & 3;
    int c[4], r[4], row;
    for (int i = 0; i < 4; i++) c[i] = job->coef[b * 16 + q + 4 * i];
    const int mode = job->mode[b];
    idct4_quad(c, q, r, row);
    if (mode == 1) { /* h264idct_template.c:144-156 */
        int dc = (job->coef[b * 16] + 32) >> 6;
        r[0] = r[1] = r[2] = r[3] = dc;
    }
    if (mode)
        add_row4(win + (job->by[b] + row) * pitch + job->bx[b], r);
}

struct Idct8Job {
    int16_t coef[4 * 64];
    uint8_t mode[4], bx[4], by[4];
};
__global__ void __launch_bounds__(64)
k_idct8(uint8_t *win, int pitch, const Idct8Job *job)
{
    __shared__ int16_t blk[4 * 64];
    const int lane = lane_id(), b = (lane >> 3) & 3, i = lane & 7;
    for (int k = lane; k < 256; k += 64) blk[k] = job->coef[k];
    __syncthreads();
    const bool active = lane < 32;
    const int mode = job->mode[b];
    int r[8];
    idct8_lds(blk + b * 64, i, active, r);
    if (active && mode) {
        if (mode == 1) { /* h264idct_template.c:159-171 */
            int dc = (job->coef[b * 64] + 32) >> 6;
            for (int k = 0; k < 8; k++) r[k] = dc;
        }
        add_col(win + job->by[b] * pitch + job->bx[b] + i, pitch, r, 8);
    }
}

/* run up to 16 4x4 jobs against one destination plane window */
struct BlockReq {
    int off;        /* byte offset of the block from `dst` */
    int16_t *coef;  /* host block */
    int mode;
};
static void run_idct4(uint8_t *dst, int stride, const BlockReq *req, int n)
{
    if (!n) return;
    Arena &a = arena();
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
    int bx[16], by[16];
    for (int i = 0; i < n; i++) {
        /* offsets are 4*x + 4*y*stride with small x,y (h264_slice.c:485-494); recover x,y */
        int y = req[i].off >= 0 ? (req[i].off + stride / 2) / stride : -((-req[i].off + stride / 2) / stride);
        int x = req[i].off - y * stride;
        bx[i] = x; by[i] = y;
        if (x < minx) minx = x; if (y < miny) miny = y;
        if (x + 4 > maxx) maxx = x + 4; if (y + 4 > maxy) maxy = y + 4;
    }
    Win w = win_pack(a, dst + miny * (ptrdiff_t)stride + minx, stride, maxx - minx, maxy - miny);
    size_t joff = a.take(sizeof(Idct4Job));
    Idct4Job *job = a.h<Idct4Job>(joff);
    std::memset(job, 0, sizeof(*job));
    for (int i = 0; i < n; i++) {
        std::memcpy(job->coef + i * 16, req[i].coef, 32);
        job->mode[i] = (uint8_t)req[i].mode;
        job->bx[i] = (uint8_t)(bx[i] - minx);
        job->by[i] = (uint8_t)(by[i] - miny);
    }
    a.upload();
    LAUNCH1(k_idct4, a, a.d<uint8_t>(w.off), w.pitch, a.d<Idct4Job>(joff));
    a.download();
    for (int i = 0; i < n; i++) {
        if (!req[i].mode) continue;
        win_unpack(a, w, dst + by[i] * (ptrdiff_t)stride + bx[i], stride, bx[i] - minx, by[i] - miny, 4, 4);
        if (req[i].mode == 2) std::memset(req[i].coef, 0, 32);
        else req[i].coef[0] = 0;
    }
}
static void run_idct8(uint8_t *dst, int stride, const BlockReq *req, int n)
{
    if (!n) return;
    Arena &a = arena();
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
    int bx[4], by[4];
    for (int i = 0; i < n; i++) {
        int y = req[i].off >= 0 ? (req[i].off + stride / 2) / stride : -((-req[i].off + stride / 2) / stride);
        int x = req[i].off - y * stride;
        bx[i] = x; by[i] = y;
        if (x < minx) minx = x; if (y < miny) miny = y;
        if (x + 8 > maxx) maxx = x + 8; if (y + 8 > maxy) maxy = y + 8;
    }
    Win w = win_pack(a, dst + miny * (ptrdiff_t)stride + minx, stride, maxx - minx, maxy - miny);
    size_t joff = a.take(sizeof(Idct8Job));
    Idct8Job *job = a.h<Idct8Job>(joff);
    std::memset(job, 0, sizeof(*job));
    for (int i = 0; i < n; i++) {
        std::memcpy(job->coef + i * 64, req[i].coef, 128);
        job->mode[i] = (uint8_t)req[i].mode;
        job->bx[i] = (uint8_t)(bx[i] - minx);
        job->by[i] = (uint8_t)(by[i] - miny);
    }
    a.upload();
    LAUNCH1(k_idct8, a, a.d<uint8_t>(w.off), w.pitch, a.d<Idct8Job>(joff));
    a.download();
    for (int i = 0; i < n; i++) {
        if (!req[i].mode) continue;
        win_unpack(a, w, dst + by[i] * (ptrdiff_t)stride + bx[i], stride, bx[i] - minx, by[i] - miny, 8, 8);
        if (req[i].mode == 2) std::memset(req[i].coef, 0, 128);
        else req[i].coef[0] = 0;
    }
}

static int scan8(int i)
{
    int p = i >> 4, b = i & 15;
    int x = (b & 1) + 2 * ((b >> 2) & 1), y = ((b >> 1) & 1) + 2 * (b >> 3);
    return 4 + x + 8 * (1 + y + 5 * p);
}

static void t1_idct_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, block, 2}; run_idct4(dst, stride, &r, 1); }
static void t1_idct_dc_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, block, 1}; run_idct4(dst, stride, &r, 1); }
static void t1_idct8_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, block, 2}; run_idct8(dst, stride, &r, 1); }
static void t1_idct8_dc_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, block, 1}; run_idct8(dst, stride, &r, 1); }

/* dispatch rules of h264idct_template.c:174-214 */
static void t1_idct_add16(uint8_t *dst, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    BlockReq r[16]; int n = 0;
    for (int i = 0; i < 16; i++) {
        int nnz = nnzc[scan8(i)];
        if (nnz) r[n++] = BlockReq{off[i], block + i * 16, (nnz == 1 && block[i * 16]) ? 1 : 2};
    }
    run_idct4(dst, stride, r, n);
}
static void t1_idct_add16intra(uint8_t *dst, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    BlockReq r[16]; int n = 0;
    for (int i = 0; i < 16; i++) {
        if (nnzc[scan8(i)]) r[n++] = BlockReq{off[i], block + i * 16, 2};
        else if (block[i * 16]) r[n++] = BlockReq{off[i], block + i * 16, 1};
    }
    run_idct4(dst, stride, r, n);
}
static void t1_idct8_add4(uint8_t *dst, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    BlockReq r[4]; int n = 0;
    for (int i = 0; i < 16; i += 4) {
        int nnz = nnzc[scan8(i)];
        if (nnz) r[n++] = BlockReq{off[i], block + i * 16, (nnz == 1 && block[i * 16]) ? 1 : 2};
    }
    run_idct8(dst, stride, r, n);
}
static void t1_idct_add8(uint8_t **dest, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    for (int j = 1; j < 3; j++) {
        BlockReq r[4]; int n = 0;
        for (int i = j * 16; i < j * 16 + 4; i++) {
            if (nnzc[scan8(i)]) r[n++] = BlockReq{off[i], block + i * 16, 2};
            else if (block[i * 16]) r[n++] = BlockReq{off[i], block + i * 16, 1};
        }
        run_idct4(dest[j - 1], stride, r, n);
    }
}

/* ff_h264_idct_add8_422 h264idct_template.c:216-238: the second four blocks of a plane sit at block_offset[i + 4]
 * and are counted at scan8[i + 4] */
static void t1_idct_add8_422(uint8_t **dest, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    for (int j = 1; j < 3; j++) {
        BlockReq r[8]; int n = 0;
        for (int i = j * 16; i < j * 16 + 8; i++) {
            const int k = i < j * 16 + 4 ? i : i + 4;
            if (nnzc[scan8(k)]) r[n++] = BlockReq{off[k], block + i * 16, 2};
            else if (block[i * 16]) r[n++] = BlockReq{off[k], block + i * 16, 1};
        }
        run_idct4(dest[j - 1], stride, r, n);
    }
}

/* DC transforms */
__global__ void __launch_bounds__(64) k_luma_dc(int16_t *out, const int16_t *in, int qmul)
{
    if (lane_id() == 0) {
        int v[16], o[16];
        for (int i = 0; i < 16; i++) v[i] = in[i];
        luma_dc_dequant(v, qmul, o);
        for (int k = 0; k < 16; k++) out[k] = (int16_t)o[k];
    }
}
static void t1_luma_dc_dequant_idct(int16_t *output, int16_t *input, int qmul)
{
    Arena &a = arena();
    size_t in = a.take(32), out = a.take(32);
    std::memcpy(a.h<int16_t>(in), input, 32);
    a.upload();
    LAUNCH1(k_luma_dc, a, a.d<int16_t>(out), a.d<int16_t>(in), qmul);
    a.download();
    const int16_t *o = a.h<int16_t>(out);
    for (int k = 0; k < 16; k++) output[luma_dc_slot(k)] = o[k];
}
__global__ void __launch_bounds__(64) k_chroma_dc(int16_t *v, int qmul)
{
    if (lane_id() == 0) {
        int a = v[0], b = v[1], c = v[2], d = v[3];
        chroma_dc_dequant(a, b, c, d, qmul);
        v[0] = (int16_t)a; v[1] = (int16_t)b; v[2] = (int16_t)c; v[3] = (int16_t)d;
    }
}
static void t1_chroma_dc_dequant_idct(int16_t *block, int qmul)
{
    Arena &a = arena();
    size_t off = a.take(8);
    int16_t *h = a.h<int16_t>(off);
    for (int k = 0; k < 4; k++) h[k] = block[16 * k];
    a.upload();
    LAUNCH1(k_chroma_dc, a, a.d<int16_t>(off), qmul);
    a.download();
    for (int k = 0; k < 4; k++) block[16 * k] = h[k];
}
/* ff_h264_chroma422_dc_dequant_idct h264idct_template.c:277-303: 2x4 Hadamard of the eight DC levels
 * (block[32 * i + 16 * {0,1}]), (x * qmul + 128) >> 8, written back in place */
__global__ void __launch_bounds__(64) k_chroma422_dc(int16_t *v, int qmul)
{
    if (lane_id() == 0) {
        int t[8];
        for (int i = 0; i < 4; i++) { t[2 * i] = v[2 * i] + v[2 * i + 1]; t[2 * i + 1] = v[2 * i] - v[2 * i + 1]; }
        for (int i = 0; i < 2; i++) {
            const int z0 = t[i] + t[4 + i], z1 = t[i] - t[4 + i], z2 = t[2 + i] - t[6 + i], z3 = t[2 + i] + t[6 + i];
            v[0 + i] = (int16_t)(((z0 + z3) * qmul + 128) >> 8);
            v[2 + i] = (int16_t)(((z1 + z2) * qmul + 128) >> 8);
            v[4 + i] = (int16_t)(((z1 - z2) * qmul + 128) >> 8);
            v[6 + i] = (int16_t)(((z0 - z3) * qmul + 128) >> 8);
        }
    }
}
static void t1_chroma422_dc_dequant_idct(int16_t *block, int qmul)
{
    Arena &a = arena();
    size_t off = a.take(16);
    int16_t *h = a.h<int16_t>(off);
    for (int i = 0; i < 4; i++) { h[2 * i] = block[32 * i]; h[2 * i + 1] = block[32 * i + 16]; }
    a.upload();
    LAUNCH1(k_chroma422_dc, a, a.d<int16_t>(off), qmul);
    a.download();
    /* output k of row i lands at block[32 * i + {0, 16}] (x_offset[] = {0, 16}, stride 32) */
    for (int i = 0; i < 4; i++) { block[32 * i] = h[2 * i]; block[32 * i + 16] = h[2 * i + 1]; }
}

/* ------------------------------------------------------------------------- */
/* weighted prediction                                                         */
/* ------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64)
k_weight(uint8_t *p, int pitch, int w, int h, int ld, int wt, int off)
{
    weight_block(p, pitch, w, h, ld, wt, off);
}
__global__ void __launch_bounds__(64)
k_biweight(uint8_t *d, const uint8_t *s, int pitch, int w, int h, int ld, int wd, int ws, int off)
{
    biweight_block(d, s, pitch, w, h, ld, wd, ws, off);
}
template <int W>
static void weight_shim(uint8_t *block, int stride, int height, int log2_denom, int weight, int offset)
{
    Arena &a = arena();
    Win w = win_pack(a, block, stride, W, height);
    a.upload();
    LAUNCH1(k_weight, a, a.d<uint8_t>(w.off), w.pitch, W, height, log2_denom, weight, offset);
    a.download();
    win_unpack(a, w, block, stride, 0, 0, W, height);
}
template <int W>
static void biweight_shim(uint8_t *dst, uint8_t *src, int stride, int height, int log2_denom,
                          int weightd, int weights, int offset)
{
    Arena &a = arena();
    Win d = win_pack(a, dst, stride, W, height);
    Win s = win_pack(a, src, stride, W, height);
    a.upload();
    LAUNCH1(k_biweight, a, a.d<uint8_t>(d.off), a.d<uint8_t>(s.off), d.pitch, W, height, log2_denom, weightd, weights, offset);
    a.download();
    win_unpack(a, d, dst, stride, 0, 0, W, height);
}

/* ------------------------------------------------------------------------- */
/* deblocking edge filters                                                     */
/* ------------------------------------------------------------------------- */
/* window sample (across index k in [-R,R), line n) = win[(k+R)*xs + n*ys] */
struct LfJob {
    int xs, ys, nlines, inner, alpha, beta, kind; /* kind: 0 luma, 1 luma intra, 2 chroma, 3 chroma intra */
    int R;
    int8_t tc0[4];
};
__global__ void __launch_bounds__(64) k_loopfilter(uint8_t *win, const LfJob *jp)
{
    const LfJob j = *jp;
    const int n = lane_id();
    if (n >= j.nlines) return;
    uint8_t *c = win + j.R * j.xs + n * j.ys; /* q0 */
#define PX(k) c[(k) * j.xs]
    if (j.kind == 0) {
        int p2 = PX(-3), p1 = PX(-2), p0 = PX(-1), q0 = PX(0), q1 = PX(1), q2 = PX(2);
        lf_luma_line(p2, p1, p0, q0, q1, q2, j.alpha, j.beta, j.tc0[n / j.inner]);
        PX(-2) = (uint8_t)p1; PX(-1) = (uint8_t)p0; PX(0) = (uint8_t)q0; PX(1) = (uint8_t)q1;
    } else if (j.kind == 1) {
        int p3 = PX(-4), p2 = PX(-3), p1 = PX(-2), p0 = PX(-1), q0 = PX(0), q1 = PX(1), q2 = PX(2), q3 = PX(3);
        lf_luma_intra_line(p3, p2, p1, p0, q0, q1, q2, q3, j.alpha, j.beta);
        PX(-3) = (uint8_t)p2; PX(-2) = (uint8_t)p1; PX(-1) = (uint8_t)p0;
        PX(0) = (uint8_t)q0; PX(1) = (uint8_t)q1; PX(2) = (uint8_t)q2;
    } else {
        int p1 = PX(-2), p0 = PX(-1), q0 = PX(0), q1 = PX(1);
        if (j.kind == 2) lf_chroma_line(p1, p0, q0, q1, j.alpha, j.beta, j.tc0[n / j.inner]);
        else lf_chroma_intra_line(p1, p0, q0, q1, j.alpha, j.beta);
        PX(-1) = (uint8_t)p0; PX(0) = (uint8_t)q0;
    }
#undef PX
}
/* vertical_edge: samples across the edge are adjacent in memory ("h_loop_filter") */
static void lf_shim(uint8_t *pix, int stride, int alpha, int beta, const int8_t *tc0, int kind, int vertical_edge, int inner)
{
    Arena &a = arena();
    const int R = kind == 1 ? 4 : (kind == 0 ? 3 : 2), W = kind <= 1 ? 3 : 1;
    const int nlines = 4 * inner;
    Win w = vertical_edge ? win_pack(a, pix - R, stride, 2 * R, nlines)
                          : win_pack(a, pix - R * (ptrdiff_t)stride, stride, nlines, 2 * R);
    size_t joff = a.take(sizeof(LfJob));
    LfJob *j = a.h<LfJob>(joff);
    j->xs = vertical_edge ? 1 : w.pitch;
    j->ys = vertical_edge ? w.pitch : 1;
    j->nlines = nlines; j->inner = inner; j->alpha = alpha; j->beta = beta; j->kind = kind; j->R = R;
    for (int i = 0; i < 4; i++) j->tc0[i] = tc0 ? tc0[i] : 0;
    a.upload();
    LAUNCH1(k_loopfilter, a, a.d<uint8_t>(w.off), a.d<LfJob>(joff));
    a.download();
    if (vertical_edge) win_unpack(a, w, pix - W, stride, R - W, 0, 2 * W, nlines);
    else               win_unpack(a, w, pix - W * (ptrdiff_t)stride, stride, 0, R - W, nlines, 2 * W);
}
#define LF_TC(name, kind, vert, inner) \
    static void name(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0) { lf_shim(pix, stride, alpha, beta, tc0, kind, vert, inner); }
#define LF_IN(name, kind, vert, inner) \
    static void name(uint8_t *pix, int stride, int alpha, int beta) { lf_shim(pix, stride, alpha, beta, nullptr, kind, vert, inner); }
LF_TC(t1_v_lf_luma, 0, 0, 4) LF_TC(t1_h_lf_luma, 0, 1, 4) LF_TC(t1_h_lf_luma_mbaff, 0, 1, 2)
LF_IN(t1_v_lf_luma_intra, 1, 0, 4) LF_IN(t1_h_lf_luma_intra, 1, 1, 4) LF_IN(t1_h_lf_luma_mbaff_intra, 1, 1, 2)
LF_TC(t1_v_lf_chroma, 2, 0, 2) LF_TC(t1_h_lf_chroma, 2, 1, 2) LF_TC(t1_h_lf_chroma_mbaff, 2, 1, 1)
LF_IN(t1_v_lf_chroma_intra, 3, 0, 2) LF_IN(t1_h_lf_chroma_intra, 3, 1, 2) LF_IN(t1_h_lf_chroma_mbaff_intra, 3, 1, 1)
/* 4:2:2: the chroma edge of a macroblock is 16 lines high (h264dsp_template.c:276-283, :321-328) */
LF_TC(t1_h_lf_chroma422, 2, 1, 4) LF_TC(t1_h_lf_chroma422_mbaff, 2, 1, 2)
LF_IN(t1_h_lf_chroma422_intra, 3, 1, 4) LF_IN(t1_h_lf_chroma422_mbaff_intra, 3, 1, 2)

/* ---- a4: transform-bypass residual add, h264addpx_template.c:30-72: dst += residual without
 * clipping (wraps like the reference's pixel type), block cleared afterwards ---------------------- */
__global__ void __launch_bounds__(64) k_add_pixels(uint8_t *dst, int pitch, const int16_t *blk, int n)
{
    for (int i = lane_id(); i < n * n; i += 64) {
        const int y = i / n, x = i - y * n;
        dst[y * pitch + x] = (uint8_t)(dst[y * pitch + x] + blk[i]);
    }
}
template <int N> static void add_pixels_clear_shim(uint8_t *dst, int16_t *block, int stride)
{
    Arena &a = arena();
    Win w = win_pack(a, dst, stride, N, N);
    const size_t b = a.take(N * N * 2);
    std::memcpy(a.h<int16_t>(b), block, N * N * 2);
    a.upload();
    LAUNCH1(k_add_pixels, a, a.d<uint8_t>(w.off), w.pitch, a.d<const int16_t>(b), N);
    a.download();
    win_unpack(a, w, dst, stride, 0, 0, N, N);
    std::memset(block, 0, N * N * 2);
}

void ff_h264dsp_init_mi355x(H264DSPContext *c, const int bit_depth, const int chroma_format_idc)
{
    /* like an arch hook: only the variants this backend implements are overridden
     * (8-bit samples; 4:0:0, 4:2:0 and 4:2:2 — 4:4:4 chroma goes through the luma entries); everything else keeps
     * the C default; 9 and 10 bit: h264_tier1_hbd.hip) */
    if (bit_depth != 8) { h264dsp_init_hbd(c, bit_depth, chroma_format_idc); return; }
    c->weight_h264_pixels_tab[0] = weight_shim<16>;   c->weight_h264_pixels_tab[1] = weight_shim<8>;
    c->weight_h264_pixels_tab[2] = weight_shim<4>;    c->weight_h264_pixels_tab[3] = weight_shim<2>;
    c->biweight_h264_pixels_tab[0] = biweight_shim<16>; c->biweight_h264_pixels_tab[1] = biweight_shim<8>;
    c->biweight_h264_pixels_tab[2] = biweight_shim<4>;  c->biweight_h264_pixels_tab[3] = biweight_shim<2>;
    c->h264_v_loop_filter_luma = t1_v_lf_luma;
    c->h264_h_loop_filter_luma = t1_h_lf_luma;
    c->h264_h_loop_filter_luma_mbaff = t1_h_lf_luma_mbaff;
    c->h264_v_loop_filter_luma_intra = t1_v_lf_luma_intra;
    c->h264_h_loop_filter_luma_intra = t1_h_lf_luma_intra;
    c->h264_h_loop_filter_luma_mbaff_intra = t1_h_lf_luma_mbaff_intra;
    c->h264_v_loop_filter_chroma = t1_v_lf_chroma;
    c->h264_v_loop_filter_chroma_intra = t1_v_lf_chroma_intra;
    c->h264_idct_add = t1_idct_add;
    c->h264_idct8_add = t1_idct8_add;
    c->h264_idct_dc_add = t1_idct_dc_add;
    c->h264_idct8_dc_add = t1_idct8_dc_add;
    c->h264_idct_add16 = t1_idct_add16;
    c->h264_idct8_add4 = t1_idct8_add4;
    c->h264_idct_add16intra = t1_idct_add16intra;
    c->h264_luma_dc_dequant_idct = t1_luma_dc_dequant_idct;
    c->h264_add_pixels4_clear = add_pixels_clear_shim<4>;
    c->h264_add_pixels8_clear = add_pixels_clear_shim<8>;
    if (chroma_format_idc <= 1) {
        c->h264_h_loop_filter_chroma = t1_h_lf_chroma;
        c->h264_h_loop_filter_chroma_mbaff = t1_h_lf_chroma_mbaff;
        c->h264_h_loop_filter_chroma_intra = t1_h_lf_chroma_intra;
        c->h264_h_loop_filter_chroma_mbaff_intra = t1_h_lf_chroma_mbaff_intra;
        c->h264_idct_add8 = t1_idct_add8;
        c->h264_chroma_dc_dequant_idct = t1_chroma_dc_dequant_idct;
    } else {   /* 4:2:2, and 4:4:4 like the reference (h264dsp.c:80-130: every idc > 1 gets the 4:2:2 forms; 4:4:4 decoding does not call them) */
        c->h264_h_loop_filter_chroma = t1_h_lf_chroma422;
        c->h264_h_loop_filter_chroma_mbaff = t1_h_lf_chroma422_mbaff;
        c->h264_h_loop_filter_chroma_intra = t1_h_lf_chroma422_intra;
        c->h264_h_loop_filter_chroma_mbaff_intra = t1_h_lf_chroma422_mbaff_intra;
        c->h264_idct_add8 = t1_idct_add8_422;
        c->h264_chroma_dc_dequant_idct = t1_chroma422_dc_dequant_idct;
    }
}

void ff_h264qpel_init_mi355x(H264QpelContext *c, int bit_depth)
{
    if (bit_depth != 8) { h264qpel_init_hbd(c, bit_depth); return; }
#define QROW(tab, idx, SIZE, AVG) \
    c->tab[idx][0] = qpel_shim<SIZE, 0, AVG>;   c->tab[idx][1] = qpel_shim<SIZE, 1, AVG>;   \
    c->tab[idx][2] = qpel_shim<SIZE, 2, AVG>;   c->tab[idx][3] = qpel_shim<SIZE, 3, AVG>;   \
    c->tab[idx][4] = qpel_shim<SIZE, 4, AVG>;   c->tab[idx][5] = qpel_shim<SIZE, 5, AVG>;   \
    c->tab[idx][6] = qpel_shim<SIZE, 6, AVG>;   c->tab[idx][7] = qpel_shim<SIZE, 7, AVG>;   \
    c->tab[idx][8] = qpel_shim<SIZE, 8, AVG>;   c->tab[idx][9] = qpel_shim<SIZE, 9, AVG>;   \
    c->tab[idx][10] = qpel_shim<SIZE, 10, AVG>; c->tab[idx][11] = qpel_shim<SIZE, 11, AVG>; \
    c->tab[idx][12] = qpel_shim<SIZE, 12, AVG>; c->tab[idx][13] = qpel_shim<SIZE, 13, AVG>; \
    c->tab[idx][14] = qpel_shim<SIZE, 14, AVG>; c->tab[idx][15] = qpel_shim<SIZE, 15, AVG>;
    QROW(put_h264_qpel_pixels_tab, 0, 16, 0) QROW(put_h264_qpel_pixels_tab, 1, 8, 0)
    QROW(put_h264_qpel_pixels_tab, 2, 4, 0)  QROW(put_h264_qpel_pixels_tab, 3, 2, 0)
    QROW(avg_h264_qpel_pixels_tab, 0, 16, 1) QROW(avg_h264_qpel_pixels_tab, 1, 8, 1)
    QROW(avg_h264_qpel_pixels_tab, 2, 4, 1)
#undef QROW
}

void ff_h264chroma_init_mi355x(H264ChromaContext *c, int bit_depth)
{
    if (bit_depth != 8) { h264chroma_init_hbd(c, bit_depth); return; }
    c->put_h264_chroma_pixels_tab[0] = chroma_shim<8, 0>; c->put_h264_chroma_pixels_tab[1] = chroma_shim<4, 0>;
    c->put_h264_chroma_pixels_tab[2] = chroma_shim<2, 0>;
    c->avg_h264_chroma_pixels_tab[0] = chroma_shim<8, 1>; c->avg_h264_chroma_pixels_tab[1] = chroma_shim<4, 1>;
    c->avg_h264_chroma_pixels_tab[2] = chroma_shim<2, 1>;
}

/* ------------------------------------------------------------------------- */
/* intra prediction                                                            */
/* ------------------------------------------------------------------------- */
struct PredJob {
    uint16_t T[1 + 32], L[1 + 16];
    int kind, mode, has_tl, has_tr;
};
__global__ void __launch_bounds__(64) k_pred(const PredJob *jp, uint8_t *out, int pitch)
{
    __shared__ PredScratch s;
    const int lane = lane_id();
    if (lane < 33) s.T[lane] = jp->T[lane];
    if (lane < 17) s.L[lane] = jp->L[lane];
    __syncthreads();
    intra_pred_wave(s, jp->kind, jp->mode, jp->has_tl, jp->has_tr, out, pitch);
}

/* gather only the edge samples the reference reads for this (kind, mode, availability) */
static void pred_shim(uint8_t *src, ptrdiff_t stride, int kind, int mode, int has_tl, int has_tr, const uint8_t *topright)
{
    Arena &a = arena();
    const int N = kind == 0 ? 4 : (kind == 3 ? 16 : 8);          /* width */
    const int NH = kind == 4 ? 16 : N;                              /* height: 8x16 for 4:2:2 chroma */
    size_t joff = a.take(sizeof(PredJob));
    PredJob *j = a.h<PredJob>(joff);
    std::memset(j, 0, sizeof(*j));
    j->kind = kind; j->mode = mode; j->has_tl = has_tl; j->has_tr = has_tr;
    int top = 0, left = 0, corner = 0, tr = 0;
    if (kind <= 1) {
        int needs = pred_luma_needs(mode);
        top = needs & 1; left = (needs >> 1) & 1; corner = (needs >> 2) & 1; tr = (needs >> 3) & 1;
        if (kind == 1) {
            if ((top || left) && has_tl) corner = 1;
            if (top && has_tr) tr = 1;          /* t7's filter reads p[8,-1] */
            if (tr && !has_tr) tr = 0;
        }
    } else if (kind == 3) {
        top = mode == 0 || mode == 2 || mode == 3 || mode == 5;
        left = mode == 0 || mode == 1 || mode == 3 || mode == 4;
        corner = mode == 3;
    } else {
        top = mode == 0 || mode == 2 || mode == 3 || mode == 5 || mode == 7 || mode == 8;
        left = mode == 0 || mode == 1 || mode == 3 || mode == 4 || mode >= 7;
        corner = mode == 3;
    }
    if (top) for (int i = 0; i < N; i++) j->T[1 + i] = src[i - stride];
    if (left) for (int i = 0; i < NH; i++) j->L[1 + i] = src[-1 + i * stride];
    if (corner) j->T[0] = j->L[0] = src[-1 - stride];
    if (tr) {
        if (kind == 0) for (int i = 0; i < 4; i++) j->T[5 + i] = topright[i];
        else for (int i = 0; i < 8; i++) j->T[9 + i] = src[8 + i - stride];
    }
    size_t ooff = a.take((size_t)NH * 16);
    a.upload();
    LAUNCH1(k_pred, a, a.d<PredJob>(joff), a.d<uint8_t>(ooff), 16);
    a.download();
    const uint8_t *o = a.h<uint8_t>(ooff);
    for (int y = 0; y < NH; y++) std::memcpy(src + y * stride, o + y * 16, (size_t)N);
}
template <int M> static void p4_shim(uint8_t *s, const uint8_t *tr, ptrdiff_t st) { pred_shim(s, st, 0, M, 0, 1, tr); }
template <int M> static void p8l_shim(uint8_t *s, int tl, int tr, ptrdiff_t st) { pred_shim(s, st, 1, M, tl != 0, tr != 0, nullptr); }
template <int M> static void p8_shim(uint8_t *s, ptrdiff_t st) { pred_shim(s, st, 2, M, 0, 0, nullptr); }
template <int M> static void p16_shim(uint8_t *s, ptrdiff_t st) { pred_shim(s, st, 3, M, 0, 0, nullptr); }
template <int M> static void p8x16_shim(uint8_t *s, ptrdiff_t st) { pred_shim(s, st, 4, M, 0, 0, nullptr); }

/* ---- a10, lossless variants: prediction + residual as a running sum (h264pred_template.c:1127-1354) ----
 * Lane = (block, line); the sum starts at the neighbouring sample (or the (1,2,1)-filtered edge for the
 * 8x8 `filter_add` slots) and wraps at 8 bits at every step like the reference's pixel type.  Blocks that
 * feed each other (a block below / right of another one of the same call) run in rounds. */
struct PredAddJob {
    int16_t coef[16 * 16];
    uint8_t bx[16], by[16];          /* block origin inside the window */
    int32_t n, size, horizontal, filtered, has_tl, has_tr;
};
__global__ void __launch_bounds__(64) k_pred_add(uint8_t *win, int pitch, const PredAddJob *job)
{
    const int lane = lane_id(), size = job->size, b = lane / size, i = lane - b * size;
    const bool mine = b < job->n;
    const int bx = mine ? job->bx[b] : 0, by = mine ? job->by[b] : 0, hz = job->horizontal;
#define PX(x, y) win[(by + (y)) * pitch + bx + (x)]
    for (int round = 0; round < 4; round++) {
        /* blocks whose origin along the prediction direction is `round` blocks from the window's first block */
        const int along = hz ? bx - 1 : by - 1;      /* the window starts one sample before the first block */
        if (mine && (job->filtered || (along >> 2) == round) && (!job->filtered || round == 0)) {
            int v;
            if (!job->filtered) v = hz ? PX(-1, i) : PX(i, -1);
            else if (!hz) {      /* PREDICT_8x8_LOAD_TOP :857-862 */
                const int c = PX(i, -1);
                const int lft = i == 0 ? (job->has_tl ? PX(-1, -1) : c) : PX(i - 1, -1);
                const int rgt = i == 7 ? (job->has_tr ? PX(8, -1) : c) : PX(i + 1, -1);
                v = (lft + 2 * c + rgt + 2) >> 2;
            } else {             /* PREDICT_8x8_LOAD_LEFT :849-853 */
                const int c = PX(-1, i);
                const int up = i == 0 ? (job->has_tl ? PX(-1, -1) : c) : PX(-1, i - 1);
                v = i == 7 ? (PX(-1, 6) + 3 * c + 2) >> 2 : (up + 2 * c + PX(-1, i + 1) + 2) >> 2;
            }
            const int16_t *blk = job->coef + b * size * size;
            for (int k = 0; k < size; k++) {
                v = (v + (hz ? blk[i * size + k] : blk[k * size + i])) & 0xFF;
                if (hz) PX(k, i) = (uint8_t)v; else PX(i, k) = (uint8_t)v;
            }
        }
        __syncthreads();
    }
#undef PX
}
static void pred_add_run(uint8_t *pix, const int *offs, int nblk, int16_t *block, ptrdiff_t stride, int size, int horizontal,
                         int filtered, int has_tl, int has_tr)
{
    Arena &a = arena();
    int bx[16], by[16], minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
    for (int i = 0; i < nblk; i++) {
        const int off = offs ? offs[i] : 0;
        const int y = off >= 0 ? (off + (int)stride / 2) / (int)stride : -((-off + (int)stride / 2) / (int)stride);
        bx[i] = off - y * (int)stride; by[i] = y;
        if (bx[i] < minx) minx = bx[i];
        if (by[i] < miny) miny = by[i];
        if (bx[i] + size > maxx) maxx = bx[i] + size;
        if (by[i] + size > maxy) maxy = by[i] + size;
    }
    /* exactly the samples the reference reads: one line before the blocks along the prediction direction,
     * and for the filtered 8x8 forms the corner / the sample past the edge only when the flags say so */
    int x0 = minx, y0 = miny, x1 = maxx;
    if (horizontal) x0 -= 1; else y0 -= 1;
    if (filtered && !horizontal) { x0 -= has_tl ? 1 : 0; x1 += has_tr ? 1 : 0; }
    if (filtered && horizontal) y0 -= has_tl ? 1 : 0;
    /* the kernel addresses blocks relative to a window that starts one sample before them in both axes */
    const int wx0 = minx - 1, wy0 = miny - 1;
    Win w = win_pack(a, nullptr, 0, maxx + 1 - wx0, maxy - wy0, 0, 0);
    for (int y = y0; y < maxy; y++)
        std::memcpy(a.h<uint8_t>(w.off) + (size_t)(y - wy0) * w.pitch + (x0 - wx0), pix + y * stride + x0, (size_t)(x1 - x0));
    const size_t joff = a.take(sizeof(PredAddJob));
    PredAddJob *job = a.h<PredAddJob>(joff);
    std::memset(job, 0, sizeof(*job));
    std::memcpy(job->coef, block, sizeof(int16_t) * (size_t)nblk * size * size);
    for (int i = 0; i < nblk; i++) { job->bx[i] = (uint8_t)(bx[i] - wx0); job->by[i] = (uint8_t)(by[i] - wy0); }
    job->n = nblk; job->size = size; job->horizontal = horizontal; job->filtered = filtered; job->has_tl = has_tl; job->has_tr = has_tr;
    a.upload();
    LAUNCH1(k_pred_add, a, a.d<uint8_t>(w.off), w.pitch, a.d<const PredAddJob>(joff));
    a.download();
    for (int i = 0; i < nblk; i++)
        win_unpack(a, w, pix + by[i] * stride + bx[i], stride, bx[i] - wx0, by[i] - wy0, size, size);
    std::memset(block, 0, sizeof(int16_t) * (size_t)nblk * size * size);
}
template <int SIZE, int HZ> static void pred_add_shim(uint8_t *pix, int16_t *block, ptrdiff_t stride)
{
    pred_add_run(pix, nullptr, 1, block, stride, SIZE, HZ, 0, 0, 0);
}
template <int HZ> static void pred8x8l_filter_add_shim(uint8_t *pix, int16_t *block, int tl, int tr, ptrdiff_t stride)
{
    pred_add_run(pix, nullptr, 1, block, stride, 8, HZ, 1, tl != 0, tr != 0);
}
template <int NBLK, int HZ> static void pred_multi_add_shim(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    pred_add_run(pix, block_offset, NBLK, block, stride, 4, HZ, 0, 0, 0);
}
/* pred8x16_{vertical,horizontal}_add :1326-1354: blocks 0..3 at block_offset[0..3], 4..7 at block_offset[8..11] */
template <int HZ> static void pred8x16_add_shim(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    int offs[8];
    for (int i = 0; i < 4; i++) { offs[i] = block_offset[i]; offs[4 + i] = block_offset[8 + i]; }
    pred_add_run(pix, offs, 8, block, stride, 4, HZ, 0, 0, 0);
}

void ff_h264_pred_init_mi355x(H264PredContext *h, int codec_id, const int bit_depth, const int chroma_format_idc)
{
    if (codec_id != MI355_AV_CODEC_ID_H264) return;
    if (bit_depth != 8) { h264pred_init_hbd(h, bit_depth, chroma_format_idc); return; }   /* idc > 1: the 8x16 forms, as h264pred.c:470-565 selects them */
    h->pred4x4[0] = p4_shim<0>; h->pred4x4[1] = p4_shim<1>; h->pred4x4[2] = p4_shim<2>; h->pred4x4[3] = p4_shim<3>;
    h->pred4x4[4] = p4_shim<4>; h->pred4x4[5] = p4_shim<5>; h->pred4x4[6] = p4_shim<6>; h->pred4x4[7] = p4_shim<7>;
    h->pred4x4[8] = p4_shim<8>; h->pred4x4[9] = p4_shim<9>; h->pred4x4[10] = p4_shim<10>; h->pred4x4[11] = p4_shim<11>;
    h->pred8x8l[0] = p8l_shim<0>; h->pred8x8l[1] = p8l_shim<1>; h->pred8x8l[2] = p8l_shim<2>; h->pred8x8l[3] = p8l_shim<3>;
    h->pred8x8l[4] = p8l_shim<4>; h->pred8x8l[5] = p8l_shim<5>; h->pred8x8l[6] = p8l_shim<6>; h->pred8x8l[7] = p8l_shim<7>;
    h->pred8x8l[8] = p8l_shim<8>; h->pred8x8l[9] = p8l_shim<9>; h->pred8x8l[10] = p8l_shim<10>; h->pred8x8l[11] = p8l_shim<11>;
    if (chroma_format_idc <= 1) {
        h->pred8x8[0] = p8_shim<0>; h->pred8x8[1] = p8_shim<1>; h->pred8x8[2] = p8_shim<2>; h->pred8x8[3] = p8_shim<3>;
        h->pred8x8[4] = p8_shim<4>; h->pred8x8[5] = p8_shim<5>; h->pred8x8[6] = p8_shim<6>; h->pred8x8[7] = p8_shim<7>;
        h->pred8x8[8] = p8_shim<8>; h->pred8x8[9] = p8_shim<9>; h->pred8x8[10] = p8_shim<10>;
    } else {   /* 4:2:2: the same slots hold the 8x16 predictors (h264pred.c:470-531) */
        h->pred8x8[0] = p8x16_shim<0>; h->pred8x8[1] = p8x16_shim<1>; h->pred8x8[2] = p8x16_shim<2>; h->pred8x8[3] = p8x16_shim<3>;
        h->pred8x8[4] = p8x16_shim<4>; h->pred8x8[5] = p8x16_shim<5>; h->pred8x8[6] = p8x16_shim<6>; h->pred8x8[7] = p8x16_shim<7>;
        h->pred8x8[8] = p8x16_shim<8>; h->pred8x8[9] = p8x16_shim<9>; h->pred8x8[10] = p8x16_shim<10>;
    }
    h->pred16x16[0] = p16_shim<0>; h->pred16x16[1] = p16_shim<1>; h->pred16x16[2] = p16_shim<2>; h->pred16x16[3] = p16_shim<3>;
    h->pred16x16[4] = p16_shim<4>; h->pred16x16[5] = p16_shim<5>; h->pred16x16[6] = p16_shim<6>;
    /* lossless (transform bypass) forms: VERT_PRED 0 / HOR_PRED 1; VERT_PRED8x8 2 / HOR_PRED8x8 1 (h264pred.c:551-565) */
    h->pred4x4_add[0] = pred_add_shim<4, 0>;   h->pred4x4_add[1] = pred_add_shim<4, 1>;
    h->pred8x8l_add[0] = pred_add_shim<8, 0>;  h->pred8x8l_add[1] = pred_add_shim<8, 1>;
    h->pred8x8l_filter_add[0] = pred8x8l_filter_add_shim<0>; h->pred8x8l_filter_add[1] = pred8x8l_filter_add_shim<1>;
    if (chroma_format_idc <= 1) { h->pred8x8_add[2] = pred_multi_add_shim<4, 0>; h->pred8x8_add[1] = pred_multi_add_shim<4, 1>; }
    else { h->pred8x8_add[2] = pred8x16_add_shim<0>; h->pred8x8_add[1] = pred8x16_add_shim<1>; }
    h->pred16x16_add[2] = pred_multi_add_shim<16, 0>; h->pred16x16_add[1] = pred_multi_add_shim<16, 1>;
}

/* ------------------------------------------------------------------------- */
/* VideoDSPContext                                                             */
/* ------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64)
k_emu_edge(uint8_t *buf, int bpitch, const uint8_t *region, int rpitch, int rx0, int ry0,
           int bw, int bh, int sx, int sy, int w, int h)
{
    /* region holds plane samples [rx0..] x [ry0..]; every output reads the plane at
     * clamped coordinates (videodsp_template.c:24-96) */
    for (int i = lane_id(); i < bw * bh; i += 64) {
        int y = i / bw, x = i - y * bw;
        int cx = clip3(sx + x, 0, w - 1), cy = clip3(sy + y, 0, h - 1);
        buf[y * bpitch + x] = region[(cy - ry0) * rpitch + (cx - rx0)];
    }
}
static void t1_emulated_edge_mc(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize,
                                int block_w, int block_h, int src_x, int src_y, int w, int h)
{
    if (!w || !h) return;
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    /* the part of the plane the block can touch after clamping */
    const int rx0 = cl(src_x, 0, w - 1), rx1 = cl(src_x + block_w - 1, 0, w - 1);
    const int ry0 = cl(src_y, 0, h - 1), ry1 = cl(src_y + block_h - 1, 0, h - 1);
    const uint8_t *origin = src - src_y * src_linesize - src_x;
    for (int y0 = 0; y0 < block_h; y0 += 64) {        /* arena-sized strips; blocks are <= 71 rows */
        Arena &a = arena();
        const int bh = block_h - y0 < 64 ? block_h - y0 : 64;
        Win r = win_pack(a, origin + ry0 * src_linesize + rx0, src_linesize, rx1 - rx0 + 1, ry1 - ry0 + 1);
        Win o = win_pack(a, nullptr, 0, block_w, bh, 0, 0);
        a.upload();
        LAUNCH1(k_emu_edge, a, a.d<uint8_t>(o.off), o.pitch, a.d<uint8_t>(r.off), r.pitch, rx0, ry0,
                block_w, bh, src_x, src_y + y0, w, h);
        a.download();
        win_unpack(a, o, buf + y0 * buf_linesize, buf_linesize, 0, 0, block_w, bh);
    }
}

void ff_videodsp_init_mi355x(VideoDSPContext *ctx, int bpc)
{
    if (bpc > 8) { videodsp_init_hbd(ctx, bpc); return; }
    ctx->emulated_edge_mc = t1_emulated_edge_mc;
    /* prefetch stays the C no-op: a host cache hint has no device meaning */
}
