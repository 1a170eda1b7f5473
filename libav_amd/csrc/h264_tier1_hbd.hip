/*
 * h264_tier1_hbd.hip — Tier-1 entry points for 9- and 10-bit H.264: the same pointer tables as h264_tier1.hip
 * (H264DSPContext, H264QpelContext, H264ChromaContext, H264PredContext, VideoDSPContext), filled for the
 * BIT_DEPTH 9 / 10 instantiations of the reference's templates (h264dsp.c:37-47, :57-135; h264qpel.c:47-100;
 * h264chroma.c:35-52; h264pred.c:408-565; videodsp.c:35-42): samples are 16-bit (`pixel` = uint16_t), coefficients
 * 32-bit (`dctcoef` = int32_t behind the tables' int16_t pointers), strides stay in bytes.
 *
 * One synchronous launch per call, like the 8-bit file: gather the touched window into the staging arena, launch,
 * scatter the written extent back, reproduce the reference's side effects on the coefficient block.  The arithmetic is
 * the integer formulation of the templates with the bit depth as a parameter; the loop-filter lines and the intra
 * predictors are the same device functions the 8-bit tables use (h264_dev.h), instantiated for the sample range.
 * These tables are the per-call (slow) boundary: what matters here is that a High 10 stream decodes through the
 * device bit-exactly, not the rate.
 */
#include "mi355_rt.h"
#include "h264_dev.h"
#include "../../include/mi355dsp.h"

using namespace mi355;

namespace {

typedef uint16_t px;

#define LAUNCH_HBD(kernel, a, ...) hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, (a).stream, __VA_ARGS__)

/* ---- motion compensation: h264qpel_template.c:77-300 (6-tap, 16 quarter positions), h264chroma_template.c:28-200 ---- */
__global__ void __launch_bounds__(64)
k_hbd_qpel(const px *win, int wp, px *dst, int dp, int size, int mx, int my, int avg, int maxv)
{
    /* window sample (x, y) of the block sits at win[(y + 2) * wp + x + 2] */
#define S(x, y) ((int)win[((y) + 2) * wp + (x) + 2])
    for (int i = lane_id(); i < size * size; i += 64) {
        const int y = i / size, x = i - y * size;
        auto rawh = [&](int xx, int yy) { return tap6(S(xx - 2, yy), S(xx - 1, yy), S(xx, yy), S(xx + 1, yy), S(xx + 2, yy), S(xx + 3, yy)); };
        auto hh = [&](int xx, int yy) { return clip3((rawh(xx, yy) + 16) >> 5, 0, maxv); };
        auto vv = [&](int xx, int yy) { return clip3((tap6(S(xx, yy - 2), S(xx, yy - 1), S(xx, yy), S(xx, yy + 1), S(xx, yy + 2), S(xx, yy + 3)) + 16) >> 5, 0, maxv); };
        /* the reference keeps the first pass of the 2-D positions in int16_t, biased by `pad` at 10 bit (h264qpel_template.c:119-146):
         * samples inside the bit depth's range fit, samples outside it (planes of transform-bypass streams, whose residual adds do
         * not clip) wrap — and so does this */
        const int pad = maxv > 511 ? -10 * maxv : 0;
        auto tmph = [&](int xx, int yy) { return (int)(int16_t)(rawh(xx, yy) + pad) - pad; };
        auto hv = [&](int xx, int yy) {
            return clip3((tap6(tmph(xx, yy - 2), tmph(xx, yy - 1), tmph(xx, yy), tmph(xx, yy + 1), tmph(xx, yy + 2), tmph(xx, yy + 3)) + 512) >> 10, 0, maxv);
        };
        int v;
        if (my == 0) v = mx == 0 ? S(x, y) : (mx == 2 ? hh(x, y) : f2(S(x + (mx == 3), y), hh(x, y)));
        else if (mx == 0) v = my == 2 ? vv(x, y) : f2(S(x, y + (my == 3)), vv(x, y));
        else if (mx == 2 && my == 2) v = hv(x, y);
        else if (mx == 2) v = f2(hh(x, y + (my == 3)), hv(x, y));
        else if (my == 2) v = f2(vv(x + (mx == 3), y), hv(x, y));
        else v = f2(hh(x, y + (my == 3)), vv(x + (mx == 3), y));
        dst[y * dp + x] = (px)(avg ? f2(dst[y * dp + x], v) : v);
    }
#undef S
}

template <int SIZE, int POS, int AVG, int BD>
void qpel_shim(uint8_t *dst, const uint8_t *src, ptrdiff_t stride)
{
    Arena &a = arena();
    constexpr int mx = POS & 3, my = POS >> 2;
    /* rows / columns the reference position reads (h264qpel_template.c:380-531), as in the 8-bit shim */
    const int x0 = mx ? -2 : 0, x1 = mx ? SIZE + 3 : SIZE;
    const int y0 = my ? -2 : 0, y1 = my ? SIZE + 3 : SIZE;
    Win w = win_pack(a, nullptr, 0, (SIZE + 5) * 2, SIZE + 5, 0, 0);       /* zero-filled */
    uint8_t *wp = a.h<uint8_t>(w.off);
    if ((mx & 1) && (my & 1)) {
        const int hrow = my == 3, vcol = mx == 3;
        for (int y = hrow; y < hrow + SIZE; y++)
            std::memcpy(wp + (size_t)(y + 2) * w.pitch, src + y * stride - 4, (size_t)(SIZE + 5) * 2);
        for (int y = -2; y < SIZE + 3; y++)
            std::memcpy(wp + (size_t)(y + 2) * w.pitch + 2 * (2 + vcol), src + y * stride + 2 * vcol, (size_t)SIZE * 2);
    } else
    for (int y = y0; y < y1; y++)
        std::memcpy(wp + (size_t)(y + 2) * w.pitch + 2 * (x0 + 2), src + y * stride + 2 * x0, (size_t)(x1 - x0) * 2);
    Win d = win_pack(a, dst, stride, SIZE * 2, SIZE);
    a.upload();
    LAUNCH_HBD(k_hbd_qpel, a, a.d<px>(w.off), w.pitch / 2, a.d<px>(d.off), d.pitch / 2, SIZE, mx, my, AVG, (1 << BD) - 1);
    a.download();
    win_unpack(a, d, dst, stride, 0, 0, SIZE * 2, SIZE);
}

__global__ void __launch_bounds__(64)
k_hbd_chroma(const px *win, int wp, px *dst, int dp, int w, int h, int fx, int fy, int avg)
{
    const int A = (8 - fx) * (8 - fy), B = fx * (8 - fy), C = (8 - fx) * fy, D = fx * fy;
    for (int i = lane_id(); i < w * h; i += 64) {
        const int y = i / w, x = i - y * w;
        const int v = (A * win[y * wp + x] + B * win[y * wp + x + 1] + C * win[(y + 1) * wp + x] + D * win[(y + 1) * wp + x + 1] + 32) >> 6;
        dst[y * dp + x] = (px)(avg ? f2(dst[y * dp + x], v) : v);
    }
}
template <int W, int AVG>
void chroma_shim(uint8_t *dst, uint8_t *src, ptrdiff_t stride, int h, int x, int y)
{
    Arena &a = arena();
    /* the reference never touches the extra column / row when its weight is zero */
    Win w = win_pack(a, src, stride, (W + 1) * 2, h + 1, (x ? W + 1 : W) * 2, y ? h + 1 : h);
    Win d = win_pack(a, dst, stride, W * 2, h);
    a.upload();
    LAUNCH_HBD(k_hbd_chroma, a, a.d<px>(w.off), w.pitch / 2, a.d<px>(d.off), d.pitch / 2, W, h, x, y, AVG);
    a.download();
    win_unpack(a, d, dst, stride, 0, 0, W * 2, h);
}

/* ---- weighted prediction: h264dsp_template.c:30-98 ------------------------------------------------------------------- */
__global__ void __launch_bounds__(64)
k_hbd_weight(px *p, int pitch, int w, int h, int ld, int wt, int off, int bd)
{
    int o = (int)((unsigned)off << (ld + (bd - 8)));
    if (ld) o += 1 << (ld - 1);
    for (int i = lane_id(); i < w * h; i += 64) {
        const int y = i / w, x = i - y * w;
        p[y * pitch + x] = (px)clip3((p[y * pitch + x] * wt + o) >> ld, 0, (1 << bd) - 1);
    }
}
__global__ void __launch_bounds__(64)
k_hbd_biweight(px *d, const px *s, int pitch, int w, int h, int ld, int wd, int ws, int off, int bd)
{
    const int o = (int)((unsigned)((((int)((unsigned)off << (bd - 8))) + 1) | 1) << ld);
    for (int i = lane_id(); i < w * h; i += 64) {
        const int y = i / w, x = i - y * w;
        d[y * pitch + x] = (px)clip3((s[y * pitch + x] * ws + d[y * pitch + x] * wd + o) >> (ld + 1), 0, (1 << bd) - 1);
    }
}
template <int W, int BD>
void weight_shim(uint8_t *block, int stride, int height, int log2_denom, int weight, int offset)
{
    Arena &a = arena();
    Win w = win_pack(a, block, stride, W * 2, height);
    a.upload();
    LAUNCH_HBD(k_hbd_weight, a, a.d<px>(w.off), w.pitch / 2, W, height, log2_denom, weight, offset, BD);
    a.download();
    win_unpack(a, w, block, stride, 0, 0, W * 2, height);
}
template <int W, int BD>
void biweight_shim(uint8_t *dst, uint8_t *src, int stride, int height, int log2_denom, int weightd, int weights, int offset)
{
    Arena &a = arena();
    Win d = win_pack(a, dst, stride, W * 2, height);
    Win s = win_pack(a, src, stride, W * 2, height);
    a.upload();
    LAUNCH_HBD(k_hbd_biweight, a, a.d<px>(d.off), a.d<px>(s.off), d.pitch / 2, W, height, log2_denom, weightd, weights, offset, BD);
    a.download();
    win_unpack(a, d, dst, stride, 0, 0, W * 2, height);
}

/* ---- inverse transforms: h264idct_template.c:33-172 with 32-bit coefficients -------------------------------------------- */
struct IdctJob {                /* up to 16 4x4 or 4 8x8 blocks of one plane window */
    int32_t coef[16 * 16];
    uint8_t mode[16], bx[16], by[16];      /* 0 skip, 1 dc only, 2 full; block origin in samples */
    int32_t n, size;
};
__device__ inline void hbd_idct4(const int32_t *c, int r[16])
{
    int t[16];
    for (int i = 0; i < 4; i++) {
        const int c0 = c[i] + (i == 0 ? 32 : 0);
        const int z0 = c0 + c[i + 8], z1 = c0 - c[i + 8], z2 = (c[i + 4] >> 1) - c[i + 12], z3 = c[i + 4] + (c[i + 12] >> 1);
        t[i] = z0 + z3; t[i + 4] = z1 + z2; t[i + 8] = z1 - z2; t[i + 12] = z0 - z3;
    }
    for (int i = 0; i < 4; i++) {
        const int z0 = t[4 * i] + t[4 * i + 2], z1 = t[4 * i] - t[4 * i + 2], z2 = (t[4 * i + 1] >> 1) - t[4 * i + 3], z3 = t[4 * i + 1] + (t[4 * i + 3] >> 1);
        /* residual of column i, rows 0..3 */
        r[i] = (z0 + z3) >> 6; r[4 + i] = (z1 + z2) >> 6; r[8 + i] = (z1 - z2) >> 6; r[12 + i] = (z0 - z3) >> 6;
    }
}
__global__ void __launch_bounds__(64) k_hbd_idct(px *win, int pitch, const IdctJob *job, int maxv)
{
    const int b = lane_id();
    if (b >= job->n || !job->mode[b]) return;
    px *d = win + job->by[b] * pitch + job->bx[b];
    if (job->size == 4) {
        int r[16];
        if (job->mode[b] == 1) { const int dc = (job->coef[b * 16] + 32) >> 6; for (int k = 0; k < 16; k++) r[k] = dc; }
        else hbd_idct4(job->coef + b * 16, r);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) d[y * pitch + x] = (px)clip3(d[y * pitch + x] + r[4 * y + x], 0, maxv);
        return;
    }
    const int32_t *c = job->coef + b * 64;
    if (job->mode[b] == 1) {
        const int dc = (c[0] + 32) >> 6;
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) d[y * pitch + x] = (px)clip3(d[y * pitch + x] + dc, 0, maxv);
        return;
    }
    int t[64];
    for (int i = 0; i < 8; i++) {          /* first pass over block[i + 8 * k], second over block[k + 8 * i] (:84-134) */
        int in[8], out[8];
        for (int k = 0; k < 8; k++) in[k] = c[i + 8 * k] + ((i == 0 && k == 0) ? 32 : 0);
        idct8_1d(in, out);
        for (int k = 0; k < 8; k++) t[i + 8 * k] = out[k];
    }
    for (int i = 0; i < 8; i++) {
        int in[8], out[8];
        for (int k = 0; k < 8; k++) in[k] = t[k + 8 * i];
        idct8_1d(in, out);
        for (int k = 0; k < 8; k++) d[k * pitch + i] = (px)clip3(d[k * pitch + i] + (out[k] >> 6), 0, maxv);
    }
}
struct BlockReq {
    int off;            /* byte offset of the block from `dst` */
    int32_t *coef;      /* the host block, as the 32-bit coefficients it holds */
    int mode;
};
template <int BD>
void run_idct(uint8_t *dst, int stride, const BlockReq *req, int n, int size)
{
    if (!n) return;
    Arena &a = arena();
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
    int bx[16], by[16];
    for (int i = 0; i < n; i++) {
        /* offsets are 2 * 4 * x + 4 * y * stride bytes with small x, y (h264_slice.c:485-494): recover samples and rows */
        const int y = req[i].off >= 0 ? (req[i].off + stride / 2) / stride : -((-req[i].off + stride / 2) / stride);
        const int x = (req[i].off - y * stride) / 2;
        bx[i] = x; by[i] = y;
        if (x < minx) minx = x;
        if (y < miny) miny = y;
        if (x + size > maxx) maxx = x + size;
        if (y + size > maxy) maxy = y + size;
    }
    Win w = win_pack(a, dst + miny * (ptrdiff_t)stride + 2 * minx, stride, (maxx - minx) * 2, maxy - miny);
    const size_t joff = a.take(sizeof(IdctJob));
    IdctJob *job = a.h<IdctJob>(joff);
    std::memset(job, 0, sizeof(*job));
    job->n = n; job->size = size;
    for (int i = 0; i < n; i++) {
        std::memcpy(job->coef + i * size * size, req[i].coef, (size_t)size * size * 4);
        job->mode[i] = (uint8_t)req[i].mode;
        job->bx[i] = (uint8_t)(bx[i] - minx);
        job->by[i] = (uint8_t)(by[i] - miny);
    }
    a.upload();
    LAUNCH_HBD(k_hbd_idct, a, a.d<px>(w.off), w.pitch / 2, a.d<IdctJob>(joff), (1 << BD) - 1);
    a.download();
    for (int i = 0; i < n; i++) {
        if (!req[i].mode) continue;
        win_unpack(a, w, dst + by[i] * (ptrdiff_t)stride + 2 * bx[i], stride, (bx[i] - minx) * 2, by[i] - miny, size * 2, size);
        if (req[i].mode == 2) std::memset(req[i].coef, 0, (size_t)size * size * 4);
        else req[i].coef[0] = 0;
    }
}
int scan8(int i)
{
    const int p = i >> 4, b = i & 15;
    const int x = (b & 1) + 2 * ((b >> 2) & 1), y = ((b >> 1) & 1) + 2 * (b >> 3);
    return 4 + x + 8 * (1 + y + 5 * p);
}
inline int32_t *co(int16_t *block, int i) { return reinterpret_cast<int32_t *>(block) + i * 16; }      /* block i of a macroblock */

template <int BD> void t_idct_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, co(block, 0), 2}; run_idct<BD>(dst, stride, &r, 1, 4); }
template <int BD> void t_idct_dc_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, co(block, 0), 1}; run_idct<BD>(dst, stride, &r, 1, 4); }
template <int BD> void t_idct8_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, co(block, 0), 2}; run_idct<BD>(dst, stride, &r, 1, 8); }
template <int BD> void t_idct8_dc_add(uint8_t *dst, int16_t *block, int stride) { BlockReq r{0, co(block, 0), 1}; run_idct<BD>(dst, stride, &r, 1, 8); }
/* dispatch rules of h264idct_template.c:174-238 */
template <int BD> void t_idct_add16(uint8_t *dst, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    BlockReq r[16]; int n = 0;
    for (int i = 0; i < 16; i++) {
        const int nnz = nnzc[scan8(i)];
        if (nnz) r[n++] = BlockReq{off[i], co(block, i), (nnz == 1 && co(block, i)[0]) ? 1 : 2};
    }
    run_idct<BD>(dst, stride, r, n, 4);
}
template <int BD> void t_idct_add16intra(uint8_t *dst, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    BlockReq r[16]; int n = 0;
    for (int i = 0; i < 16; i++) {
        if (nnzc[scan8(i)]) r[n++] = BlockReq{off[i], co(block, i), 2};
        else if (co(block, i)[0]) r[n++] = BlockReq{off[i], co(block, i), 1};
    }
    run_idct<BD>(dst, stride, r, n, 4);
}
template <int BD> void t_idct8_add4(uint8_t *dst, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    BlockReq r[4]; int n = 0;
    for (int i = 0; i < 16; i += 4) {
        const int nnz = nnzc[scan8(i)];
        if (nnz) r[n++] = BlockReq{off[i], co(block, i), (nnz == 1 && co(block, i)[0]) ? 1 : 2};
    }
    run_idct<BD>(dst, stride, r, n, 8);
}
template <int BD> void t_idct_add8(uint8_t **dest, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    for (int j = 1; j < 3; j++) {
        BlockReq r[4]; int n = 0;
        for (int i = j * 16; i < j * 16 + 4; i++) {
            if (nnzc[scan8(i)]) r[n++] = BlockReq{off[i], co(block, i), 2};
            else if (co(block, i)[0]) r[n++] = BlockReq{off[i], co(block, i), 1};
        }
        run_idct<BD>(dest[j - 1], stride, r, n, 4);
    }
}
template <int BD> void t_idct_add8_422(uint8_t **dest, const int *off, int16_t *block, int stride, const uint8_t nnzc[15 * 8])
{
    for (int j = 1; j < 3; j++) {
        BlockReq r[8]; int n = 0;
        for (int i = j * 16; i < j * 16 + 8; i++) {
            const int k = i < j * 16 + 4 ? i : i + 4;
            if (nnzc[scan8(k)]) r[n++] = BlockReq{off[k], co(block, i), 2};
            else if (co(block, i)[0]) r[n++] = BlockReq{off[k], co(block, i), 1};
        }
        run_idct<BD>(dest[j - 1], stride, r, n, 4);
    }
}

/* DC transforms :240-310: 32-bit in and out, element positions as in the 8-bit tables */
__global__ void __launch_bounds__(64) k_hbd_dc(int32_t *v, int qmul, int kind)
{
    if (lane_id() != 0) return;
    if (kind == 0) {                    /* luma: 16 values in, luma_dc_dequant order out */
        /* the butterflies of luma_dc_dequant (h264_dev.h) without its 16-bit store: dctcoef is 32 bits wide here */
        int t[16];
        for (int i = 0; i < 4; i++) {
            const int s = v[4 * i] + v[4 * i + 1], d = v[4 * i] - v[4 * i + 1];
            const int e = v[4 * i + 2] - v[4 * i + 3], u = v[4 * i + 2] + v[4 * i + 3];
            t[4 * i] = s + u; t[4 * i + 1] = s - u; t[4 * i + 2] = d - e; t[4 * i + 3] = d + e;
        }
        for (int i = 0; i < 4; i++) {
            const int s = t[i] + t[8 + i], d = t[i] - t[8 + i];
            const int e = t[4 + i] - t[12 + i], u = t[4 + i] + t[12 + i];
            v[16 + 4 * i + 0] = ((s + u) * qmul + 128) >> 8;
            v[16 + 4 * i + 1] = ((d + e) * qmul + 128) >> 8;
            v[16 + 4 * i + 2] = ((d - e) * qmul + 128) >> 8;
            v[16 + 4 * i + 3] = ((s - u) * qmul + 128) >> 8;
        }
    } else if (kind == 1) {             /* chroma 4:2:0, :312-324 */
        const int a = v[0], b = v[1], c = v[2], d = v[3];
        const int s0 = a + b, d0 = a - b, s1 = c + d, d1 = c - d;
        v[0] = ((s0 + s1) * qmul) >> 7; v[1] = ((d0 + d1) * qmul) >> 7; v[2] = ((s0 - s1) * qmul) >> 7; v[3] = ((d0 - d1) * qmul) >> 7;
    } else {                            /* chroma 4:2:2, :275-310 */
        int t[8];
        for (int i = 0; i < 4; i++) { t[2 * i] = v[2 * i] + v[2 * i + 1]; t[2 * i + 1] = v[2 * i] - v[2 * i + 1]; }
        for (int i = 0; i < 2; i++) {
            const int z0 = t[i] + t[4 + i], z1 = t[i] - t[4 + i], z2 = t[2 + i] - t[6 + i], z3 = t[2 + i] + t[6 + i];
            v[0 + i] = ((z0 + z3) * qmul + 128) >> 8;
            v[2 + i] = ((z1 + z2) * qmul + 128) >> 8;
            v[4 + i] = ((z1 - z2) * qmul + 128) >> 8;
            v[6 + i] = ((z0 - z3) * qmul + 128) >> 8;
        }
    }
}
void t_luma_dc_dequant_idct(int16_t *output, int16_t *input, int qmul)
{
    Arena &a = arena();
    const size_t off = a.take(32 * 4);
    std::memcpy(a.h<int32_t>(off), input, 64);
    a.upload();
    LAUNCH_HBD(k_hbd_dc, a, a.d<int32_t>(off), qmul, 0);
    a.download();
    const int32_t *o = a.h<int32_t>(off) + 16;
    for (int k = 0; k < 16; k++) reinterpret_cast<int32_t *>(output)[luma_dc_slot(k)] = o[k];
}
void t_chroma_dc_dequant_idct(int16_t *block16, int qmul)
{
    int32_t *block = reinterpret_cast<int32_t *>(block16);
    Arena &a = arena();
    const size_t off = a.take(16);
    int32_t *h = a.h<int32_t>(off);
    for (int k = 0; k < 4; k++) h[k] = block[16 * k];
    a.upload();
    LAUNCH_HBD(k_hbd_dc, a, a.d<int32_t>(off), qmul, 1);
    a.download();
    for (int k = 0; k < 4; k++) block[16 * k] = h[k];
}
void t_chroma422_dc_dequant_idct(int16_t *block16, int qmul)
{
    int32_t *block = reinterpret_cast<int32_t *>(block16);
    Arena &a = arena();
    const size_t off = a.take(32);
    int32_t *h = a.h<int32_t>(off);
    for (int i = 0; i < 4; i++) { h[2 * i] = block[32 * i]; h[2 * i + 1] = block[32 * i + 16]; }
    a.upload();
    LAUNCH_HBD(k_hbd_dc, a, a.d<int32_t>(off), qmul, 2);
    a.download();
    for (int i = 0; i < 4; i++) { block[32 * i] = h[2 * i]; block[32 * i + 16] = h[2 * i + 1]; }
}

/* transform-bypass residual add, h264addpx_template.c:30-72: no clipping, wraps like the sample type */
__global__ void __launch_bounds__(64) k_hbd_add_pixels(px *dst, int pitch, const int32_t *blk, int n)
{
    for (int i = lane_id(); i < n * n; i += 64) {
        const int y = i / n, x = i - y * n;
        dst[y * pitch + x] = (px)(dst[y * pitch + x] + blk[i]);
    }
}
template <int N> void add_pixels_clear_shim(uint8_t *dst, int16_t *block, int stride)
{
    Arena &a = arena();
    Win w = win_pack(a, dst, stride, N * 2, N);
    const size_t b = a.take(N * N * 4);
    std::memcpy(a.h<int32_t>(b), block, N * N * 4);
    a.upload();
    LAUNCH_HBD(k_hbd_add_pixels, a, a.d<px>(w.off), w.pitch / 2, a.d<const int32_t>(b), N);
    a.download();
    win_unpack(a, w, dst, stride, 0, 0, N * 2, N);
    std::memset(block, 0, N * N * 4);
}

/* ---- deblocking edge filters: h264dsp_template.c:104-330 ----------------------------------------------------------------- */
struct LfJob {
    int xs, ys, nlines, inner, alpha, beta, kind;      /* kind: 0 luma, 1 luma intra, 2 chroma, 3 chroma intra; strides in samples */
    int R;
    int tc[4];                                          /* already scaled for the bit depth */
};
template <int BD>
__global__ void __launch_bounds__(64) k_hbd_loopfilter(px *win, const LfJob *jp)
{
    constexpr int MAXV = (1 << BD) - 1;
    const LfJob j = *jp;
    const int n = lane_id();
    if (n >= j.nlines) return;
    px *c = win + j.R * j.xs + n * j.ys;      /* q0 */
#define PX(k) c[(k) * j.xs]
    if (j.kind == 0) {
        int p2 = PX(-3), p1 = PX(-2), p0 = PX(-1), q0 = PX(0), q1 = PX(1), q2 = PX(2);
        lf_luma_line<MAXV>(p2, p1, p0, q0, q1, q2, j.alpha, j.beta, j.tc[n / j.inner]);
        PX(-2) = (px)p1; PX(-1) = (px)p0; PX(0) = (px)q0; PX(1) = (px)q1;
    } else if (j.kind == 1) {
        int p3 = PX(-4), p2 = PX(-3), p1 = PX(-2), p0 = PX(-1), q0 = PX(0), q1 = PX(1), q2 = PX(2), q3 = PX(3);
        lf_luma_intra_line(p3, p2, p1, p0, q0, q1, q2, q3, j.alpha, j.beta);
        PX(-3) = (px)p2; PX(-2) = (px)p1; PX(-1) = (px)p0; PX(0) = (px)q0; PX(1) = (px)q1; PX(2) = (px)q2;
    } else {
        int p1 = PX(-2), p0 = PX(-1), q0 = PX(0), q1 = PX(1);
        if (j.kind == 2) lf_chroma_line<MAXV>(p1, p0, q0, q1, j.alpha, j.beta, j.tc[n / j.inner]);
        else lf_chroma_intra_line(p1, p0, q0, q1, j.alpha, j.beta);
        PX(-1) = (px)p0; PX(0) = (px)q0;
    }
#undef PX
}
/* vertical_edge: samples across the edge are adjacent in memory ("h_loop_filter") */
template <int BD>
void lf_shim(uint8_t *pix, int stride, int alpha, int beta, const int8_t *tc0, int kind, int vertical_edge, int inner)
{
    Arena &a = arena();
    const int R = kind == 1 ? 4 : (kind == 0 ? 3 : 2), W = kind <= 1 ? 3 : 1;
    const int nlines = 4 * inner;
    Win w = vertical_edge ? win_pack(a, pix - 2 * R, stride, 4 * R, nlines)
                          : win_pack(a, pix - R * (ptrdiff_t)stride, stride, nlines * 2, 2 * R);
    const size_t joff = a.take(sizeof(LfJob));
    LfJob *j = a.h<LfJob>(joff);
    j->xs = vertical_edge ? 1 : w.pitch / 2;
    j->ys = vertical_edge ? w.pitch / 2 : 1;
    j->nlines = nlines; j->inner = inner; j->kind = kind; j->R = R;
    j->alpha = alpha << (BD - 8); j->beta = beta << (BD - 8);               /* :110-111 */
    for (int i = 0; i < 4; i++) {
        const int t = tc0 ? tc0[i] : 0;
        j->tc[i] = kind == 0 ? t * (1 << (BD - 8)) : (t - 1) * (1 << (BD - 8)) + 1;      /* :113, :228 */
    }
    a.upload();
    LAUNCH_HBD(k_hbd_loopfilter<BD>, a, a.d<px>(w.off), a.d<LfJob>(joff));
    a.download();
    if (vertical_edge) win_unpack(a, w, pix - 2 * W, stride, 2 * (R - W), 0, 4 * W, nlines);
    else               win_unpack(a, w, pix - W * (ptrdiff_t)stride, stride, 0, R - W, nlines * 2, 2 * W);
}
#define LF_TC(name, kind, vert, inner) \
    template <int BD> void name(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0) { lf_shim<BD>(pix, stride, alpha, beta, tc0, kind, vert, inner); }
#define LF_IN(name, kind, vert, inner) \
    template <int BD> void name(uint8_t *pix, int stride, int alpha, int beta) { lf_shim<BD>(pix, stride, alpha, beta, nullptr, kind, vert, inner); }
LF_TC(v_lf_luma, 0, 0, 4) LF_TC(h_lf_luma, 0, 1, 4) LF_TC(h_lf_luma_mbaff, 0, 1, 2)
LF_IN(v_lf_luma_intra, 1, 0, 4) LF_IN(h_lf_luma_intra, 1, 1, 4) LF_IN(h_lf_luma_mbaff_intra, 1, 1, 2)
LF_TC(v_lf_chroma, 2, 0, 2) LF_TC(h_lf_chroma, 2, 1, 2) LF_TC(h_lf_chroma_mbaff, 2, 1, 1)
LF_IN(v_lf_chroma_intra, 3, 0, 2) LF_IN(h_lf_chroma_intra, 3, 1, 2) LF_IN(h_lf_chroma_mbaff_intra, 3, 1, 1)
LF_TC(h_lf_chroma422, 2, 1, 4) LF_TC(h_lf_chroma422_mbaff, 2, 1, 2)
LF_IN(h_lf_chroma422_intra, 3, 1, 4) LF_IN(h_lf_chroma422_mbaff_intra, 3, 1, 2)
#undef LF_TC
#undef LF_IN

/* ---- intra prediction: h264pred_template.c, the predictors of h264_dev.h on 16-bit samples ----------------------------- */
struct PredJob {
    uint16_t T[1 + 32], L[1 + 16];
    int kind, mode, has_tl, has_tr;
};
template <int BD>
__global__ void __launch_bounds__(64) k_hbd_pred(const PredJob *jp, px *out, int pitch)
{
    __shared__ PredScratch s;
    const int lane = lane_id();
    if (lane < 33) s.T[lane] = jp->T[lane];
    if (lane < 17) s.L[lane] = jp->L[lane];
    __syncthreads();
    intra_pred_wave<px, BD>(s, jp->kind, jp->mode, jp->has_tl, jp->has_tr, out, pitch);
}
/* gather only the edge samples the reference reads for this (kind, mode, availability): as the 8-bit shim */
template <int BD>
void pred_shim(uint8_t *src8, ptrdiff_t stride, int kind, int mode, int has_tl, int has_tr, const uint8_t *topright8)
{
    Arena &a = arena();
    const px *src = reinterpret_cast<const px *>(src8), *topright = reinterpret_cast<const px *>(topright8);
    const ptrdiff_t st = stride / 2;
    const int N = kind == 0 ? 4 : (kind == 3 ? 16 : 8), NH = kind == 4 ? 16 : N;
    const size_t joff = a.take(sizeof(PredJob));
    PredJob *j = a.h<PredJob>(joff);
    std::memset(j, 0, sizeof(*j));
    j->kind = kind; j->mode = mode; j->has_tl = has_tl; j->has_tr = has_tr;
    int top = 0, left = 0, corner = 0, tr = 0;
    if (kind <= 1) {
        const int needs = pred_luma_needs(mode);
        top = needs & 1; left = (needs >> 1) & 1; corner = (needs >> 2) & 1; tr = (needs >> 3) & 1;
        if (kind == 1) {
            if ((top || left) && has_tl) corner = 1;
            if (top && has_tr) tr = 1;
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
    if (top) for (int i = 0; i < N; i++) j->T[1 + i] = (int16_t)src[i - st];
    if (left) for (int i = 0; i < NH; i++) j->L[1 + i] = (int16_t)src[-1 + i * st];
    if (corner) j->T[0] = j->L[0] = (int16_t)src[-1 - st];
    if (tr) {
        if (kind == 0) for (int i = 0; i < 4; i++) j->T[5 + i] = (int16_t)topright[i];
        else for (int i = 0; i < 8; i++) j->T[9 + i] = (int16_t)src[8 + i - st];
    }
    const size_t ooff = a.take((size_t)NH * 32);
    a.upload();
    LAUNCH_HBD(k_hbd_pred<BD>, a, a.d<PredJob>(joff), a.d<px>(ooff), 16);
    a.download();
    const uint8_t *o = a.h<uint8_t>(ooff);
    for (int y = 0; y < NH; y++) std::memcpy(src8 + y * stride, o + y * 32, (size_t)N * 2);
}
template <int M, int BD> void p4_shim(uint8_t *s, const uint8_t *tr, ptrdiff_t st) { pred_shim<BD>(s, st, 0, M, 0, 1, tr); }
template <int M, int BD> void p8l_shim(uint8_t *s, int tl, int tr, ptrdiff_t st) { pred_shim<BD>(s, st, 1, M, tl != 0, tr != 0, nullptr); }
template <int M, int BD> void p8_shim(uint8_t *s, ptrdiff_t st) { pred_shim<BD>(s, st, 2, M, 0, 0, nullptr); }
template <int M, int BD> void p16_shim(uint8_t *s, ptrdiff_t st) { pred_shim<BD>(s, st, 3, M, 0, 0, nullptr); }
template <int M, int BD> void p8x16_shim(uint8_t *s, ptrdiff_t st) { pred_shim<BD>(s, st, 4, M, 0, 0, nullptr); }

/* lossless forms: prediction + residual as a running sum (h264pred_template.c:1127-1354), wrapping at 16 bits like
 * the reference's sample type; blocks that feed each other run in rounds (see the 8-bit kernel) */
struct PredAddJob {
    int32_t coef[16 * 16];
    uint8_t bx[16], by[16];
    int32_t n, size, horizontal, filtered, has_tl, has_tr;
};
__global__ void __launch_bounds__(64) k_hbd_pred_add(px *win, int pitch, const PredAddJob *job)
{
    const int lane = lane_id(), size = job->size, b = lane / size, i = lane - b * size;
    const bool mine = b < job->n;
    const int bx = mine ? job->bx[b] : 0, by = mine ? job->by[b] : 0, hz = job->horizontal;
#define PX(x, y) win[(by + (y)) * pitch + bx + (x)]
    for (int round = 0; round < 4; round++) {
        const int along = hz ? bx - 1 : by - 1;
        if (mine && (job->filtered || (along >> 2) == round) && (!job->filtered || round == 0)) {
            int v;
            if (!job->filtered) v = hz ? PX(-1, i) : PX(i, -1);
            else if (!hz) {
                const int c = PX(i, -1);
                const int lft = i == 0 ? (job->has_tl ? PX(-1, -1) : c) : PX(i - 1, -1);
                const int rgt = i == 7 ? (job->has_tr ? PX(8, -1) : c) : PX(i + 1, -1);
                v = (lft + 2 * c + rgt + 2) >> 2;
            } else {
                const int c = PX(-1, i);
                const int up = i == 0 ? (job->has_tl ? PX(-1, -1) : c) : PX(-1, i - 1);
                v = i == 7 ? (PX(-1, 6) + 3 * c + 2) >> 2 : (up + 2 * c + PX(-1, i + 1) + 2) >> 2;
            }
            const int32_t *blk = job->coef + b * size * size;
            for (int k = 0; k < size; k++) {
                v = (v + (hz ? blk[i * size + k] : blk[k * size + i])) & 0xFFFF;
                if (hz) PX(k, i) = (px)v; else PX(i, k) = (px)v;
            }
        }
        __syncthreads();
    }
#undef PX
}
void pred_add_run(uint8_t *pix, const int *offs, int nblk, int16_t *block, ptrdiff_t stride, int size, int horizontal,
                  int filtered, int has_tl, int has_tr)
{
    Arena &a = arena();
    int bx[16], by[16], minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
    for (int i = 0; i < nblk; i++) {
        const int off = offs ? offs[i] : 0;
        const int y = off >= 0 ? (off + (int)stride / 2) / (int)stride : -((-off + (int)stride / 2) / (int)stride);
        bx[i] = (off - y * (int)stride) / 2; by[i] = y;
        if (bx[i] < minx) minx = bx[i];
        if (by[i] < miny) miny = by[i];
        if (bx[i] + size > maxx) maxx = bx[i] + size;
        if (by[i] + size > maxy) maxy = by[i] + size;
    }
    int x0 = minx, y0 = miny, x1 = maxx;
    if (horizontal) x0 -= 1; else y0 -= 1;
    if (filtered && !horizontal) { x0 -= has_tl ? 1 : 0; x1 += has_tr ? 1 : 0; }
    if (filtered && horizontal) y0 -= has_tl ? 1 : 0;
    const int wx0 = minx - 1, wy0 = miny - 1;
    Win w = win_pack(a, nullptr, 0, (maxx + 1 - wx0) * 2, maxy - wy0, 0, 0);
    for (int y = y0; y < maxy; y++)
        std::memcpy(a.h<uint8_t>(w.off) + (size_t)(y - wy0) * w.pitch + 2 * (x0 - wx0), pix + y * stride + 2 * x0, (size_t)(x1 - x0) * 2);
    const size_t joff = a.take(sizeof(PredAddJob));
    PredAddJob *job = a.h<PredAddJob>(joff);
    std::memset(job, 0, sizeof(*job));
    std::memcpy(job->coef, block, 4 * (size_t)nblk * size * size);
    for (int i = 0; i < nblk; i++) { job->bx[i] = (uint8_t)(bx[i] - wx0); job->by[i] = (uint8_t)(by[i] - wy0); }
    job->n = nblk; job->size = size; job->horizontal = horizontal; job->filtered = filtered; job->has_tl = has_tl; job->has_tr = has_tr;
    a.upload();
    LAUNCH_HBD(k_hbd_pred_add, a, a.d<px>(w.off), w.pitch / 2, a.d<const PredAddJob>(joff));
    a.download();
    for (int i = 0; i < nblk; i++)
        win_unpack(a, w, pix + by[i] * stride + 2 * bx[i], stride, 2 * (bx[i] - wx0), by[i] - wy0, size * 2, size);
    std::memset(block, 0, 4 * (size_t)nblk * size * size);
}
template <int SIZE, int HZ> void pred_add_shim(uint8_t *pix, int16_t *block, ptrdiff_t stride) { pred_add_run(pix, nullptr, 1, block, stride, SIZE, HZ, 0, 0, 0); }
template <int HZ> void pred8x8l_filter_add_shim(uint8_t *pix, int16_t *block, int tl, int tr, ptrdiff_t stride)
{
    pred_add_run(pix, nullptr, 1, block, stride, 8, HZ, 1, tl != 0, tr != 0);
}
template <int NBLK, int HZ> void pred_multi_add_shim(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    pred_add_run(pix, block_offset, NBLK, block, stride, 4, HZ, 0, 0, 0);
}
template <int HZ> void pred8x16_add_shim(uint8_t *pix, const int *block_offset, int16_t *block, ptrdiff_t stride)
{
    int offs[8];
    for (int i = 0; i < 4; i++) { offs[i] = block_offset[i]; offs[4 + i] = block_offset[8 + i]; }
    pred_add_run(pix, offs, 8, block, stride, 4, HZ, 0, 0, 0);
}

/* ---- emulated_edge_mc for 16-bit samples (videodsp_template.c, ff_emulated_edge_mc_16) -------------------------------------- */
__global__ void __launch_bounds__(64)
k_hbd_emu_edge(px *buf, int bpitch, const px *region, int rpitch, int rx0, int ry0, int bw, int bh, int sx, int sy, int w, int h)
{
    for (int i = lane_id(); i < bw * bh; i += 64) {
        const int y = i / bw, x = i - y * bw;
        const int cx = clip3(sx + x, 0, w - 1), cy = clip3(sy + y, 0, h - 1);
        buf[y * bpitch + x] = region[(cy - ry0) * rpitch + (cx - rx0)];
    }
}
void t_emulated_edge_mc(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize,
                        int block_w, int block_h, int src_x, int src_y, int w, int h)
{
    if (!w || !h) return;
    auto cl = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const int rx0 = cl(src_x, 0, w - 1), rx1 = cl(src_x + block_w - 1, 0, w - 1);
    const int ry0 = cl(src_y, 0, h - 1), ry1 = cl(src_y + block_h - 1, 0, h - 1);
    const uint8_t *origin = src - src_y * src_linesize - 2 * (ptrdiff_t)src_x;
    for (int y0 = 0; y0 < block_h; y0 += 64) {
        Arena &a = arena();
        const int bh = block_h - y0 < 64 ? block_h - y0 : 64;
        Win r = win_pack(a, origin + ry0 * src_linesize + 2 * rx0, src_linesize, (rx1 - rx0 + 1) * 2, ry1 - ry0 + 1);
        Win o = win_pack(a, nullptr, 0, block_w * 2, bh, 0, 0);
        a.upload();
        LAUNCH_HBD(k_hbd_emu_edge, a, a.d<px>(o.off), o.pitch / 2, a.d<px>(r.off), r.pitch / 2, rx0, ry0, block_w, bh, src_x, src_y + y0, w, h);
        a.download();
        win_unpack(a, o, buf + y0 * buf_linesize, buf_linesize, 0, 0, block_w * 2, bh);
    }
}

template <int BD> void fill_dsp(H264DSPContext *c, int chroma_format_idc)
{
    c->weight_h264_pixels_tab[0] = weight_shim<16, BD>;   c->weight_h264_pixels_tab[1] = weight_shim<8, BD>;
    c->weight_h264_pixels_tab[2] = weight_shim<4, BD>;    c->weight_h264_pixels_tab[3] = weight_shim<2, BD>;
    c->biweight_h264_pixels_tab[0] = biweight_shim<16, BD>; c->biweight_h264_pixels_tab[1] = biweight_shim<8, BD>;
    c->biweight_h264_pixels_tab[2] = biweight_shim<4, BD>;  c->biweight_h264_pixels_tab[3] = biweight_shim<2, BD>;
    c->h264_v_loop_filter_luma = v_lf_luma<BD>;
    c->h264_h_loop_filter_luma = h_lf_luma<BD>;
    c->h264_h_loop_filter_luma_mbaff = h_lf_luma_mbaff<BD>;
    c->h264_v_loop_filter_luma_intra = v_lf_luma_intra<BD>;
    c->h264_h_loop_filter_luma_intra = h_lf_luma_intra<BD>;
    c->h264_h_loop_filter_luma_mbaff_intra = h_lf_luma_mbaff_intra<BD>;
    c->h264_v_loop_filter_chroma = v_lf_chroma<BD>;
    c->h264_v_loop_filter_chroma_intra = v_lf_chroma_intra<BD>;
    c->h264_idct_add = t_idct_add<BD>;
    c->h264_idct8_add = t_idct8_add<BD>;
    c->h264_idct_dc_add = t_idct_dc_add<BD>;
    c->h264_idct8_dc_add = t_idct8_dc_add<BD>;
    c->h264_idct_add16 = t_idct_add16<BD>;
    c->h264_idct8_add4 = t_idct8_add4<BD>;
    c->h264_idct_add16intra = t_idct_add16intra<BD>;
    c->h264_luma_dc_dequant_idct = t_luma_dc_dequant_idct;
    c->h264_add_pixels4_clear = add_pixels_clear_shim<4>;
    c->h264_add_pixels8_clear = add_pixels_clear_shim<8>;
    if (chroma_format_idc <= 1) {
        c->h264_h_loop_filter_chroma = h_lf_chroma<BD>;
        c->h264_h_loop_filter_chroma_mbaff = h_lf_chroma_mbaff<BD>;
        c->h264_h_loop_filter_chroma_intra = h_lf_chroma_intra<BD>;
        c->h264_h_loop_filter_chroma_mbaff_intra = h_lf_chroma_mbaff_intra<BD>;
        c->h264_idct_add8 = t_idct_add8<BD>;
        c->h264_chroma_dc_dequant_idct = t_chroma_dc_dequant_idct;
    } else {
        c->h264_h_loop_filter_chroma = h_lf_chroma422<BD>;
        c->h264_h_loop_filter_chroma_mbaff = h_lf_chroma422_mbaff<BD>;
        c->h264_h_loop_filter_chroma_intra = h_lf_chroma422_intra<BD>;
        c->h264_h_loop_filter_chroma_mbaff_intra = h_lf_chroma422_mbaff_intra<BD>;
        c->h264_idct_add8 = t_idct_add8_422<BD>;
        c->h264_chroma_dc_dequant_idct = t_chroma422_dc_dequant_idct;
    }
}
template <int BD> void fill_qpel(H264QpelContext *c)
{
#define QROW(tab, idx, SIZE, AVG) \
    c->tab[idx][0] = qpel_shim<SIZE, 0, AVG, BD>;   c->tab[idx][1] = qpel_shim<SIZE, 1, AVG, BD>;   \
    c->tab[idx][2] = qpel_shim<SIZE, 2, AVG, BD>;   c->tab[idx][3] = qpel_shim<SIZE, 3, AVG, BD>;   \
    c->tab[idx][4] = qpel_shim<SIZE, 4, AVG, BD>;   c->tab[idx][5] = qpel_shim<SIZE, 5, AVG, BD>;   \
    c->tab[idx][6] = qpel_shim<SIZE, 6, AVG, BD>;   c->tab[idx][7] = qpel_shim<SIZE, 7, AVG, BD>;   \
    c->tab[idx][8] = qpel_shim<SIZE, 8, AVG, BD>;   c->tab[idx][9] = qpel_shim<SIZE, 9, AVG, BD>;   \
    c->tab[idx][10] = qpel_shim<SIZE, 10, AVG, BD>; c->tab[idx][11] = qpel_shim<SIZE, 11, AVG, BD>; \
    c->tab[idx][12] = qpel_shim<SIZE, 12, AVG, BD>; c->tab[idx][13] = qpel_shim<SIZE, 13, AVG, BD>; \
    c->tab[idx][14] = qpel_shim<SIZE, 14, AVG, BD>; c->tab[idx][15] = qpel_shim<SIZE, 15, AVG, BD>;
    QROW(put_h264_qpel_pixels_tab, 0, 16, 0) QROW(put_h264_qpel_pixels_tab, 1, 8, 0)
    QROW(put_h264_qpel_pixels_tab, 2, 4, 0)  QROW(put_h264_qpel_pixels_tab, 3, 2, 0)
    QROW(avg_h264_qpel_pixels_tab, 0, 16, 1) QROW(avg_h264_qpel_pixels_tab, 1, 8, 1)
    QROW(avg_h264_qpel_pixels_tab, 2, 4, 1)
#undef QROW
}
template <int BD> void fill_pred(H264PredContext *h, int chroma_format_idc)
{
#define P12(tab, shim) \
    h->tab[0] = shim<0, BD>; h->tab[1] = shim<1, BD>; h->tab[2] = shim<2, BD>; h->tab[3] = shim<3, BD>; h->tab[4] = shim<4, BD>; h->tab[5] = shim<5, BD>; \
    h->tab[6] = shim<6, BD>; h->tab[7] = shim<7, BD>; h->tab[8] = shim<8, BD>; h->tab[9] = shim<9, BD>; h->tab[10] = shim<10, BD>;
    P12(pred4x4, p4_shim)  h->pred4x4[11] = p4_shim<11, BD>;
    P12(pred8x8l, p8l_shim) h->pred8x8l[11] = p8l_shim<11, BD>;
    if (chroma_format_idc <= 1) { P12(pred8x8, p8_shim) } else { P12(pred8x8, p8x16_shim) }
#undef P12
    h->pred16x16[0] = p16_shim<0, BD>; h->pred16x16[1] = p16_shim<1, BD>; h->pred16x16[2] = p16_shim<2, BD>; h->pred16x16[3] = p16_shim<3, BD>;
    h->pred16x16[4] = p16_shim<4, BD>; h->pred16x16[5] = p16_shim<5, BD>; h->pred16x16[6] = p16_shim<6, BD>;
    h->pred4x4_add[0] = pred_add_shim<4, 0>;   h->pred4x4_add[1] = pred_add_shim<4, 1>;
    h->pred8x8l_add[0] = pred_add_shim<8, 0>;  h->pred8x8l_add[1] = pred_add_shim<8, 1>;
    h->pred8x8l_filter_add[0] = pred8x8l_filter_add_shim<0>; h->pred8x8l_filter_add[1] = pred8x8l_filter_add_shim<1>;
    if (chroma_format_idc <= 1) { h->pred8x8_add[2] = pred_multi_add_shim<4, 0>; h->pred8x8_add[1] = pred_multi_add_shim<4, 1>; }
    else { h->pred8x8_add[2] = pred8x16_add_shim<0>; h->pred8x8_add[1] = pred8x16_add_shim<1>; }
    h->pred16x16_add[2] = pred_multi_add_shim<16, 0>; h->pred16x16_add[1] = pred_multi_add_shim<16, 1>;
}

}  // namespace

/* called by the hooks of h264_tier1.hip for bit depths above 8 */
namespace mi355 {
void h264dsp_init_hbd(H264DSPContext *c, int bit_depth, int chroma_format_idc)
{
    if (bit_depth == 9) fill_dsp<9>(c, chroma_format_idc);
    else if (bit_depth == 10) fill_dsp<10>(c, chroma_format_idc);
}
void h264qpel_init_hbd(H264QpelContext *c, int bit_depth)
{
    if (bit_depth == 9) fill_qpel<9>(c);
    else if (bit_depth == 10) fill_qpel<10>(c);
}
void h264chroma_init_hbd(H264ChromaContext *c, int bit_depth)
{
    if (bit_depth != 9 && bit_depth != 10) return;
    c->put_h264_chroma_pixels_tab[0] = chroma_shim<8, 0>; c->put_h264_chroma_pixels_tab[1] = chroma_shim<4, 0>;
    c->put_h264_chroma_pixels_tab[2] = chroma_shim<2, 0>;
    c->avg_h264_chroma_pixels_tab[0] = chroma_shim<8, 1>; c->avg_h264_chroma_pixels_tab[1] = chroma_shim<4, 1>;
    c->avg_h264_chroma_pixels_tab[2] = chroma_shim<2, 1>;
}
void h264pred_init_hbd(H264PredContext *h, int bit_depth, int chroma_format_idc)
{
    if (bit_depth == 9) fill_pred<9>(h, chroma_format_idc);
    else if (bit_depth == 10) fill_pred<10>(h, chroma_format_idc);
}
void videodsp_init_hbd(VideoDSPContext *ctx, int bpc)
{
    if (bpc > 8 && bpc <= 16) ctx->emulated_edge_mc = t_emulated_edge_mc;
}
}  // namespace mi355
