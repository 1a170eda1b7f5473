/*
 * hevc_ctb_fast.h — the matrix unit on byte planes for the two hot shapes of an HEVC coding tree block (k_hevc_recon_ctbs, hevc_ctb.hip):
 *
 *   - 16x16 / 32x32 inverse DCT (hevcdsp_template.c:140-236, matrix hevcdsp.c:25-90): both passes are products with the transform matrix,
 *     whose entries fit a signed byte (|t| <= 90).  The 16-bit operand (coefficients; the first pass's clipped results) goes in as a
 *     high-byte and a low-byte plane: x = 256 hi + (lo' + 128) with hi = x >> 8 (signed) and lo' = (x & 255) - 128, so
 *     sum x t = 256 sum hi t + sum lo' t + 128 sum t — two v_mfma_i32_16x16x32_i8 per 16x16 tile of results, the constant (and the
 *     pass's rounding term) in the accumulator's start value.  Exact integers throughout; (x + 64) >> 7 and the clip to int16 sit
 *     between the passes as the reference has them (IDCT :208-236).
 *   - 8-tap / 4-tap motion compensation to the 14-bit intermediate (put_hevc_qpel / put_hevc_epel :729-1089) and put_unweighted_pred
 *     (:1091-1113) of blocks whose sides are multiples of 16: a pass is a product of the window with a Toeplitz matrix of the taps.
 *
 * What makes the chain of products cheap is that the unit's result layout IS its operand layout for a contraction over the result's
 * rows: a lane receives D[4 (l / 16) + t][l % 16], t = 0..3 — four values that are CONSECUTIVE in the row index — and supplies, as an
 * operand, eight values consecutive in the contraction index for row / column l % 16.  The order of a contraction's terms is free as
 * long as both operands use the same one, so the next product takes slot (g, s) = term 16 (s / 4) + 4 g + s % 4: the lane's own four
 * results of two row tiles, packed to bytes, no exchange between lanes and no LDS.  Coefficients arrive row-major (the contraction of the
 * first pass runs down the columns): a product with a shifted identity transposes them first (the unit as the transposer, as
 * h264_recon_fast.h does for H.264's vertical filters).
 *
 * Included inside the anonymous namespace of hevc_ctb.hip, after hevc_dev.h (dct_coef, clip_i16, pk_*).
 */
#ifndef MI355_HEVC_CTB_FAST_H
#define MI355_HEVC_CTB_FAST_H

/* ---- primitives: one instruction each on the device, their plain meaning in the emulator ---- */
#ifdef MI355_HIP_EMU_H
/* v_perm_b32: byte i of the result = byte sel[i] of {s0 (4..7), s1 (0..3)}; 0x0C = 0 */
static inline uint32_t cf_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t b = (sel >> (8 * i)) & 0xFF;
        const uint32_t v = b <= 7 ? (uint32_t)((src >> (8 * b)) & 0xFF) : (b >= 0x0D ? 0xFFu : 0u);
        r |= v << (8 * i);
    }
    return r;
}
/* v_mfma_i32_16x16x32_i8: D[i][j] = c + sum_k A[i][k] B[k][j], signed bytes; lane l supplies bytes 8 (l / 16) .. + 7 of row l % 16 of A and of
 * column l % 16 of B and receives D[4 (l / 16) + t][l % 16], t = 0..3 */
static inline void cf_mfma(uint64_t a, uint64_t b, const int c[4], int d[4])
{
    const int lane = (int)(threadIdx.x & 63), g = lane >> 4, j = lane & 15;
    for (int t = 0; t < 4; t++) d[t] = c[t];
    for (int kg = 0; kg < 4; kg++) {
        const uint64_t bv = (uint64_t)(uint32_t)__shfl((int)(uint32_t)b, j + 16 * kg) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(b >> 32), j + 16 * kg) << 32);
        for (int t = 0; t < 4; t++) {
            const uint64_t av = (uint64_t)(uint32_t)__shfl((int)(uint32_t)a, 4 * g + t + 16 * kg) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(a >> 32), 4 * g + t + 16 * kg) << 32);
            for (int k = 0; k < 8; k++) d[t] += (int)(int8_t)(av >> (8 * k)) * (int)(int8_t)(bv >> (8 * k));
        }
    }
}
static inline uint64_t cf_lds64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
static inline void cf_st64(uint8_t *p, uint64_t v) { std::memcpy(p, &v, 8); }
#else
__device__ __forceinline__ uint32_t cf_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
typedef int cf_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cf_mfma(uint64_t a, uint64_t b, const int c[4], int d[4])
{
    cf_v4i acc = { c[0], c[1], c[2], c[3] };
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a, (long)b, acc, 0, 0, 0);
    d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3];
}
__device__ __forceinline__ uint64_t cf_lds64(const uint8_t *p) { return *reinterpret_cast<const uint64_t *>(p); }       /* p on 8 bytes */
__device__ __forceinline__ void cf_st64(uint8_t *p, uint64_t v) { *reinterpret_cast<uint64_t *>(p) = v; }
#endif
__device__ __forceinline__ uint64_t cf_u64(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

/* four 32-bit values whose low bytes are wanted -> one dword of them */
__device__ __forceinline__ uint32_t cf_pack_b0(const int v[4])
{
    const uint32_t a = cf_perm((uint32_t)v[1], (uint32_t)v[0], 0x0C0C0400u), b = cf_perm((uint32_t)v[3], (uint32_t)v[2], 0x0C0C0400u);
    return cf_perm(b, a, 0x05040100u);
}
/* four values in int16 range -> their low bytes (lo) and their high bytes (hi), a dword each */
__device__ __forceinline__ void cf_split4(const int v[4], uint32_t &lo, uint32_t &hi)
{
    const uint32_t a = cf_perm((uint32_t)v[1], (uint32_t)v[0], 0x05010400u), b = cf_perm((uint32_t)v[3], (uint32_t)v[2], 0x05010400u);     /* lo0 lo1 hi0 hi1 */
    lo = cf_perm(b, a, 0x05040100u);
    hi = cf_perm(b, a, 0x07060302u);
}
/* eight int16 (four dwords) -> their eight low bytes and their eight high bytes */
__device__ __forceinline__ void cf_planes8(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint64_t &lo, uint64_t &hi)
{
    lo = cf_u64(cf_perm(w1, w0, 0x06040200u), cf_perm(w3, w2, 0x06040200u));
    hi = cf_u64(cf_perm(w1, w0, 0x07050301u), cf_perm(w3, w2, 0x07050301u));
}
constexpr uint32_t CF_SIGN4 = 0x80808080u;
constexpr uint64_t CF_SIGN8 = 0x8080808080808080ull;

/* ---- the transform matrix as operands (compile-time tables): word [t][lane] = for output index n = 16 t + lane % 16 the eight matrix entries
 * T[k][n] of the lane's slots, k = 16 (s / 4) + 4 (lane / 16) + s % 4 (rows of the SIZE-point matrix; slots past it: 0).  The same word serves
 * as the second operand of the first pass (column n of T) and as the first operand of the second pass (row n of T transposed). */
struct CfTabs {
    uint64_t op32[2][64];
    uint64_t op16[64];
    int32_t sum32[32], sum16[16];       /* sum over k of T[k][n]: what the low plane's offset of 128 adds up to */
};
constexpr uint64_t cf_op_word(int size, int t, int lane)
{
    const int j = lane & 15, g = lane >> 4;
    uint64_t w = 0;
    for (int s = 0; s < 8; s++) {
        const int k = 16 * (s / 4) + 4 * g + s % 4;
        const int v = k < size ? dct_coef(k * (32 / size), 16 * t + j) : 0;
        w |= (uint64_t)(uint8_t)(int8_t)v << (8 * s);
    }
    return w;
}
constexpr CfTabs cf_make_tabs()
{
    CfTabs t{};
    for (int l = 0; l < 64; l++) { t.op32[0][l] = cf_op_word(32, 0, l); t.op32[1][l] = cf_op_word(32, 1, l); t.op16[l] = cf_op_word(16, 0, l); }
    for (int n = 0; n < 32; n++) { int s = 0; for (int k = 0; k < 32; k++) s += dct_coef(k, n); t.sum32[n] = s; }
    for (int n = 0; n < 16; n++) { int s = 0; for (int k = 0; k < 16; k++) s += dct_coef(2 * k, n); t.sum16[n] = s; }
    return t;
}
__device__ const CfTabs k_cf_tabs = cf_make_tabs();

/* ---- inverse DCT 16x16 / 32x32 + add_residual into the block's samples in LDS ---------------------------------------------- */
struct CfRaw { uint32_t w[2][4]; };      /* a lane's coefficients as fetched: row 16 kt + lane % 16, columns 8 (lane / 16) .. + 7, kt = 0, 1 */

/* which rows / columns of a block carry coefficients: one 16x16 corner unless col_limit says more (include/mi355_hevc_batch.h: rows and
 * columns 0 .. col_limit + 3) */
template <int LOG2> __device__ __forceinline__ bool cf_idct_dense(int col_limit) { return LOG2 == 5 && col_limit + 4 > 16; }

/* issue the loads (16 bytes per lane and row tile; the lanes of a request cover sixteen whole rows = 1 KB of the block, contiguous) */
template <int LOG2>
__device__ __forceinline__ void cf_idct_load(CfRaw &raw, const uint8_t *coeffs, int col_limit, int lane)
{
    constexpr int N = 1 << LOG2;
    const int j = lane & 15, g = lane >> 4;
    const bool dense = cf_idct_dense<LOG2>(col_limit);
#pragma unroll
    for (int kt = 0; kt < 2; kt++) {
        const bool have = kt == 0 ? (dense || g < 2) : dense;      /* the corner only: columns 0..15 = the row's first two pieces */
        if (have) __builtin_memcpy(raw.w[kt], coeffs + 2 * ((16 * kt + j) * N + 8 * g), 16);
        else raw.w[kt][0] = raw.w[kt][1] = raw.w[kt][2] = raw.w[kt][3] = 0u;
    }
}

/* tile: the block's sample (0, 0) in LDS, `pitch` bytes per row; WIDE: 16-bit samples */
/* DENSE (compile-time: which tiles take part is then known, and the products of a pass are one stretch of straight-line code) */
template <int LOG2, bool WIDE, bool DENSE>
__device__ __forceinline__ void cf_idct_run_n(const CfRaw &raw, int bd, uint8_t *tile, int pitch, int lane)
{
    constexpr int NT = (1 << LOG2) / 16;
    const int j = lane & 15, g = lane >> 4;
    constexpr bool dense = DENSE;
    const int zero4[4] = { 0, 0, 0, 0 };
    /* byte planes of the fetched rows: first operands of the transposing products */
    uint64_t alo[2], ahi[2];
#pragma unroll
    for (int kt = 0; kt < 2; kt++) cf_planes8(raw.w[kt][0], raw.w[kt][1], raw.w[kt][2], raw.w[kt][3], alo[kt], ahi[kt]);
    /* transposed: a1[xt] = for column x = 16 xt + j the coefficients C[k][x] of the lane's eight slots */
    uint64_t a1lo[NT], a1hi[NT];
#pragma unroll
    for (int xt = 0; xt < NT; xt++) {
        a1lo[xt] = CF_SIGN8; a1hi[xt] = 0;
        if (xt == 0 || dense) {
            const uint64_t ident = g == 2 * xt + (j >> 3) ? 1ull << (8 * (j & 7)) : 0ull;        /* I[c][x'] = (c == 16 xt + x'), c = 8 g + s */
            uint32_t lo[2] = { 0u, 0u }, hi[2] = { 0u, 0u };
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
                if (kt == 0 || dense) {
                    int d[4];
                    cf_mfma(alo[kt], ident, zero4, d);
                    lo[kt] = cf_pack_b0(d);
                    cf_mfma(ahi[kt], ident, zero4, d);
                    hi[kt] = cf_pack_b0(d);
                }
            a1lo[xt] = cf_u64(lo[0], lo[1]) ^ CF_SIGN8;
            a1hi[xt] = cf_u64(hi[0], hi[1]);
        }
    }
    /* first pass, down the columns: D1[x][y] = sum_k C[k][x] T[k][y]; (sum + 64) >> 7, clipped to int16 (IDCT :219-224, SCALE :43) */
    uint64_t op[NT];
    int c1[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        op[t] = LOG2 == 5 ? k_cf_tabs.op32[t][lane] : k_cf_tabs.op16[lane];
        c1[t] = 128 * (LOG2 == 5 ? k_cf_tabs.sum32[16 * t + j] : k_cf_tabs.sum16[j]) + 64;
    }
    uint64_t b2lo[NT], b2hi[NT];
#pragma unroll
    for (int yt = 0; yt < NT; yt++) {
        uint32_t lo[2] = { CF_SIGN4, CF_SIGN4 }, hi[2] = { 0u, 0u };         /* a column tile without coefficients: zeros */
        const int cc[4] = { c1[yt], c1[yt], c1[yt], c1[yt] };
#pragma unroll
        for (int xt = 0; xt < NT; xt++)
            if (xt == 0 || dense) {
                int h[4], l[4], v[4];
                cf_mfma(a1hi[xt], op[yt], zero4, h);
                cf_mfma(a1lo[xt], op[yt], cc, l);
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = clip_i16(((h[t] << 8) + l[t]) >> 7);
                cf_split4(v, lo[xt], hi[xt]);
                lo[xt] ^= CF_SIGN4;
            }
        b2lo[yt] = cf_u64(lo[0], lo[1]);
        b2hi[yt] = cf_u64(hi[0], hi[1]);
    }
    /* second pass, along the rows: D2[x'][y] = sum_x T[x][x'] D1[x][y]; (sum + (1 << (19 - bd))) >> (20 - bd), clipped; then add_residual
     * (:43-82): the lane holds samples x' = 16 xt + 4 g .. + 3 of row y = 16 yt + j */
    const int shift2 = 20 - bd, add2 = 1 << (shift2 - 1), maxv = (1 << bd) - 1;
#pragma unroll
    for (int xt = 0; xt < NT; xt++) {
        int cc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) cc[t] = 128 * (LOG2 == 5 ? k_cf_tabs.sum32[16 * xt + 4 * g + t] : k_cf_tabs.sum16[4 * g + t]) + add2;
#pragma unroll
        for (int yt = 0; yt < NT; yt++) {
            int h[4], l[4], r[4];
            cf_mfma(op[xt], b2hi[yt], zero4, h);
            cf_mfma(op[xt], b2lo[yt], cc, l);
#pragma unroll
            for (int t = 0; t < 4; t++) r[t] = clip_i16(((h[t] << 8) + l[t]) >> shift2);
            const uint32_t r01 = pk_make(r[0], r[1]), r23 = pk_make(r[2], r[3]);
            uint8_t *p = tile + (16 * yt + j) * pitch + ((16 * xt + 4 * g) << (WIDE ? 1 : 0));
            if (WIDE) {
                uint2 *q = reinterpret_cast<uint2 *>(p);
                const uint2 s = *q;
                *q = make_uint2(pk_clip_max(pk_adds(s.x, r01), maxv), pk_clip_max(pk_adds(s.y, r23), maxv));
            } else {
                uint32_t *q = reinterpret_cast<uint32_t *>(p);
                const uint32_t s = *q;
                const uint32_t o01 = pk_clip_max(pk_adds(mi355_widen_lo(s), r01), maxv), o23 = pk_clip_max(pk_adds(mi355_widen_hi(s), r23), maxv);
                *q = cf_perm(o23, o01, 0x06040200u);
            }
        }
    }
}

template <int LOG2, bool WIDE>
__device__ __forceinline__ void cf_idct_run(const CfRaw &raw, int col_limit, int bd, uint8_t *tile, int pitch, int lane)
{
    if (cf_idct_dense<LOG2>(col_limit)) cf_idct_run_n<LOG2, WIDE, true>(raw, bd, tile, pitch, lane);
    else cf_idct_run_n<LOG2, WIDE, false>(raw, bd, tile, pitch, lane);
}

/* ---- motion compensation of one tile (16 or 32 samples each way) + put_unweighted_pred into LDS -------------------------------------- */
constexpr int CF_WIN_PITCH = 48, CF_WIN_ROWS = 48;
struct __attribute__((aligned(16))) CfWin {
    uint8_t lo[CF_WIN_ROWS * CF_WIN_PITCH];      /* low bytes of the window's samples, - 128 (^ 0x80): signed bytes */
    uint8_t hi[CF_WIN_ROWS * CF_WIN_PITCH];      /* high bytes (16-bit samples) */
};
/* eight taps as signed bytes in a 64-bit word -> the word whose byte s is tap[s + d] (0 outside 0..7) */
__device__ __forceinline__ uint64_t cf_taps_at(uint64_t taps, int d)
{
    if (d >= 8 || d <= -8) return 0ull;
    return d >= 0 ? taps >> (8 * d) : taps << (-8 * d);
}
struct CfPass { uint64_t taps; int ext, shift, sum; };      /* taps from the window's first row / column on; ext = taps - 1 */
__device__ __forceinline__ CfPass cf_pass(const int8_t *f, int ntaps, int shift)
{
    /* the table row IS the word (taps as consecutive signed bytes); every interpolation filter of the standard sums to 64 (hevcdsp.c:92-115) */
    CfPass p;
    if (ntaps == 8) { uint64_t w; __builtin_memcpy(&w, f, 8); p.taps = w; }
    else { uint32_t w; __builtin_memcpy(&w, f, 4); p.taps = w; }
    p.sum = 64;
    p.ext = ntaps - 1; p.shift = shift;
    return p;
}
__device__ __forceinline__ CfPass cf_pass_one(int tap, int shift)      /* no filter in this direction: one tap on the sample itself */
{
    CfPass p;
    p.taps = (uint64_t)(uint8_t)(int8_t)tap; p.sum = tap; p.ext = 0; p.shift = shift;
    return p;
}

/* src: the first sample the taps touch (block sample (0, 0) minus the taps before it in each FILTERED direction); sb: bytes per source row, a multiple of
 * the piece size (16 bytes of 16-bit samples, 8 of 8-bit ones: eight samples).  tw, th: 16 or 32.  The tile's samples go to `tile` (LDS, `pitch` bytes
 * per row) as put_unweighted_pred makes them. */
/* XT, YT: the tile's 16-sample columns / rows (1 or 2), RT: the 16-row tiles of the window (the tile's rows + the vertical taps' reach: YT or YT + 1) — compile-time,
 * so that the products of a pass are one stretch of straight-line code: their LDS reads go out together and the matrix unit's latency of one product is covered by
 * the next (with run-time bounds every product sat in a branch of its own: read, wait, product, wait, eight idle cycles, arithmetic — six times over) */
/* NPL = 2: the tile of BOTH chroma planes of a block (a 16x16 tile each: XT = YT = 1) — the two windows are fetched in ONE round trip into rows 0 .. and CF_WIN_SECOND .. of the
 * wave's window and worked on one after the other (one after the other from fetch to store, a chroma job's wave waited out two memory round trips while the luma job's wave
 * beside it waited out one: the chroma waves were the last at the block's barrier).  Both windows start at the same offset inside their first piece (the caller checks). */
constexpr int CF_WIN_SECOND = 24;
template <bool WIDE, int XT, int YT, int RT, int NPL = 1>
__device__ __forceinline__ void cf_mc_tile_n(CfWin &w, const uint8_t *src, const uint8_t *src_b, ptrdiff_t sb, const CfPass ph, const CfPass pv, int bd,
                                             uint8_t *tile, uint8_t *tile_b, int pitch, int lane)
{
    static_assert(NPL == 1 || (XT == 1 && YT == 1), "two planes: 16x16 tiles (19 rows of window each)");
    constexpr int tw = 16 * XT, th = 16 * YT;
    constexpr int PB = WIDE ? 16 : 8;
    const int j = lane & 15, g = lane >> 4;
    const int off = (int)((uintptr_t)src & (PB - 1)) >> (WIDE ? 1 : 0);
    const int rows = th + pv.ext, npr = (off + tw + ph.ext + 7) >> 3;
    /* the window(s) -> byte planes in LDS, eight samples per lane and piece, all loads in flight together.  A lane keeps its piece of a row and
     * walks down the rows (8 pieces a row and 8 rows a round for a 32-wide tile, 4 and 16 for a 16-wide one: addresses are the lane's first one plus a
     * round's constant; pieces past the row's last and rows past the window's last are not fetched) */
    {
        constexpr int ppr_log = tw > 16 ? 3 : 2, rstep = 64 >> ppr_log;
        const int r0 = lane >> ppr_log, p = lane & ((1 << ppr_log) - 1);
        const uint32_t voff = (uint32_t)r0 * (uint32_t)sb + (uint32_t)(PB * p);
        const int lds0 = r0 * CF_WIN_PITCH + 8 * p;
        constexpr int NR = 16 * RT / rstep;           /* rounds: the window has at most 16 RT rows */
        uint32_t v[NPL][NR][4];
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            const uint8_t *s0 = pl ? src_b : src, *base = s0 - ((uintptr_t)s0 & (PB - 1));
#pragma unroll
            for (int u = 0; u < NR; u++)
                if (p < npr && r0 + u * rstep < rows) {
                    const uint8_t *q = base + (ptrdiff_t)(u * rstep) * sb + voff;
                    if (WIDE) __builtin_memcpy(v[pl][u], q, 16);
                    else { __builtin_memcpy(v[pl][u], q, 8); v[pl][u][2] = v[pl][u][3] = 0u; }
                }
        }
        MI355_ISSUE_FENCE();
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
#pragma unroll
            for (int u = 0; u < NR; u++)
                if (p < npr && r0 + u * rstep < rows) {
                    const int at = lds0 + (u * rstep + pl * CF_WIN_SECOND) * CF_WIN_PITCH;
                    if (WIDE) {
                        uint64_t lo, hi;
                        cf_planes8(v[pl][u][0], v[pl][u][1], v[pl][u][2], v[pl][u][3], lo, hi);
                        cf_st64(w.lo + at, lo ^ CF_SIGN8);
                        cf_st64(w.hi + at, hi);
                    } else {
                        cf_st64(w.lo + at, cf_u64(v[pl][u][0], v[pl][u][1]) ^ CF_SIGN8);
                    }
                }
    }
    MI355_WAVE_SYNC();
    const int zero4[4] = { 0, 0, 0, 0 };
    /* A 16-sample COLUMN of the tile at a time, both passes (the first pass's results of one column — RT x 2 registers — are all the second pass of that column
     * needs: held for both columns at once they are twice the registers).
     * horizontal pass: t[r][x] = (sum_c W[r][c] tapH[c - off - x]) >> shift; the lane receives column x = 16 xt + j, rows 16 rt + 4 g .. + 3.
     * vertical pass: v[x][y] = (sum_r t[r][x] tapV[r - y]) >> shift over the rows of two row tiles (slot (g, s) = row 16 (s / 4) + 4 g + s % 4 of the
     * pair); the lane receives row y = 16 yt + j, columns 16 xt + 4 g .. + 3 of the 14-bit intermediate; put_unweighted_pred (:1092-1113) on top */
    const uint64_t toep_h = cf_taps_at(ph.taps, 8 * g - off - j);
    const int ch = 128 * ph.sum, ch4[4] = { ch, ch, ch, ch };
    const uint64_t toep_v = cf_u64((uint32_t)cf_taps_at(pv.taps, 4 * g - j), (uint32_t)cf_taps_at(pv.taps, 16 + 4 * g - j));
    /* ((sum >> shift) + rnd) >> sh14 with rnd = 1 << (sh14 - 1) is (sum + (rnd << shift)) >> (shift + sh14): one shift, the addend in the accumulator's start value */
    const int sh14 = 14 - bd, maxv = (1 << bd) - 1, shv = pv.shift + sh14;
    const int cv = 128 * pv.sum + ((1 << (sh14 - 1)) << pv.shift), cv4[4] = { cv, cv, cv, cv };
#pragma unroll
    for (int pl = 0; pl < NPL; pl++)
#pragma unroll
    for (int xt = 0; xt < XT; xt++) {
        MI355_SCHED_BARRIER();
        uint8_t *const out = pl ? tile_b : tile;
        uint32_t tlo[YT + 1], thi[YT + 1];          /* [row tile]; row tile YT where the window has none: what an unfiltered tile pairs its last rows with */
#pragma unroll
        for (int rt = 0; rt < YT + 1; rt++) {
            tlo[rt] = CF_SIGN4; thi[rt] = 0u;
            if (rt < RT) {
                /* (the second window's last rows lie past the wave's window: what is read there — the other byte plane, the next wave's window, past the workgroup's
                 * LDS: zeros — only meets taps that are zero; the emulator's arrays have no such neighbourhood: it reads the window's last row instead) */
#ifdef MI355_HIP_EMU_H
                const int wrow = 16 * rt + j + pl * CF_WIN_SECOND < CF_WIN_ROWS ? 16 * rt + j + pl * CF_WIN_SECOND : CF_WIN_ROWS - 1;
#else
                const int wrow = 16 * rt + j + pl * CF_WIN_SECOND;
#endif
                const int a = wrow * CF_WIN_PITCH + 16 * xt + 8 * g;
                int l[4], v[4];
                cf_mfma(cf_lds64(w.lo + a), toep_h, ch4, l);
                if (WIDE) {
                    int h[4];
                    cf_mfma(cf_lds64(w.hi + a), toep_h, zero4, h);
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = ((h[t] << 8) + l[t]) >> ph.shift;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = l[t] >> ph.shift;
                }
                cf_split4(v, tlo[rt], thi[rt]);
                tlo[rt] ^= CF_SIGN4;
            }
        }
#pragma unroll
        for (int yt = 0; yt < YT; yt++) {
            int h[4], l[4], s[4];
            cf_mfma(cf_u64(thi[yt], thi[yt + 1]), toep_v, zero4, h);
            cf_mfma(cf_u64(tlo[yt], tlo[yt + 1]), toep_v, cv4, l);
#pragma unroll
            for (int t = 0; t < 4; t++) s[t] = med3i(((h[t] << 8) + l[t]) >> shv, 0, maxv);
            uint8_t *p = out + (16 * yt + j) * pitch + ((16 * xt + 4 * g) << (WIDE ? 1 : 0));
            if (WIDE) *reinterpret_cast<uint2 *>(p) = make_uint2((uint32_t)s[0] | ((uint32_t)s[1] << 16), (uint32_t)s[2] | ((uint32_t)s[3] << 16));
            else *reinterpret_cast<uint32_t *>(p) = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24);
        }
    }
    MI355_WAVE_SYNC();      /* the window may be overwritten by the wave's next tile */
}
/* tw, th: 16 or 32 */
template <bool WIDE>
__device__ __forceinline__ void cf_mc_tile(CfWin &w, const uint8_t *src, ptrdiff_t sb, int tw, int th, const CfPass ph, const CfPass pv, int bd,
                                           uint8_t *tile, int pitch, int lane)
{
    const bool more = ((th + pv.ext + 15) >> 4) > (th >> 4);         /* the vertical taps reach into one more 16-row tile of the window */
    if (tw == 32 && th == 32) { if (more) cf_mc_tile_n<WIDE, 2, 2, 3>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); else cf_mc_tile_n<WIDE, 2, 2, 2>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); }
    else if (tw == 16 && th == 16) { if (more) cf_mc_tile_n<WIDE, 1, 1, 2>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); else cf_mc_tile_n<WIDE, 1, 1, 1>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); }
    else if (tw == 32) { if (more) cf_mc_tile_n<WIDE, 2, 1, 2>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); else cf_mc_tile_n<WIDE, 2, 1, 1>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); }
    else { if (more) cf_mc_tile_n<WIDE, 1, 2, 3>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); else cf_mc_tile_n<WIDE, 1, 2, 2>(w, src, src, sb, ph, pv, bd, tile, tile, pitch, lane); }
}
/* the 16x16 tiles of both chroma planes of a block at once (windows at the same offset inside their first pieces) */
template <bool WIDE>
__device__ __forceinline__ void cf_mc_tile_pair(CfWin &w, const uint8_t *src, const uint8_t *src_b, ptrdiff_t sb, const CfPass ph, const CfPass pv, int bd,
                                                uint8_t *tile, uint8_t *tile_b, int pitch, int lane)
{
    if (pv.ext) cf_mc_tile_n<WIDE, 1, 1, 2, 2>(w, src, src_b, sb, ph, pv, bd, tile, tile_b, pitch, lane);
    else cf_mc_tile_n<WIDE, 1, 1, 1, 2>(w, src, src_b, sb, ph, pv, bd, tile, tile_b, pitch, lane);
}

#endif
