/*
 * hevc_batch_dev.h — the per-job bodies that more than one kernel file runs: a transform unit with its residual add
 * (hevc_residual_run) and a fused motion-compensation + prediction block (hevc_mcpred_taps).  hevc_batch.hip: one job per
 * wavefront-sized workgroup, destinations in the picture; hevc_ctb.hip: the jobs of one coding tree block by the waves of one
 * workgroup, destinations in the block's LDS tile (DST_GLOBAL = false: the destination pointer is used as it comes).
 * Included after hevc_dev.h, inside the including file's anonymous namespace.
 */
/* ---- transform units: two per wavefront (a 32-point transform occupies 32 lanes) ---------------------- */
/* one transform unit on a half wave (`half`, lanes hl = 0..31); `on`: this half has a unit.  Every lane of the wave walks through every barrier. */
template <bool DST_GLOBAL = true>
__device__ __forceinline__ void hevc_residual_run(IdctScratch &s, mi355_hevc_tu_job j, const bool on, const int half, const int hl, const int bd)
{
    j.coeffs = mi355_global_v(j.coeffs);
    if (DST_GLOBAL) j.dst = mi355_global_v(j.dst);
    const int size = 1 << j.log2_size, cnt = size * size;
    int16_t *c = s.c[half];
    /* coefficients -> LDS: eight per lane and access where the block allows it (16-byte aligned, 8x8 and larger), two otherwise */
    if (on) {
        /* an inverse DCT's coefficients lie in rows 0 .. col_limit + 3 of its block (what the first pass's pruning relies on, hevc_idct_half;
         * hevcdsp_template.c:208-236 — the reference's even part still READS rows below that (every second row of a 16x16 block, every fourth of a 32x32
         * one), which hold zeros in every block the decoder hands over: with the diagonal scan no coefficient lies that low (hevcdec.c:1178-1196)): those
         * rows are not fetched but zeroed in LDS — half of a 32x32 block's 2 KB at col_limit 12.  A precondition of this entry point (include/mi355_hevc_batch.h),
         * not of c->idct[] in general: tests/hevc_batch.py draws col_limit from the block's last coefficient, as the decoder does */
        const int rows = j.kind == MI355_HEVC_TU_IDCT && j.col_limit + 4 < size ? j.col_limit + 4 : size, ncopy = rows << j.log2_size;
        if (cnt >= 64 && (reinterpret_cast<uintptr_t>(j.coeffs) & 15) == 0) {
            for (int i = hl; i < cnt / 8; i += 32) reinterpret_cast<uint4 *>(c)[i] = i < ncopy / 8 ? reinterpret_cast<const uint4 *>(j.coeffs)[i] : make_uint4(0u, 0u, 0u, 0u);
        } else {
            for (int i = hl; i < cnt / 2; i += 32) reinterpret_cast<uint32_t *>(c)[i] = i < ncopy / 2 ? reinterpret_cast<const uint32_t *>(j.coeffs)[i] : 0u;
        }
    }
    MI355_HEVC_SYNC();
    /* every lane walks through every kind's barriers; `on && kind` selects who works */
    {
        const bool k = on && j.kind == MI355_HEVC_TU_IDCT_DC;      /* hevcdsp_template.c:238-252 */
        const int shift = 14 - bd, add = 1 << (shift - 1);
        const int v = (((c[0] + 1) >> 1) + add) >> shift;
        MI355_HEVC_SYNC();
        if (k) for (int i = hl; i < cnt; i += 32) c[i] = (int16_t)v;
        MI355_HEVC_SYNC();
    }
    if (on && j.kind == MI355_HEVC_TU_SKIP) {                      /* :84-98, 4x4 only */
        const int shift = 13 - bd, off = 1 << (shift - 1);
        if (hl < 16) c[hl] = (int16_t)((c[hl] + off) >> shift);
    }
    MI355_HEVC_SYNC();
    hevc_dst4_wave(c, bd, hl, on && j.kind == MI355_HEVC_TU_DST4);
    const bool tr = on && j.kind == MI355_HEVC_TU_IDCT;
    hevc_idct_half<4>(c, hl, tr && j.log2_size == 2, j.col_limit, bd);
    hevc_idct_half<8>(c, hl, tr && j.log2_size == 3, j.col_limit, bd);
    hevc_idct_half<16>(c, hl, tr && j.log2_size == 4, j.col_limit, bd);
    hevc_idct_half<32>(c, hl, tr && j.log2_size == 5, j.col_limit, bd);
    if (!on) return;
    if (!j.dst) {
        for (int i = hl; i < cnt / 2; i += 32) reinterpret_cast<uint32_t *>(j.coeffs)[i] = reinterpret_cast<const uint32_t *>(c)[i];
        return;
    }
    /* add_residual (:51-82): two samples per lane and step, both in one register (the residuals are neighbours in LDS: one
     * dword; sample + residual saturates at int16 and is clipped to the sample range after — the same value as the reference's
     * clip of the int sum); PCM blocks store their samples instead.  Sizes are powers of two: rows by shifts. */
    const int lhw = j.log2_size - 1, hw = 1 << lhw, maxv = (1 << bd) - 1;
    const uint32_t keep = j.kind == MI355_HEVC_TU_PCM ? 0u : 0xFFFFFFFFu;
    const uint32_t *cw = reinterpret_cast<const uint32_t *>(c);
    if (bd > 8 && size >= 8 && ((reinterpret_cast<uintptr_t>(j.dst) | (uintptr_t)j.dst_stride) & 7) == 0) {
        /* four samples per lane and step (8 bytes each way) */
        const int lqw = j.log2_size - 2, qw = 1 << lqw;
        for (int i = hl; i < cnt / 4; i += 32) {
            const int y = i >> lqw, xq = i & (qw - 1);
            uint2 *p = reinterpret_cast<uint2 *>(j.dst + (size_t)y * j.dst_stride) + xq;
            const uint2 v = *p, r = *reinterpret_cast<const uint2 *>(cw + 2 * i);
            *p = make_uint2(pk_clip_max(pk_adds(v.x & keep, r.x), maxv), pk_clip_max(pk_adds(v.y & keep, r.y), maxv));
        }
        return;
    }
    for (int i = hl; i < cnt / 2; i += 32) {
        const int y = i >> lhw, xp = i & (hw - 1);
        uint8_t *row = j.dst + (size_t)y * j.dst_stride;
        const uint32_t r = cw[i];
        if (bd > 8) {
            uint32_t *p = reinterpret_cast<uint32_t *>(row) + xp;
            *p = pk_clip_max(pk_adds(*p & keep, r), maxv);
        } else {
            uint16_t *p = reinterpret_cast<uint16_t *>(row) + xp;
            const uint32_t v = mi355_widen_lo((uint32_t)*p & keep);                /* two bytes -> two 16-bit values */
            const uint32_t o = pk_clip_max(pk_adds(v, r), maxv);
            *p = (uint16_t)((o & 0xFFu) | ((o >> 8) & 0xFF00u));
        }
    }
}

/* ---- MC + prediction fused: the 14-bit intermediate never leaves the CU ------------------------------------------- */
template <int KIND>
struct HevcMcToSamples {      /* results (and, for the two-reference kinds, the kept tile of reference 1) -> samples */
    uint8_t *dst; int stride, bd, amode;                     /* stride in bytes; amode: bytes of guaranteed alignment of a 4-sample segment */
    HevcPredParams p; const int16_t *other;
    __device__ __forceinline__ int px(int a, int r, int x) const
    {
        HevcPredParams q = p;
        q.mode = KIND;                                       /* compile-time kind: hevc_pred_px's dispatch folds away */
        return hevc_pred_px(q, a, (KIND & 1) ? other[r * HEVC_MC_KEEP_PITCH + x] : 0, bd);
    }
    __device__ __forceinline__ void put2(int r, int x, uint32_t v) const
    {
        uint8_t *d = dst + (ptrdiff_t)r * stride;
        if (KIND == 0 && ((bd > 8 && amode >= 4) || (bd <= 8 && amode >= 2))) {
            /* put_unweighted_pred (hevcdsp_template.c:1092-1113: (a + (1 << (shift - 1))) >> shift, clipped) on the pair: the rounding add saturates at int16,
             * where the reference's int sum is beyond the sample range on the same side anyway */
            const int shift = 14 - bd, rnd = 1 << (shift - 1);
            const uint32_t o = pk_clip_max(pk_ashr(pk_adds(v, (uint32_t)rnd * 0x00010001u), shift), (1 << bd) - 1);
            if (bd > 8) reinterpret_cast<uint32_t *>(d)[x >> 1] = o;
            else reinterpret_cast<uint16_t *>(d)[x >> 1] = (uint16_t)((o & 0xFFu) | ((o >> 8) & 0xFF00u));
            return;
        }
        const int v0 = px((int16_t)(v & 0xFFFF), r, x), v1 = px((int16_t)(v >> 16), r, x + 1);
        if (bd > 8) {
            if (amode >= 4) reinterpret_cast<uint32_t *>(d)[x >> 1] = (uint32_t)v0 | ((uint32_t)v1 << 16);
            else { reinterpret_cast<uint16_t *>(d)[x] = (uint16_t)v0; reinterpret_cast<uint16_t *>(d)[x + 1] = (uint16_t)v1; }
        } else {
            if (amode >= 2) reinterpret_cast<uint16_t *>(d)[x >> 1] = (uint16_t)(v0 | (v1 << 8));
            else { d[x] = (uint8_t)v0; d[x + 1] = (uint8_t)v1; }
        }
    }
    __device__ __forceinline__ void put4(int r, int x0, uint32_t lo, uint32_t hi, int n) const
    {
        if (n >= 2) put2(r, x0, lo);
        else if (n == 1) stpx(dst + (ptrdiff_t)r * stride, x0, px((int16_t)(lo & 0xFFFF), r, x0), bd);
        if (n >= 4) put2(r, x0 + 2, hi);
        else if (n == 3) stpx(dst + (ptrdiff_t)r * stride, x0 + 2, px((int16_t)(hi & 0xFFFF), r, x0 + 2), bd);
    }
};
template <int TAPS, int KIND, bool DST_GLOBAL = true>
__device__ inline void hevc_mcpred_taps(const mi355_hevc_mcpred_job &j, int bd, HevcMcScratch &s, int16_t *keep)
{
    const int px = bd > 8 ? 2 : 1;
    constexpr int before = TAPS == 8 ? 3 : 1;
    constexpr bool two = (KIND & 1) != 0;
    const bool pair = TAPS == 4 && j.chroma == 2;        /* both chroma planes of the block (unweighted kinds) */
    const uint8_t *src0 = mi355_global(j.src0), *src1 = two ? mi355_global(j.src1) : nullptr;
    uint8_t *dst = DST_GLOBAL ? mi355_global(j.dst) : j.dst;
    const uint8_t *src0b = pair ? mi355_global(j.src0_b) : nullptr, *src1b = pair && two ? mi355_global(j.src1_b) : nullptr;
    uint8_t *dstb = pair ? (DST_GLOBAL ? mi355_global(j.dst_b) : j.dst_b) : nullptr;
    const unsigned al = (unsigned)(uintptr_t)dst | (unsigned)j.dst_stride | (pair ? (unsigned)(uintptr_t)dstb : 0u);
    const int amode = bd > 8 ? ((al & 3) == 0 ? 4 : 2) : ((al & 1) == 0 ? 2 : 1);
    const HevcPredParams pp{ j.kind, j.denom, j.w0, j.w1, j.o0, j.o1 };
    if (pair && KIND == 0) {
        /* one reference: the planes share every pass of a tile (hevc_mc_tile_pair, tiles of HEVC_MC_PAIR_TILE_H rows) */
        const int bx = j.mx0 ? before : 0, by = j.my0 ? before : 0;
        for (int ty = 0; ty < j.height; ty += HEVC_MC_PAIR_TILE_H)
        for (int tx = 0; tx < j.width; tx += HEVC_MC_TILE) {
            const int tw = j.width - tx < HEVC_MC_TILE ? j.width - tx : HEVC_MC_TILE, th = j.height - ty < HEVC_MC_PAIR_TILE_H ? j.height - ty : HEVC_MC_PAIR_TILE_H;
            const ptrdiff_t so = (ptrdiff_t)(ty - by) * j.src0_stride + (ptrdiff_t)(tx - bx) * px, dof = (ptrdiff_t)ty * j.dst_stride + (ptrdiff_t)tx * px;
            const HevcMcToSamples<KIND> sa{ dst + dof, j.dst_stride, bd, amode, pp, nullptr }, sb{ dstb + dof, j.dst_stride, bd, amode, pp, nullptr };
            hevc_mc_tile_pair(sa, sb, src0 + so, src0b + so, j.src0_stride, tw, th, j.mx0, j.my0, bd, s);
        }
        return;
    }
    /* two references: tiles of 16 rows, so that windows, first-pass results (23 rows each) and the kept tile share the scratch */
    constexpr int TH = two ? HEVC_MC_BI_TILE_H : HEVC_MC_TILE_H;
    for (int plane = 0; plane < (pair ? 2 : 1); plane++) {           /* a pair with two references: plane by plane */
        const uint8_t *p0 = plane ? src0b : src0, *p1 = plane ? src1b : src1;
        uint8_t *pd = plane ? dstb : dst;
        for (int ty = 0; ty < j.height; ty += TH)
        for (int tx = 0; tx < j.width; tx += HEVC_MC_TILE) {
            const int tw = j.width - tx < HEVC_MC_TILE ? j.width - tx : HEVC_MC_TILE, th = j.height - ty < TH ? j.height - ty : TH;
            if (two) {
                const int bx = j.mx1 ? before : 0, by = j.my1 ? before : 0;
                hevc_mc_tile<TAPS>(HevcMcToTile{ keep }, p1 + (ptrdiff_t)(ty - by) * j.src1_stride + (ptrdiff_t)(tx - bx) * px, j.src1_stride, tw, th, j.mx1, j.my1, bd, s);
            }
            const int bx = j.mx0 ? before : 0, by = j.my0 ? before : 0;
            const HevcMcToSamples<KIND> sink{ pd + (ptrdiff_t)ty * j.dst_stride + (ptrdiff_t)tx * px, j.dst_stride, bd, amode, pp, two ? keep : nullptr };
            hevc_mc_tile<TAPS>(sink, p0 + (ptrdiff_t)(ty - by) * j.src0_stride + (ptrdiff_t)(tx - bx) * px, j.src0_stride, tw, th, j.mx0, j.my0, bd, s);
        }
    }
}
