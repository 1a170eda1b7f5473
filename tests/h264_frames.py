"""Synthetic H.264 macroblock records for the Tier-2 frame path
and ctypes/numpy mirrors of include/mi355_h264_frame.h.

`synth_frames()` builds well-formed pictures: every field is what the reference's
slice decoder would hold when it calls ff_h264_hl_decode_mb() (mb_type bits, mv/ref per
4x4 block with partition replication, dequantised transposed coefficients, nnz masks,
availability masks and remapped intra modes).  Inputs come from splitmix64 only.
"""
import ctypes as C
import os

import numpy as np

from rng import SplitMix64

MB_DT = np.dtype([
    ("mb_type", "<u4"), ("nnz_mask", "<u4"), ("cbp", "<u2"), ("qp", "i1"), ("flags", "u1"),
    ("alpha", "i1"), ("beta", "i1"), ("i16mode", "u1"), ("chroma_mode", "u1"),
    ("topleft", "<u2"), ("topright", "<u2"), ("sub", "u1", 4), ("ref_idx", "i1", (2, 4)),
    ("dc_qmul", "<u4", 3), ("slice_id", "u1"), ("intra_level", "u1"), ("qpc", "u1", 2),
    ("i4mode", "i1", 16)])   # i4mode doubles as inter.ref_pic[2][4] (+8 reserved) for inter MBs
assert MB_DT.itemsize == 64

MAX_REFS, MAX_SLOTS = 16, 32
SLICE_DT = np.dtype([
    ("use_weight", "u1"), ("use_weight_chroma", "u1"), ("luma_denom", "u1"), ("chroma_denom", "u1"),
    ("list_count", "u1"), ("rsv", "u1", 3), ("ref_slot", "u1", (2, MAX_REFS)),
    ("luma_weight", "<i2", (MAX_REFS, 2, 2)), ("chroma_weight", "<i2", (MAX_REFS, 2, 2, 2)),
    ("implicit_weight", "<i2", (MAX_REFS, MAX_REFS)), ("chroma_qp_table", "u1", (2, 52)),
    ("implicit_weight_field", "<i2", (2, 2 * MAX_REFS, 2 * MAX_REFS))])
assert SLICE_DT.itemsize == 1040 + 4096


def as_slices(a):
    """slice records of any age -> SLICE_DT (the committed fixtures predate implicit_weight_field: their records are the first 1040 bytes)"""
    a = np.asarray(a)
    if a.dtype == SLICE_DT:
        return a
    out = np.zeros(a.shape if a.dtype.names else (a.size // 1040,), SLICE_DT)
    if a.dtype.names:
        for k in a.dtype.names:
            out[k] = a[k]
    else:
        raw = a.view(np.uint8).reshape(-1, 1040)
        out.view(np.uint8).reshape(-1, SLICE_DT.itemsize)[:, :1040] = raw
    return out


class Frame(C.Structure):
    _fields_ = [("mb_width", C.c_int32), ("mb_height", C.c_int32),
                ("dst", C.c_void_p * 3), ("dst_stride", C.c_int32 * 2),
                ("recon", C.c_void_p * 3), ("recon_stride", C.c_int32 * 2),
                ("ref", (C.c_void_p * 3) * MAX_SLOTS),
                ("mb", C.c_void_p), ("mv", C.c_void_p * 2), ("coef", C.c_void_p),
                ("slices", C.c_void_p), ("nslices", C.c_int32), ("max_intra_level", C.c_int32),
                ("intra_list", C.c_void_p), ("intra_level_start", C.c_void_p),
                ("max_level_width", C.c_int32), ("reserved", C.c_int32),     # reserved = field_picture
                ("surface_layout", C.c_int32), ("flags", C.c_int32)]


assert C.sizeof(Frame) == 920


def tile_planes(y, cb, cr):
    """planes (H, W), (H/2, W/2) x 2 -> macroblock-tiled surfaces (include/mi355_h264_frame.h): luma tiles (256 B per
    macroblock, raster order), chroma tiles (128 B per macroblock: 8 rows of Cb, 8 rows of Cr) as flat uint8 arrays"""
    h, w = y.shape
    mh, mw = h // 16, w // 16
    ty = y.reshape(mh, 16, mw, 16).transpose(0, 2, 1, 3)
    tc = np.concatenate([cb.reshape(mh, 8, mw, 8).transpose(0, 2, 1, 3), cr.reshape(mh, 8, mw, 8).transpose(0, 2, 1, 3)], axis=2)
    return np.ascontiguousarray(ty).reshape(-1), np.ascontiguousarray(tc).reshape(-1)


def untile_planes(ty, tc, mw, mh):
    """inverse of tile_planes (leading batch dimensions are kept)"""
    lead = ty.shape[:-1]
    y = ty.reshape(lead + (mh, mw, 16, 16)).swapaxes(-3, -2).reshape(lead + (16 * mh, 16 * mw))
    c = tc.reshape(lead + (mh, mw, 2, 8, 8))
    cb = c[..., 0, :, :].swapaxes(-3, -2).reshape(lead + (8 * mh, 8 * mw))
    cr = c[..., 1, :, :].swapaxes(-3, -2).reshape(lead + (8 * mh, 8 * mw))
    return [y, cb, cr]

# mb_type bits
I4, I16, PCM, T16x16, T16x8, T8x16, T8x8 = 1, 2, 4, 8, 16, 32, 64
P0L0, P1L0, P0L1, P1L1, DCT8 = 0x1000, 0x2000, 0x4000, 0x8000, 0x01000000
F_LEFT, F_TOP, F_NODB, F_WEIGHTED = 1, 2, 4, 8
ZIGZAG4 = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
# the standard's chroma QP mapping (Table 8-15) for chroma_qp_index_offset = 0
CHROMA_QP = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
DQ0 = [10, 11, 13, 14, 16, 18]


def blk_xy(i):
    return (i & 1) + 2 * ((i >> 2) & 1), ((i >> 1) & 1) + 2 * (i >> 3)


def blk_index(x4, y4):
    return (x4 & 1) + 2 * (y4 & 1) + 4 * (x4 >> 1) + 8 * (y4 >> 1)


def luma_dc_slot(k):
    return 16 * ([0, 2, 8, 10][k >> 2] + [0, 1, 4, 5][k & 3])


def dc_qmul(qp):
    return (DQ0[qp % 6] * 16) << (qp // 6 + 2)


class FrameSet:
    """F independent pictures of identical geometry, host side (numpy)."""

    def __init__(self, nframes, mb_w, mb_h, nrefs):
        self.F, self.mb_w, self.mb_h, self.nrefs = nframes, mb_w, mb_h, nrefs
        nmb = mb_w * mb_h
        self.W, self.H = 16 * mb_w, 16 * mb_h
        self.mb = np.zeros((nframes, nmb), MB_DT)
        self.mv = np.zeros((2, nframes, nmb, 16, 2), np.int16)
        self.coef = np.zeros((nframes, nmb, 384), np.int16)
        self.slices = np.zeros((nframes, 1), SLICE_DT)
        self.refs = [[None] * nrefs for _ in range(nframes)]    # [f][slot] -> (Y, Cb, Cr)
        self.use_l1 = False
        self.max_intra_level = 0
        self.max_level_width = 0
        self.intra_list = [None] * nframes       # per frame: uint32 MB indices sorted by level
        self.intra_start = [None] * nframes      # per frame: int32 offsets [max_level + 1]

    def planes(self):
        return [np.zeros((self.F, self.H, self.W), np.uint8), np.zeros((self.F, self.H // 2, self.W // 2), np.uint8),
                np.zeros((self.F, self.H // 2, self.W // 2), np.uint8)]


COEF_B = [24]   # Laplace scale of the AC levels (SURVEY.md §8d uses 24)
COEF_CLIP = [2047]   # magnitude bound of the levels (a case may raise it to reach the transform's 16-bit wrap)


def _gen_block_coefs(r, n, kind):
    """n 4x4 blocks -> (coefs[n,16], coded[n]).  §8d: coded w.p. 0.5; DC-only w.p. 0.25 else k~U{1..16}
    non-zeros at the first k zig-zag positions, Laplace(24) clipped to +-2047."""
    coded = r.uniform(n) < (0.5 if kind != "dense" else 0.9)
    dconly = r.uniform(n) < 0.25
    k = r.randint(1, 16, n)
    k = np.where(dconly, 1, k)
    vals = r.laplace_int(COEF_B[0], (n, 16), COEF_CLIP[0])
    vals = np.where(vals == 0, 1, vals)
    zz = np.array(ZIGZAG4)
    out = np.zeros((n, 16), np.int16)
    pos_rank = np.empty(16, np.int64)
    pos_rank[zz] = np.arange(16)
    keep = (pos_rank[None, :] < k[:, None]) & coded[:, None]
    out[keep] = vals[keep].astype(np.int16)
    return out, coded


def _avail_masks(top, left, topleft, topright):
    tl, tr = 0xFFFF, 0xEEEA
    if not top:
        tl, tr = 0xB3FF, 0x26EA
    if not left:
        tl &= 0xDF5F
    if not topleft:
        tl &= 0x7FFF
    if not topright:
        tr &= 0xFBFF
    return tl, tr


def _ref_plane(r, h, w, kind):
    if kind == "noise":
        return r.u8((h, w))
    # smooth: gentle gradient + small noise, so the loop-filter thresholds are actually met
    yy, xx = np.mgrid[0:h, 0:w]
    base = r.randint(60, 180)
    a = base + (xx * r.randint(-3, 3)) // 8 + (yy * r.randint(-3, 3)) // 8 + r.randint(-3, 3, (h, w))
    return np.clip(a, 0, 255).astype(np.uint8)


def synth_frames(nframes, mb_w, mb_h, seed=0x264, nrefs=4, mix="p16", intra_frac=0.0, bframes=False,
                 weighted=0, dct8_frac=0.0, mv_range=64, offsets=False, pcm_frac=0.0, refs="noise", coef_b=24, coef_clip=2047):
    """mix: 'p16' (all 16x16), 'mixed' (all partition shapes).  Returns a FrameSet."""
    fs = FrameSet(nframes, mb_w, mb_h, nrefs)
    r = SplitMix64(seed)
    nmb = mb_w * mb_h
    fs.use_l1 = bframes
    COEF_B[0] = coef_b
    COEF_CLIP[0] = coef_clip
    for f in range(nframes):
        for s in range(nrefs):
            fs.refs[f][s] = (_ref_plane(r, fs.H, fs.W, refs), _ref_plane(r, fs.H // 2, fs.W // 2, refs),
                             _ref_plane(r, fs.H // 2, fs.W // 2, refs))
        sl = fs.slices[f, 0]
        sl["list_count"] = 2 if bframes else 1
        sl["ref_slot"][0, :nrefs] = np.arange(nrefs)
        sl["ref_slot"][1, :nrefs] = np.arange(nrefs)[::-1]
        sl["chroma_qp_table"][0] = CHROMA_QP
        sl["chroma_qp_table"][1] = CHROMA_QP
        sl["use_weight"] = weighted
        if weighted:
            sl["use_weight_chroma"] = 1
            sl["luma_denom"], sl["chroma_denom"] = r.randint(0, 7), r.randint(0, 7)
            sl["luma_weight"] = r.randint(-128, 127, (MAX_REFS, 2, 2))
            sl["chroma_weight"] = r.randint(-128, 127, (MAX_REFS, 2, 2, 2))
            iw = r.randint(-64, 128, (MAX_REFS, MAX_REFS))
            iw[r.randint(0, 1, (MAX_REFS, MAX_REFS)) == 1] = 32
            sl["implicit_weight"] = iw
        is_intra = r.uniform(nmb) < intra_frac
        a_off = 2 * r.randint(-3, 3) if offsets else 0
        b_off = 2 * r.randint(-3, 3) if offsets else 0
        for m in range(nmb):
            mx, my = m % mb_w, m // mb_w
            rec = fs.mb[f, m]
            rec["qp"] = qp = r.randint(20, 40)
            rec["alpha"], rec["beta"] = a_off, b_off
            rec["flags"] = (F_LEFT if mx else 0) | (F_TOP if my else 0) | (F_WEIGHTED if weighted else 0)
            qpc = CHROMA_QP[qp]
            intra = bool(is_intra[m])
            rec["dc_qmul"] = (dc_qmul(qp), dc_qmul(qpc), dc_qmul(qpc))
            rec["ref_idx"] = -1
            t = 0
            use8 = False
            if intra:
                top, left = my > 0, mx > 0
                tl, tr = _avail_masks(top, left, top and left, top and mx + 1 < mb_w)
                rec["topleft"], rec["topright"] = tl, tr
                sel = r.uniform()
                both = top and left
                c16 = [0, 1, 2, 3] if both else ([1, 4] if left else ([2, 5] if top else [6]))
                rec["chroma_mode"] = c16[r.randint(0, len(c16) - 1)]
                if sel < pcm_frac:
                    t = PCM
                elif sel < 0.4:
                    t = I16
                    rec["i16mode"] = c16[r.randint(0, len(c16) - 1)]
                else:
                    t = I4
                    use8 = r.uniform() < 0.4
                    if use8:
                        t |= DCT8
                    for i in (range(0, 16, 4) if use8 else range(16)):
                        x4, y4 = blk_xy(i)
                        btop = top or y4 > 0
                        bleft = left or x4 > 0
                        btl = bool((tl << i) & 0x8000)
                        if btop and bleft:
                            cand = [0, 1, 2, 3, 7, 8] + ([4, 5, 6] if btl else [])
                        elif bleft:
                            cand = [1, 8, 9]
                        elif btop:
                            cand = [0, 3, 7, 10]
                        else:
                            cand = [11]
                        rec["i4mode"][i] = cand[r.randint(0, len(cand) - 1)]
            else:
                shape = 0 if mix == "p16" else r.randint(0, 3)
                nl = 2 if bframes else 1

                def pick_dir():
                    return 1 if not bframes else r.randint(1, 3)     # bit0 L0, bit1 L1

                def set_part(x4, y4, w4, h4, d, quadrants):
                    for l in range(nl):
                        if not (d >> l) & 1:
                            continue
                        ref = r.randint(0, nrefs - 1)
                        mv = r.randint(-mv_range, mv_range - 1, 2)
                        for q in quadrants:
                            rec["ref_idx"][l][q] = ref
                        for yy in range(y4, y4 + h4):
                            for xx in range(x4, x4 + w4):
                                fs.mv[l, f, m, xx + 4 * yy] = mv
                if shape == 0:
                    d = pick_dir()
                    t = T16x16 | (P0L0 if d & 1 else 0) | (P0L1 if d & 2 else 0)
                    set_part(0, 0, 4, 4, d, [0, 1, 2, 3])
                elif shape == 1:
                    d0, d1 = pick_dir(), pick_dir()
                    t = T16x8 | (P0L0 if d0 & 1 else 0) | (P0L1 if d0 & 2 else 0) | (P1L0 if d1 & 1 else 0) | (P1L1 if d1 & 2 else 0)
                    set_part(0, 0, 4, 2, d0, [0, 1])
                    set_part(0, 2, 4, 2, d1, [2, 3])
                elif shape == 2:
                    d0, d1 = pick_dir(), pick_dir()
                    t = T8x16 | (P0L0 if d0 & 1 else 0) | (P0L1 if d0 & 2 else 0) | (P1L0 if d1 & 1 else 0) | (P1L1 if d1 & 2 else 0)
                    set_part(0, 0, 2, 4, d0, [0, 2])
                    set_part(2, 0, 2, 4, d1, [1, 3])
                else:
                    t = T8x8
                    anyl = [0, 0]
                    for q in range(4):
                        d = pick_dir()
                        sub = r.randint(0, 3)
                        rec["sub"][q] = sub | (0x10 if d & 1 else 0) | (0x20 if d & 2 else 0)
                        qx, qy = 2 * (q & 1), 2 * (q >> 1)
                        # all sub-partitions of a quadrant share the reference (one ref_idx per 8x8)
                        refs = [r.randint(0, nrefs - 1) for _ in range(nl)]
                        parts = {0: [(0, 0, 2, 2)], 1: [(0, 0, 2, 1), (0, 1, 2, 1)], 2: [(0, 0, 1, 2), (1, 0, 1, 2)],
                                 3: [(0, 0, 1, 1), (1, 0, 1, 1), (0, 1, 1, 1), (1, 1, 1, 1)]}[sub]
                        for l in range(nl):
                            if not (d >> l) & 1:
                                continue
                            anyl[l] = 1
                            rec["ref_idx"][l][q] = refs[l]
                            for (px, py, pw, ph) in parts:
                                mv = r.randint(-mv_range, mv_range - 1, 2)
                                for yy in range(qy + py, qy + py + ph):
                                    for xx in range(qx + px, qx + px + pw):
                                        fs.mv[l, f, m, xx + 4 * yy] = mv
                    t |= (P0L0 | P1L0 if anyl[0] else 0) | (P0L1 | P1L1 if anyl[1] else 0)
                if shape < 3 or all((rec["sub"][q] & 3) == 0 for q in range(4)):
                    use8 = r.uniform() < dct8_frac
                    if use8:
                        t |= DCT8
            rec["mb_type"] = t
            rec["qpc"] = (qpc, qpc)
            if not intra:
                for l in range(2):
                    for q in range(4):
                        ri = int(rec["ref_idx"][l][q])
                        rec["i4mode"][4 * l + q] = np.uint8(sl["ref_slot"][l][ri]).astype(np.int8) if ri >= 0 else -1
            if t & PCM:
                fs.coef[f, m].view(np.uint8)[:384] = r.u8(384)
                rec["qp"] = 0                       # qscale_table of an I_PCM MB (h264_cabac.c / h264_cavlc.c)
                rec["qpc"] = (CHROMA_QP[0], CHROMA_QP[0])   # the record's chroma QPs always follow its qp
                rec["cbp"] = 0x2F
                rec["nnz_mask"] = 0xFFFFFF
                continue
            # residual
            mask = 0
            cf = fs.coef[f, m]
            if use8:
                for q in range(4):
                    if r.uniform() < 0.5:
                        blk = np.zeros(64, np.int16)
                        k = 1 if r.uniform() < 0.25 else r.randint(1, 64)
                        idx = np.argsort(r.uniform(64) + np.arange(64) * 0.08)[:k]
                        v = r.laplace_int(24, k, 2047)
                        blk[idx] = np.where(v == 0, 1, v)
                        if k == 1:
                            blk[:] = 0
                            blk[0] = v[0] if v[0] else 7
                        cf[q * 64:(q + 1) * 64] = blk
                        mask |= 0xF << (4 * q)
            else:
                blks, coded = _gen_block_coefs(r, 16, "sparse")
                if t & I16:
                    blks[:, 0] = 0      # DC travels separately for Intra16x16
                    coded &= (blks != 0).any(axis=1)
                cf[:256] = blks.reshape(-1)
                for i in range(16):
                    if coded[i]:
                        mask |= 1 << i
            if t & I16 and r.uniform() < 0.8:
                lv = r.laplace_int(40, 16, 2047)
                for k in range(16):
                    cf[luma_dc_slot(k)] = lv[k]
                if lv.any():
                    mask |= 1 << 24
            cmode = r.randint(0, 2)
            if cmode:
                dcs = r.laplace_int(30, 8, 2047)
                if cmode == 2:
                    blks, coded = _gen_block_coefs(r, 8, "sparse")
                    blks[:, 0] = 0
                    coded &= (blks != 0).any(axis=1)
                    cf[256:384] = blks.reshape(-1)
                    for j in range(8):
                        if coded[j]:
                            mask |= 1 << (16 + j)
                    if not coded.any():
                        cmode = 1
                for j in range(8):
                    cf[256 + 16 * j] = dcs[j]
                if dcs[:4].any():
                    mask |= 1 << 25
                if dcs[4:].any():
                    mask |= 1 << 26
                if not dcs.any() and cmode == 1:
                    cmode = 0
            cbp = cmode << 4
            for q in range(4):
                if (mask >> (4 * q)) & 0xF:
                    cbp |= 1 << q
            if t & I16 and (mask & 0xFFFF):
                cbp |= 15
            rec["cbp"] = cbp
            rec["nnz_mask"] = mask
        fs.max_intra_level = max(fs.max_intra_level, intra_schedule(fs, f))
    return fs


def intra_schedule(fs, f):
    """python twin of mi355_h264_intra_schedule() (host helper of the C ABI)"""
    lv = intra_levels(fs.mb[f], fs.mb_w, fs.mb_h)
    mx = int(lv.max()) if lv.size else 0
    order = [np.nonzero(lv == l)[0] for l in range(1, mx + 1)]
    fs.intra_list[f] = np.concatenate(order).astype(np.uint32) if order else np.zeros(0, np.uint32)
    fs.intra_start[f] = np.concatenate([[0], np.cumsum([len(o) for o in order])]).astype(np.int32)
    if order:
        fs.max_level_width = max(fs.max_level_width, max(len(o) for o in order))
    note_level_widths(fs, [len(o) for o in order])
    return mx


def note_level_widths(fs, widths):
    """fs.level_widths[l - 1] = the largest number of macroblocks any picture has on level l"""
    cur = getattr(fs, "level_widths", [])
    n = max(len(cur), len(widths))
    cur = list(cur) + [0] * (n - len(cur))
    for i, w in enumerate(widths):
        cur[i] = max(cur[i], int(w))
    fs.level_widths = cur


def intra_levels(mb, mb_w, mb_h):
    """levels in a local int array (an all-intra picture above 1080p exceeds the record's 8-bit field, which only
    receives the saturated value)"""
    lv = np.zeros(mb_w * mb_h, np.int64)
    intra = (mb["mb_type"] & 7) != 0
    for y in range(mb_h):
        for x in range(mb_w):
            xy = x + y * mb_w
            if not intra[xy]:
                continue
            m = 0
            for (dx, dy) in ((-1, 0), (-1, -1), (0, -1), (1, -1)):
                nx, ny = x + dx, y + dy
                if 0 <= nx < mb_w and 0 <= ny < mb_h:
                    m = max(m, int(lv[nx + ny * mb_w]))
            lv[xy] = m + 1
    mb["intra_level"] = np.minimum(lv, 255)
    return lv


def host_frames(fs, recon, dst, px=1, mb=None, coef=None, refs=None):
    """ctypes Frame array whose pointers are HOST addresses into fs / recon / dst (for the oracle).
    px = 2: 16-bit samples (strides in bytes); mb / coef / refs: replacements for fs's own arrays (the High 10 variant of a picture set)"""
    arr = (Frame * fs.F)()
    keep = []
    mb = fs.mb if mb is None else mb
    coef = fs.coef if coef is None else coef
    refs = fs.refs if refs is None else refs
    for f in range(fs.F):
        fr = arr[f]
        fr.mb_width, fr.mb_height = fs.mb_w, fs.mb_h
        for p in range(3):
            fr.dst[p] = dst[p][f].ctypes.data
            fr.recon[p] = recon[p][f].ctypes.data
        fr.dst_stride[0], fr.dst_stride[1] = px * fs.W, px * fs.W // 2
        fr.recon_stride[0], fr.recon_stride[1] = px * fs.W, px * fs.W // 2
        for s in range(fs.nrefs):
            for p in range(3):
                fr.ref[s][p] = refs[f][s][p].ctypes.data
        fr.mb = mb[f].ctypes.data
        fr.mv[0] = fs.mv[0, f].ctypes.data
        fr.mv[1] = fs.mv[1, f].ctypes.data if fs.use_l1 else None
        fr.coef = coef[f].ctypes.data
        fr.slices = fs.slices[f].ctypes.data
        fr.nslices = fs.slices.shape[1]
        fr.max_intra_level = int(fs.intra_start[f].shape[0]) - 1
        fr.intra_list = fs.intra_list[f].ctypes.data
        fr.intra_level_start = fs.intra_start[f].ctypes.data
        fr.max_level_width = fs.max_level_width
        fr.flags = 1 if int(fs.intra_start[f][-1]) == fs.mb_w * fs.mb_h else 0       # MI355_FRAME_NO_INTER: every macroblock is on some intra level
    return arr, keep


def widen_samples(a, sh):
    """an 8-bit plane as DeviceFrames(bit_depth=8 + sh) uploads it: shifted up, the low bits filled from the sample's position"""
    lo = (np.arange(a.shape[1], dtype=np.uint16)[None, :] * 3 + np.arange(a.shape[0], dtype=np.uint16)[:, None] * 5) & ((1 << sh) - 1)
    return np.ascontiguousarray((a.astype(np.uint16) << sh) | lo)


def chroma_422(pl):
    """a 4:2:0 chroma plane (H / 2 rows) as the 4:2:2 variant of the picture set holds it (H rows): every row twice, the second copy moved a little by its position"""
    out = np.repeat(pl.astype(np.int16), 2, axis=0)
    rr, cc = np.arange(out.shape[0])[:, None], np.arange(out.shape[1])[None, :]
    out[1::2] += ((rr[1::2] * 5 + cc * 3) % 7) - 3
    return np.clip(out, 0, 255).astype(np.uint8)


def coefs_422(c):
    """[..., 384] coefficients of 4:2:0 macroblocks (256 luma, Cb blocks 0..3, Cr blocks 0..3) -> [..., 512] of 4:2:2 ones (eight blocks a plane, Cb at 256, Cr at 384:
    include/mi355_h264_frame.h): the plane's own four blocks, then the other plane's four in reverse order with the sign turned"""
    out = np.zeros(c.shape[:-1] + (512,), c.dtype)
    out[..., :256] = c[..., :256]
    cb, cr = c[..., 256:320], c[..., 320:384]
    rev = lambda a: a.reshape(a.shape[:-1] + (4, 16))[..., ::-1, :].reshape(a.shape)
    out[..., 256:320], out[..., 320:384] = cb, -rev(cr)
    out[..., 384:448], out[..., 448:512] = cr, -rev(cb)
    return out


def widen_records(fs, sh, idc=1):
    """records and coefficients as DeviceFrames(bit_depth=8 + sh) uploads them: QPs raised by QpBdOffset, 32-bit coefficients scaled by the shift,
    an I_PCM macroblock's samples one per coefficient slot.  idc 2: the 4:2:2 variant of the set — coefs_422(), an I_PCM macroblock's 8x16 chroma samples
    (its 8x8 ones, then the other plane's), the chroma DC bits of nnz_mask read off the eight DC levels of each plane"""
    mb = fs.mb.copy()
    mb["qp"] += 6 * sh
    mb["qpc"] += 6 * sh
    pcm = (fs.mb["mb_type"] & 4) != 0
    if idc == 1:
        coef = fs.coef.astype(np.int32) << sh
        coef[pcm] = fs.coef[pcm].view(np.uint8)[:, :384].astype(np.int32) << sh
        return mb, np.ascontiguousarray(coef)
    coef = coefs_422(fs.coef.astype(np.int32)) << sh
    if pcm.any():
        b = fs.coef[pcm].view(np.uint8)[:, :384].astype(np.int32)
        coef[pcm] = np.concatenate([b[:, :256], b[:, 256:320], b[:, 320:384], b[:, 320:384], b[:, 256:320][:, ::-1]], axis=1) << sh
    dc_cb, dc_cr = (coef[..., 256:384:16] != 0).any(axis=-1), (coef[..., 384:512:16] != 0).any(axis=-1)
    keep = pcm | ((fs.mb["cbp"] & 0x30) == 0)
    nz = mb["nnz_mask"] & ~np.uint32(3 << 25)
    nz = nz | (dc_cb.astype(np.uint32) << 25) | (dc_cr.astype(np.uint32) << 26)
    mb["nnz_mask"] = np.where(keep, mb["nnz_mask"], nz)
    return mb, np.ascontiguousarray(coef)


def ref_library():
    """oracle/_ref/libref.so — the reference's own DSP objects, built where /root/reference exists (__graft_entry__.build()) and shipped with the tree; None without it"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref.so")
    return C.CDLL(path) if os.path.exists(path) else None


def run_oracle_hbd(oracle, fs, bit_depth, deblock=True, idc=1):
    """The frame-level checker above 8 bits (oracle/oracle_h264frame_hbd.c: the restated per-macroblock drivers calling THE REFERENCE'S OWN tables at
    that bit depth, oracle/_ref/libref.so) on the High 10 variant of a picture set — the pictures DeviceFrames(bit_depth=...) uploads.
    idc 2: the 4:2:2 variant of the set (chroma_422 / coefs_422: chroma planes of the luma's height, eight chroma blocks a plane) against the same drivers
    with the reference's tables initialised for chroma_format_idc 2 (oracle_h264frame_hbd_bind_cf).
    Returns (recon, dst) as uint16 plane lists, or None when libref.so is not there."""
    ref = ref_library()
    if ref is None:
        return None
    sh = bit_depth - 8
    lib = oracle.lib
    fns = [C.cast(getattr(ref, n), C.c_void_p) for n in ("ff_h264dsp_init", "ff_h264qpel_init", "ff_h264chroma_init", "ff_h264_pred_init")]
    lib.oracle_h264frame_hbd_bind_cf.restype = C.c_int
    lib.oracle_h264frame_hbd_bind_cf.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int]
    assert lib.oracle_h264frame_hbd_bind_cf(*fns, bit_depth, idc) == 0
    mb, coef = widen_records(fs, sh, idc)
    c422 = chroma_422 if idc == 2 else (lambda a: a)
    refs = [[tuple(widen_samples(pl if i == 0 else c422(pl), sh) for i, pl in enumerate(fs.refs[f][s_])) for s_ in range(fs.nrefs)] for f in range(fs.F)]
    hc = fs.H if idc == 2 else fs.H // 2
    recon = [np.zeros((fs.F, fs.H, fs.W), np.uint16), np.zeros((fs.F, hc, fs.W // 2), np.uint16), np.zeros((fs.F, hc, fs.W // 2), np.uint16)]
    dst = [np.zeros_like(a) for a in recon]
    arr, _ = host_frames(fs, recon, dst, px=2, mb=mb, coef=coef, refs=refs)
    lib.oracle_h264_recon_frame_hbd.restype = C.c_int
    lib.oracle_h264_deblock_frame_hbd.restype = C.c_int
    for f in range(fs.F):
        assert lib.oracle_h264_recon_frame_hbd(C.byref(arr[f])) == 0
        if deblock:
            assert lib.oracle_h264_deblock_frame_hbd(C.byref(arr[f])) == 0
    return recon, dst


def run_oracle(oracle, fs, deblock=True):
    """Reference-order reconstruction (+ loop filter) on the CPU oracle; returns (recon, dst) plane lists."""
    recon, dst = fs.planes(), fs.planes()
    arr, _ = host_frames(fs, recon, dst)
    lib = oracle.lib
    lib.oracle_h264_recon_frame.restype = None
    lib.oracle_h264_deblock_frame.restype = None
    for f in range(fs.F):
        lib.oracle_h264_recon_frame(C.byref(arr[f]))
        if deblock:
            lib.oracle_h264_deblock_frame(C.byref(arr[f]))
    return recon, dst


class DeviceFrames:
    """Uploads a FrameSet through the C ABI's memory helpers and builds device descriptors.
    `replicate` = total number of pictures F >= fs.F: picture f is a device-side copy of picture
    f % fs.F with its own buffers (bench.py: many independent streams from a few distinct ones)."""

    def __init__(self, prov, fs, replicate=None, pad=0, tiled=False, bit_depth=8, idc=1):
        """bit_depth 9 / 10: the same pictures as a High 10 batch for mi355_h264_decode_frames_wide_dev — 16-bit samples (the 8-bit reference
        samples shifted up, the low bits filled from the sample's position), 32-bit coefficients (scaled by the same shift), QPs raised by
        QpBdOffset; the frame-level checker for these is run_oracle_hbd() (oracle/oracle_h264frame_hbd.c on the reference's own 9 / 10-bit tables).
        pad: extra bytes per luma row (chroma rows get pad // 2): strides that are multiples of 4 but not of
        16 / 8 take the kernels' narrow-access paths.
        tiled: dst / recon / reference surfaces in the macroblock-tiled layout (pad then = extra bytes per macroblock ROW of
        luma tiles, a multiple of 256; chroma rows get half)"""
        self.lib, self.fs, self.pad, self.tiled = prov.lib, fs, pad, tiled
        self.bit_depth, px, sh = bit_depth, (2 if bit_depth > 8 else 1), bit_depth - 8
        self.px, self.idc = px, idc
        assert not (tiled and px == 2) and (idc == 1 or px == 2)      # idc 2: the 4:2:2 variant of the set (run_oracle_hbd(idc=2)), 9 / 10 bit
        hc = self.hc = fs.H if idc == 2 else fs.H // 2               # chroma rows
        lib = self.lib
        lib.mi355_malloc.restype = C.c_void_p
        lib.mi355_malloc.argtypes = [C.c_size_t]
        lib.mi355_free.argtypes = [C.c_void_p]
        lib.mi355_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.mi355_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.mi355_memcpy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.bufs = []
        G = fs.F
        F = self.F = replicate or G
        nmb = fs.mb_w * fs.mb_h
        ys, cs = px * fs.W + pad, px * fs.W // 2 + pad // 2            # strides
        ysz, csz = fs.H * ys, hc * cs
        if tiled:
            assert pad % 256 == 0
            ys, cs = fs.mb_w * 256 + pad, fs.mb_w * 128 + pad // 2   # bytes per macroblock row of tiles
            ysz, csz = fs.mb_h * ys, 0                               # plane 1 (both chroma planes) follows plane 0
            self.tys, self.tcs = ys, cs
        self.fsz = ysz + 2 * csz if not tiled else ysz + fs.mb_h * cs

        def up_rep(a, per):
            """device array of F entries of `per` bytes; first G from the host, rest copied on the device"""
            a = np.ascontiguousarray(a)
            assert a.nbytes == G * per
            p = self.alloc(F * per)
            lib.mi355_memcpy_h2d(p, a.ctypes.data, a.nbytes)
            done = G
            while done < F:            # doubling copies
                n = min(done, F - done)
                lib.mi355_memcpy_d2d(p + done * per, p, n * per)
                done += n
            return p
        mb = fs.mb
        coef, cbytes = fs.coef, 768
        if px == 2:
            (mb, coef), cbytes = widen_records(fs, sh, idc), (2048 if idc == 2 else 1536)
        self.cbytes = cbytes
        self.mb = up_rep(mb, nmb * 64)
        self.mv0 = up_rep(fs.mv[0], nmb * 64)
        self.mv1 = up_rep(fs.mv[1], nmb * 64) if fs.use_l1 else None
        self.coef = up_rep(coef, nmb * cbytes)
        nsl = fs.slices.shape[1]
        self.slices = up_rep(fs.slices, nsl * SLICE_DT.itemsize)
        self.recon = self.alloc(F * self.fsz)
        self.dst = self.alloc(F * self.fsz)
        refs_host = np.empty((G, fs.nrefs, self.fsz), np.uint8)
        for f in range(G):
            for s_ in range(fs.nrefs):
                y, cb, cr = fs.refs[f][s_]
                if tiled:
                    ty, tc = tile_planes(y, cb, cr)
                    refs_host[f, s_, :ysz].reshape(fs.mb_h, ys)[:, :fs.mb_w * 256] = ty.reshape(fs.mb_h, -1)
                    refs_host[f, s_, ysz:].reshape(fs.mb_h, cs)[:, :fs.mb_w * 128] = tc.reshape(fs.mb_h, -1)
                    continue
                if px == 2:
                    def wide(a):
                        return widen_samples(a, sh).view(np.uint8).reshape(a.shape[0], -1)
                    if idc == 2:
                        cb, cr = chroma_422(cb), chroma_422(cr)
                    y, cb, cr = wide(y), wide(cb), wide(cr)
                refs_host[f, s_, :ysz].reshape(fs.H, ys)[:, :px * fs.W] = y
                refs_host[f, s_, ysz:ysz + csz].reshape(hc, cs)[:, :px * fs.W // 2] = cb
                refs_host[f, s_, ysz + csz:].reshape(hc, cs)[:, :px * fs.W // 2] = cr
        self.refs = up_rep(refs_host, fs.nrefs * self.fsz)
        ilist = [self.up(fs.intra_list[g]) if len(fs.intra_list[g]) else None for g in range(G)]
        istart = [self.up(fs.intra_start[g]) for g in range(G)]
        arr = (Frame * F)()
        for f in range(F):
            g = f % G
            fr = arr[f]
            fr.mb_width, fr.mb_height = fs.mb_w, fs.mb_h
            for kind, base0 in (("dst", self.dst), ("recon", self.recon)):
                base = base0 + f * self.fsz
                a = getattr(fr, kind)
                a[0], a[1], a[2] = base, base + ysz, base + ysz + csz
            fr.dst_stride[0], fr.dst_stride[1] = ys, cs
            fr.recon_stride[0], fr.recon_stride[1] = ys, cs
            for s_ in range(fs.nrefs):
                base = self.refs + (f * fs.nrefs + s_) * self.fsz
                fr.ref[s_][0], fr.ref[s_][1], fr.ref[s_][2] = base, base + ysz, base + ysz + csz
            fr.mb = self.mb + f * nmb * 64
            fr.mv[0] = self.mv0 + f * nmb * 64
            fr.mv[1] = (self.mv1 + f * nmb * 64) if self.mv1 else None
            fr.coef = self.coef + f * nmb * cbytes
            fr.slices = self.slices + f * nsl * SLICE_DT.itemsize
            fr.nslices = nsl
            fr.max_intra_level = int(fs.intra_start[g].shape[0]) - 1
            fr.intra_list = ilist[g]          # read-only: shared between the copies
            fr.intra_level_start = istart[g]
            fr.max_level_width = fs.max_level_width
            fr.surface_layout = 1 if tiled else 0
            fr.flags = 1 if int(fs.intra_start[g][-1]) == fs.mb_w * fs.mb_h else 0    # MI355_FRAME_NO_INTER
        self.host_desc = arr
        self.d_desc = self.alloc(C.sizeof(arr))
        self.lib.mi355_memcpy_h2d(self.d_desc, C.addressof(arr), C.sizeof(arr))

    def alloc(self, n):
        p = self.lib.mi355_malloc(n)
        assert p, "device allocation failed"
        self.bufs.append(p)
        return p

    def h2d(self, dptr, a):
        a = np.ascontiguousarray(a)
        self.lib.mi355_memcpy_h2d(dptr, a.ctypes.data, a.nbytes)

    def up(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        self.lib.mi355_memcpy_h2d(p, a.ctypes.data, a.nbytes)
        return p

    def fetch(self, base, first=0, count=None):
        fs = self.fs
        n = count or self.F
        raw = np.empty(n * self.fsz, np.uint8)
        self.lib.mi355_memcpy_d2h(raw.ctypes.data, base + first * self.fsz, raw.nbytes)
        raw = raw.reshape(n, self.fsz)
        if self.tiled:
            ysz = fs.mb_h * self.tys
            ty = raw[:, :ysz].reshape(n, fs.mb_h, self.tys)[:, :, :fs.mb_w * 256].reshape(n, -1)
            tc = raw[:, ysz:].reshape(n, fs.mb_h, self.tcs)[:, :, :fs.mb_w * 128].reshape(n, -1)
            return untile_planes(ty, tc, fs.mb_w, fs.mb_h)
        px = self.px
        ys, cs = px * fs.W + self.pad, px * fs.W // 2 + self.pad // 2
        ysz, csz = fs.H * ys, self.hc * cs
        if px == 2:
            return [np.ascontiguousarray(raw[:, :ysz].reshape(n, fs.H, ys)[:, :, :2 * fs.W]).view(np.uint16),
                    np.ascontiguousarray(raw[:, ysz:ysz + csz].reshape(n, self.hc, cs)[:, :, :fs.W]).view(np.uint16),
                    np.ascontiguousarray(raw[:, ysz + csz:].reshape(n, self.hc, cs)[:, :, :fs.W]).view(np.uint16)]
        return [raw[:, :ysz].reshape(n, fs.H, ys)[:, :, :fs.W], raw[:, ysz:ysz + csz].reshape(n, fs.H // 2, cs)[:, :, :fs.W // 2],
                raw[:, ysz + csz:].reshape(n, fs.H // 2, cs)[:, :, :fs.W // 2]]

    def decode(self, stream=None, per_level=True):
        """per_level: intra passes sized by the per-level widths (mi355_h264_decode_frames_levels_dev); otherwise by the
        single bound max_level_width (mi355_h264_decode_frames_dev)"""
        fs = self.fs
        if per_level:
            lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
            fn = self.lib.mi355_h264_decode_frames_levels_dev
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            rc = fn(self.d_desc, self.F, fs.mb_w, fs.mb_h, fs.max_intra_level, lw, stream)
        else:
            fn = self.lib.mi355_h264_decode_frames_dev
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            rc = fn(self.d_desc, self.F, fs.mb_w, fs.mb_h, fs.max_intra_level, fs.max_level_width, stream)
        assert rc == 0, rc
        self.lib.mi355_sync.restype = C.c_int
        assert self.lib.mi355_sync(stream) == 0

    def decode_by_layout(self):
        """the three passes through the entry points that take the batch's surface layouts (what bench.py, the sessions and the bridge call:
        mi355_h264_recon_inter_layouts_dev / mi355_h264_deblock_layouts_dev — the kernel instances that carry one layout's code alone)"""
        fs, lib = self.fs, self.lib
        mask = 2 if self.tiled else 1
        lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
        for name, args in (("mi355_h264_recon_inter_layouts_dev", (fs.mb_w, fs.mb_h, mask)), ("mi355_h264_recon_intra_all_dev", (fs.mb_w, fs.mb_h, fs.max_intra_level, lw)),
                           ("mi355_h264_deblock_layouts_dev", (fs.mb_w, fs.mb_h, mask))):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int] + [C.c_int if isinstance(a, int) else C.c_void_p for a in args] + [C.c_void_p]
            assert fn(self.d_desc, self.F, *args, None) == 0, name
        assert lib.mi355_sync(None) == 0

    def decode_pipelined(self, shares=3, turns=1, calls=1):
        """mi355_h264_pipelines_*: the batch as `shares` shares on their own streams, reconstruction launches taking turns; `calls` batches one behind the other"""
        fs, lib = self.fs, self.lib
        lib.mi355_h264_pipelines_create.restype = C.c_void_p
        lib.mi355_h264_pipelines_create.argtypes = [C.c_int, C.c_int]
        lib.mi355_h264_pipelines_decode_dev.restype = C.c_int
        lib.mi355_h264_pipelines_decode_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.mi355_h264_pipelines_sync.argtypes = [C.c_void_p]
        lib.mi355_h264_pipelines_destroy.argtypes = [C.c_void_p]
        lib.mi355_h264_pipelines_destroy.restype = None
        lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
        p = lib.mi355_h264_pipelines_create(shares, turns)
        assert p
        try:
            for _ in range(calls):
                assert lib.mi355_h264_pipelines_decode_dev(p, self.d_desc, self.F, fs.mb_w, fs.mb_h, fs.max_intra_level, lw, 2 if self.tiled else 1) == 0
            assert lib.mi355_h264_pipelines_sync(p) == 0
        finally:
            lib.mi355_h264_pipelines_destroy(p)

    def decode_wide(self, bit_depth=8, idc=1, passes=7, sync=True):
        """the three passes of the SECOND kernel set (mi355_h264_decode_frames_wide_dev: High 10 / High 4:2:2, and 8-bit 4:2:0 for this
        comparison) on linear surfaces"""
        fs, lib = self.fs, self.lib
        assert not self.tiled
        lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
        fn = lib.mi355_h264_decode_frames_wide_dev
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        rc = fn(self.d_desc, self.F, fs.mb_w, fs.mb_h, fs.max_intra_level, lw, bit_depth, idc, passes, None)
        assert rc == 0, rc
        if sync:
            assert lib.mi355_sync(None) == 0

    def decode_sparse(self, poison=True):
        """the three passes with mi355_h264_recon_inter_sparse_dev: inter macroblocks whose cbp is zero do not fetch their
        coefficient block — shown by overwriting those blocks on the device with 0x7F7F first"""
        fs, lib = self.fs, self.lib
        G = fs.F
        if poison:
            coef = fs.coef.copy()
            inter_empty = ((fs.mb["mb_type"] & 7) == 0) & ((fs.mb["cbp"] & 0x3F) == 0)     # not Intra4x4 / 16x16 / PCM
            coef[inter_empty] = 0x7F7F
            # ... nor the parts of a coded macroblock its coded_block_pattern leaves out (an 8x8 luma quadrant per bit 0..3, both chroma planes for bits 4-5)
            inter = (fs.mb["mb_type"] & 7) == 0
            for q in range(4):
                coef[inter & ((fs.mb["cbp"] >> q) & 1 == 0), 64 * q:64 * q + 64] = 0x7F7F
            coef[inter & ((fs.mb["cbp"] & 0x30) == 0), 256:384] = 0x7F7F
            nmb = fs.mb_w * fs.mb_h
            for f in range(self.F):
                self.h2d(self.coef + f * nmb * 768, coef[f % G])
        lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
        for name, args in (("mi355_h264_recon_inter_sparse_dev", (fs.mb_w, fs.mb_h)), ("mi355_h264_recon_intra_levels_dev", (fs.max_intra_level, lw)),
                           ("mi355_h264_deblock_dev", (fs.mb_w, fs.mb_h))):
            fn = getattr(lib, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_int] + [C.c_int if isinstance(a, int) else C.c_void_p for a in args] + [C.c_void_p]
            assert fn(self.d_desc, self.F, *args, None) == 0, name
        assert lib.mi355_sync(None) == 0

    def free(self):
        for p in self.bufs:
            self.lib.mi355_free(p)
        self.bufs = []


def synth_frames_fast(nframes, mb_w, mb_h, seed=0x264, nrefs=4, intra_frac=0.05, mv_range=64, lib=None, refs="noise", coef_b=None, partitions="16x16"):
    """Vectorised generator for the benchmark workload (headline variant):
    P pictures, one 16x16 partition per inter MB, `intra_frac` Intra16x16 MBs (they force bS 3/4 edges),
    ref_idx ~ U{0..nrefs-1}, mv ~ U[-mv_range, mv_range) quarter samples, 24 4x4 blocks coded w.p. 0.5.
    `lib`: a loaded libmi355dsp (its host helper mi355_h264_intra_schedule builds the intra schedule).
    refs="smooth" / coef_b (Laplace scale of the levels, default 24): content on which the loop filter's conditions hold.
    partitions="mixed": SURVEY 8d's second run — each inter macroblock is 16x16, 16x8, 8x16 or 8x8 (a quarter each), the 8x8
    quadrants 8x8 / 8x4 / 4x8 / 4x4 (a quarter each); one vector per partition, one reference per partition (per quadrant in 8x8)."""
    fs = FrameSet(nframes, mb_w, mb_h, nrefs)
    r = SplitMix64(seed)
    COEF_B[0] = 24 if coef_b is None else coef_b
    COEF_CLIP[0] = 2047
    dc_scale = COEF_B[0] / 24.0
    nmb = mb_w * mb_h
    N = nframes * nmb
    for f in range(nframes):
        for s in range(nrefs):
            if refs == "noise":
                fs.refs[f][s] = (r.u8((fs.H, fs.W)), r.u8((fs.H // 2, fs.W // 2)), r.u8((fs.H // 2, fs.W // 2)))
            else:
                fs.refs[f][s] = (_ref_plane(r, fs.H, fs.W, refs), _ref_plane(r, fs.H // 2, fs.W // 2, refs), _ref_plane(r, fs.H // 2, fs.W // 2, refs))
    sl = fs.slices[:, 0]
    sl["list_count"] = 1
    sl["ref_slot"][:, 0, :nrefs] = np.arange(nrefs)
    sl["ref_slot"][:, 1, :nrefs] = np.arange(nrefs)
    sl["chroma_qp_table"][:, 0] = CHROMA_QP
    sl["chroma_qp_table"][:, 1] = CHROMA_QP
    mb = fs.mb.reshape(N)
    mbx = np.tile(np.arange(nmb) % mb_w, nframes)
    mby = np.tile(np.arange(nmb) // mb_w, nframes)
    qp = r.randint(20, 40, N)
    mb["qp"] = qp
    mb["flags"] = (mbx > 0) * F_LEFT | (mby > 0) * F_TOP
    qpc = np.array(CHROMA_QP)[qp]
    dq = np.array(DQ0)
    mb["dc_qmul"][:, 0] = (dq[qp % 6] * 16) << (qp // 6 + 2)
    mb["dc_qmul"][:, 1] = mb["dc_qmul"][:, 2] = (dq[qpc % 6] * 16) << (qpc // 6 + 2)
    intra = r.uniform(N) < intra_frac
    top, left = mby > 0, mbx > 0
    # Intra16x16 / chroma modes allowed by availability: both -> {DC,H,V,plane}; left only -> {H,LEFT_DC};
    # top only -> {V,TOP_DC}; none -> DC_128
    pick = r.randint(0, 3, (2, N))
    both = top & left
    mode = np.where(both, pick, np.where(left, np.where(pick & 1, 1, 4), np.where(top, np.where(pick & 1, 2, 5), 6)))
    mb["i16mode"] = np.where(intra, mode[0], 0)
    mb["chroma_mode"] = np.where(intra, mode[1], 0)
    tl = np.where(top, 0xFFFF, 0xB3FF)
    tl = np.where(left, tl, tl & 0xDF5F)
    tl = np.where(top & left, tl, tl & 0x7FFF)
    tr = np.where(top, 0xEEEA, 0x26EA)
    tr = np.where(top & (mbx + 1 < mb_w), tr, tr & 0xFBFF)
    mb["topleft"] = np.where(intra, tl, 0)
    mb["topright"] = np.where(intra, tr, 0)
    mb["mb_type"] = np.where(intra, I16, T16x16 | P0L0)
    ref = r.randint(0, nrefs - 1, N)
    mb["ref_idx"][:, 0, :] = np.where(intra, -1, ref)[:, None]
    mb["ref_idx"][:, 1, :] = -1
    mb["qpc"][:, 0] = mb["qpc"][:, 1] = qpc
    inter_rows = np.nonzero(~intra)[0]
    mb["i4mode"][inter_rows, 0:4] = ref[inter_rows, None]      # inter.ref_pic[0][q] = ref_slot[0][ref_idx] (identity map)
    mb["i4mode"][inter_rows, 4:8] = -1
    mv = r.randint(-mv_range, mv_range - 1, (N, 2))
    mv[intra] = 0
    fs.mv[0].reshape(N, 16, 2)[:] = mv[:, None, :]
    if partitions == "mixed":
        shape = r.randint(0, 3, N)                     # 0 16x16, 1 16x8, 2 8x16, 3 8x8
        sub = r.randint(0, 3, (N, 4))                   # per quadrant: MI355_SUB_8x8 / 8x4 / 4x8 / 4x4
        mv16 = r.randint(-mv_range, mv_range - 1, (N, 16, 2))      # a vector per 4x4 block (raster); partitions take their first block's
        refq = r.randint(0, nrefs - 1, (N, 4))          # a reference per quadrant
        b = np.arange(16)
        bx, by = b & 3, b >> 2
        quad = (bx >> 1) + 2 * (by >> 1)
        # the block whose vector a block takes (raster index), per shape
        src16 = np.zeros(16, np.int64)
        src168 = np.where(by < 2, 0, 8)
        src816 = np.where(bx < 2, 0, 2)
        q0 = (2 * (quad & 1)) + 4 * (2 * (quad >> 1))  # first block of the block's quadrant
        per_sub = np.stack([q0, q0 + 4 * (by & 1), q0 + (bx & 1), b])          # 8x8, 8x4, 4x8, 4x4
        src = np.where((shape == 0)[:, None], src16[None, :], np.where((shape == 1)[:, None], src168[None, :], np.where((shape == 2)[:, None], src816[None, :],
                       per_sub[sub[:, quad], b[None, :]])))
        mvm = np.take_along_axis(mv16, src[:, :, None].repeat(2, axis=2), axis=1)
        # references: one per partition (16x8: quadrants 0, 1 | 2, 3; 8x16: 0, 2 | 1, 3), one per quadrant in 8x8
        rq = np.where((shape == 0)[:, None], refq[:, :1].repeat(4, axis=1), np.where((shape == 1)[:, None], refq[:, [0, 0, 2, 2]],
                      np.where((shape == 2)[:, None], refq[:, [0, 1, 0, 1]], refq)))
        inter = ~intra
        fs.mv[0].reshape(N, 16, 2)[inter] = mvm[inter]
        mb["ref_idx"][inter, 0, :] = rq[inter]
        mb["i4mode"][inter, 0:4] = rq[inter]
        mt = np.array([T16x16 | P0L0, T16x8 | P0L0 | P1L0, T8x16 | P0L0 | P1L0, T8x8 | P0L0 | P1L0], np.uint32)[shape]
        mb["mb_type"] = np.where(intra, I16, mt)
        mb["sub"] = np.where(((shape == 3) & inter)[:, None], sub | 0x10, 0)
    # residual: 24 blocks per MB
    blks, coded = _gen_block_coefs(r, N * 24, "sparse")
    blks = blks.reshape(N, 24, 16)
    coded = coded.reshape(N, 24)
    # chroma DC levels travel through the 2x2 transform; Intra16x16 luma DC through the 4x4 one
    blks[:, 16:, 0] = 0
    blks[intra, :16, 0] = 0
    coded &= (blks != 0).any(axis=2)
    cmode = r.randint(0, 2, N)
    cdc = r.laplace_int(max(1, int(30 * dc_scale)), (N, 8), 2047)
    cdc[cmode == 0] = 0
    blks[cmode < 2, 16:, :] = 0
    coded[cmode < 2, 16:] = False
    blks[:, 16:, 0] = cdc
    ldc = r.laplace_int(max(1, int(40 * dc_scale)), (N, 16), 2047)
    has_ldc = intra & (r.uniform(N) < 0.8)
    slots = np.array([luma_dc_slot(k) for k in range(16)])
    cf = fs.coef.reshape(N, 384)
    cf[:] = blks.reshape(N, 384)
    ii = np.nonzero(has_ldc)[0]
    cf[ii[:, None], slots[None, :]] = ldc[ii]
    mask = (coded.astype(np.uint32) << np.arange(24, dtype=np.uint32)[None, :]).sum(axis=1).astype(np.uint32)
    mask |= (has_ldc & ldc.any(axis=1)).astype(np.uint32) << 24
    mask |= cdc[:, :4].any(axis=1).astype(np.uint32) << 25
    mask |= cdc[:, 4:].any(axis=1).astype(np.uint32) << 26
    mb["nnz_mask"] = mask
    chroma_cbp = np.where(coded[:, 16:].any(axis=1), 2, np.where(cdc.any(axis=1), 1, 0))
    cbp = chroma_cbp << 4
    for q in range(4):
        cbp |= ((mask >> (4 * q)) & 0xF != 0).astype(np.int64) << q
    cbp = np.where(intra & ((mask & 0xFFFF) != 0), cbp | 15, cbp)
    mb["cbp"] = cbp
    # intra schedule through the library's host helper when available
    for f in range(nframes):
        if lib is not None:
            lst = np.zeros(nmb, np.uint32)
            start = np.zeros(mb_w + 2 * mb_h + 2, np.int32)
            width = C.c_int(0)
            fn = lib.mi355_h264_intra_schedule
            fn.restype = C.c_int
            mx = fn(C.c_void_p(fs.mb[f].ctypes.data), mb_w, mb_h, C.c_void_p(lst.ctypes.data),
                    C.c_void_p(start.ctypes.data), C.byref(width))
            fs.intra_list[f] = lst[:start[mx]].copy()
            fs.intra_start[f] = start[:mx + 1].copy()
            fs.max_level_width = max(fs.max_level_width, width.value)
            note_level_widths(fs, np.diff(start[:mx + 1]).tolist())
        else:
            mx = intra_schedule(fs, f)
        fs.max_intra_level = max(fs.max_intra_level, mx)
    return fs
