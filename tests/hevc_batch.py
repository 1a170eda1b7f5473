"""Batched (Tier-2) HEVC entry points of include/mi355_hevc_batch.h against the oracle: each check builds a
picture's worth of independent jobs on non-overlapping cells, runs them in ONE launch on device memory and
replays the same jobs one by one through the oracle's HEVCDSPContext on a host copy."""
import ctypes as C

import numpy as np

import abi_ctypes as A
from cases_hevc import EW, QW, pixels
from rng import SplitMix64


class TuJob(C.Structure):
    _fields_ = [("coeffs", C.c_void_p), ("dst", C.c_void_p), ("dst_stride", C.c_int32), ("log2_size", C.c_uint8),
                ("col_limit", C.c_uint8), ("kind", C.c_uint8), ("reserved", C.c_uint8)]


class McJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_stride", C.c_int32), ("dst_stride", C.c_int32),
                ("width", C.c_uint8), ("height", C.c_uint8), ("mx", C.c_uint8), ("my", C.c_uint8), ("chroma", C.c_uint8),
                ("reserved", C.c_uint8 * 3)]


class PredJob(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src1", C.c_void_p), ("src2", C.c_void_p), ("dst_stride", C.c_int32), ("src_stride", C.c_int32),
                ("width", C.c_uint8), ("height", C.c_uint8), ("kind", C.c_uint8), ("denom", C.c_uint8),
                ("w0", C.c_int16), ("w1", C.c_int16), ("o0", C.c_int16), ("o1", C.c_int16)]


class LfJob(C.Structure):
    _fields_ = [("pix", C.c_void_p), ("stride", C.c_int32), ("beta", C.c_int32), ("tc", C.c_int32 * 2),
                ("no_p", C.c_uint8 * 2), ("no_q", C.c_uint8 * 2), ("horizontal_edge", C.c_uint8), ("chroma", C.c_uint8),
                ("reserved", C.c_uint8 * 2)]


class SaoJob(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("stride", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("borders", C.c_int32 * 4), ("offset_val", C.c_int32 * 5), ("cls", C.c_uint8), ("edge", C.c_uint8),
                ("c_idx", C.c_uint8), ("eo_class", C.c_uint8), ("band_position", C.c_uint8), ("vert_edge", C.c_uint8),
                ("horiz_edge", C.c_uint8), ("diag_edge", C.c_uint8)]


assert C.sizeof(LfJob) == 32


class SaoPiece(C.Structure):
    _fields_ = [("offset_val", C.c_int32 * 5), ("cls", C.c_uint8), ("type", C.c_uint8), ("eo_class", C.c_uint8), ("band_position", C.c_uint8),
                ("vert_edge", C.c_uint8), ("horiz_edge", C.c_uint8), ("diag_edge", C.c_uint8), ("borders", C.c_uint8),
                ("dx", C.c_int16), ("dy", C.c_int16), ("width", C.c_int16), ("height", C.c_int16)]


class SaoCtbJob(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("stride", C.c_int32), ("c_idx", C.c_uint8), ("npieces", C.c_uint8), ("reserved", C.c_uint8 * 2),
                ("piece", SaoPiece * 4)]


class EdgeEmuJob(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("dst_stride", C.c_int32), ("src_stride", C.c_int32), ("block_w", C.c_int32),
                ("block_h", C.c_int32), ("src_x", C.c_int32), ("src_y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


assert C.sizeof(SaoPiece) == 36 and C.sizeof(SaoCtbJob) == 168 and C.sizeof(EdgeEmuJob) == 48


class McPredJob(C.Structure):
    _fields_ = [("src0", C.c_void_p), ("src1", C.c_void_p), ("dst", C.c_void_p), ("src0_stride", C.c_int32), ("src1_stride", C.c_int32),
                ("dst_stride", C.c_int32), ("width", C.c_uint8), ("height", C.c_uint8), ("chroma", C.c_uint8), ("kind", C.c_uint8),
                ("mx0", C.c_uint8), ("my0", C.c_uint8), ("mx1", C.c_uint8), ("my1", C.c_uint8), ("denom", C.c_uint8),
                ("reserved", C.c_uint8 * 3), ("w0", C.c_int16), ("w1", C.c_int16), ("o0", C.c_int16), ("o1", C.c_int16),
                ("src0_b", C.c_void_p), ("src1_b", C.c_void_p), ("dst_b", C.c_void_p)]       # chroma == 2: the second plane


class IntraJob(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("top", C.c_void_p), ("left", C.c_void_p), ("stride", C.c_int32), ("log2_size", C.c_uint8),
                ("kind", C.c_uint8), ("c_idx", C.c_uint8), ("mode", C.c_uint8)]


class Dev:
    """device memory through the C ABI's helpers"""

    def __init__(self, lib):
        self.lib, self.bufs = lib, []
        lib.mi355_malloc.restype = C.c_void_p
        lib.mi355_malloc.argtypes = [C.c_size_t]
        lib.mi355_free.argtypes = [C.c_void_p]
        for f in ("mi355_memcpy_h2d", "mi355_memcpy_d2h"):
            getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]

    def up(self, a):
        a = np.ascontiguousarray(a)
        p = self.lib.mi355_malloc(a.nbytes + 64)
        assert p
        self.bufs.append(p)
        self.lib.mi355_memcpy_h2d(p, a.ctypes.data, a.nbytes)
        return p

    def up_jobs(self, jobs):
        arr = (type(jobs[0]) * len(jobs))(*jobs)
        p = self.lib.mi355_malloc(C.sizeof(arr))
        self.bufs.append(p)
        self.lib.mi355_memcpy_h2d(p, C.addressof(arr), C.sizeof(arr))
        return p

    def up_struct(self, arr):
        p = self.lib.mi355_malloc(C.sizeof(arr))
        self.bufs.append(p)
        self.lib.mi355_memcpy_h2d(p, C.addressof(arr), C.sizeof(arr))
        return p

    def down(self, p, like):
        out = np.empty_like(like)
        self.lib.mi355_sync(None)
        self.lib.mi355_memcpy_d2h(out.ctypes.data, p, out.nbytes)
        return out

    def free(self):
        for p in self.bufs:
            self.lib.mi355_free(p)
        self.bufs = []


def _u8p(a, off_bytes=0):
    return C.cast(a.ctypes.data + off_bytes, A.u8p)


def _i16p(a, off_elems=0):
    return C.cast(a.ctypes.data + 2 * off_elems, A.i16p)


def check_residual(prov, oracle, bd, seed, cells=(6, 8)):
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    cy, cx = cells
    pic = pixels(r, (cy * 32, cx * 32), bd)
    stride = pic.strides[0]
    n = cy * cx
    coefs = np.zeros((n, 32 * 32), np.int16)
    meta = []
    for k in range(n):
        log2 = r.randint(2, 5)
        size = 1 << log2
        kind = [0, 0, 0, 1, 2, 3][r.randint(0, 5)]
        if kind >= 2:
            log2, size = 2, 4
        lim = min(size, [1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 20, 24, 31, 32][r.randint(0, 13)])
        c = r.laplace_int(300, size * size, 32767).astype(np.int16)
        if kind == 0:
            # the boundary's contract (mi355_hevc_batch.h): rows from col_limit + 4 on hold zeros (col_limit = last_x + last_y + 4 of a diagonal scan,
            # hevcdec.c:1178-1196: no coefficient lies that low) — the kernel does not fetch them; everything above may hold anything (checkasm-like)
            c.reshape(size, size)[lim + 4:, :] = 0
        if kind == 0 and r.randint(0, 2):
            m = c.reshape(size, size)
            m[lim:, :] = 0
            m[:, lim:] = 0
        if kind == 0 and size > 4 and r.randint(0, 2) == 0:
            # a block as the DECODER makes it: a last significant position (lx, ly) of the diagonal scan, coefficients only in the 4x4 groups the scan
            # reaches before that one's (and inside it up to the position's own diagonal), and col_limit from (lx, ly) with hevcdec.c:1245-1256's caps
            lx, ly = r.randint(0, size - 1), r.randint(0, size - 1)
            if lx == 0 and ly == 0:
                lx = 1
            c = r.laplace_int(300, size * size, 32767).astype(np.int16)
            m = c.reshape(size, size)                       # m[y][x]
            ys, xs = np.mgrid[0:size, 0:size]
            gl = (lx >> 2) + (ly >> 2)
            keep = ((xs >> 2) + (ys >> 2) < gl) | (((xs >> 2) == (lx >> 2)) & ((ys >> 2) == (ly >> 2)) & ((xs & 3) + (ys & 3) <= (lx & 3) + (ly & 3)))
            m[~keep] = 0
            m[ly, lx] = m[ly, lx] or 1
            mx = max(lx, ly)
            lim = lx + ly + 4
            lim = min(4, lim) if mx < 4 else (min(8, lim) if mx < 8 else (min(24, lim) if mx < 12 else lim))
            lim = min(lim, size) if lim > size else lim
        if kind == 1:
            c[1:] = 0x1111
        coefs[k, :size * size] = c
        to_pic = r.randint(0, 3) != 0
        y0, x0 = (k // cx) * 32 + r.randint(0, (32 - size) // 4) * 4, (k % cx) * 32 + r.randint(0, (32 - size) // 4) * 4
        meta.append((log2, kind, lim, to_pic, y0, x0))
    # oracle
    c_o = oracle.hevcdsp(bd)
    pic_o, coef_o = pic.copy(), coefs.copy()
    for k, (log2, kind, lim, to_pic, y0, x0) in enumerate(meta):
        blk = coef_o[k]
        i = log2 - 2
        if kind == 0:
            c_o.idct[i](_i16p(blk), lim)
        elif kind == 1:
            c_o.idct_dc[i](_i16p(blk))
        elif kind == 2:
            c_o.transform_4x4_luma(_i16p(blk))
        else:
            c_o.dequant(_i16p(blk))
        if to_pic:
            c_o.add_residual[i](_u8p(pic_o, y0 * stride + x0 * px), _i16p(blk), stride)
    # device
    d = Dev(prov.lib)
    try:
        p_pic, p_coef = d.up(pic), d.up(coefs)
        jobs = [TuJob(p_coef + k * 2048, (p_pic + y0 * stride + x0 * px) if to_pic else None, stride, log2, lim, kind, 0)
                for k, (log2, kind, lim, to_pic, y0, x0) in enumerate(meta)]
        assert prov.lib.mi355_hevc_residual_batch_dev(C.c_void_p(d.up_jobs(jobs)), len(jobs), bd, None) == 0
        pic_g, coef_g = d.down(p_pic, pic), d.down(p_coef, coefs)
    finally:
        d.free()
    assert np.array_equal(pic_g, pic_o), "residual: picture differs (bd %d)" % bd
    for k, (log2, kind, lim, to_pic, y0, x0) in enumerate(meta):
        if not to_pic:
            assert np.array_equal(coef_g[k], coef_o[k]), "residual: coefficients of job %d differ" % k
    return len(jobs)


def check_mc(prov, oracle, bd, seed, n=48):
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    ref = pixels(r, (200, 320), bd)
    stride = ref.strides[0]
    out = np.full((n, 64 * 64), 0x2222, np.int16)
    meta = []
    for k in range(n):
        chroma = r.randint(0, 1)
        wi = r.randint(0, 7)
        w = (EW if chroma else QW)[wi]
        h = [4, 8, 12, 16, 24, 32, 64][r.randint(0, 6)]
        if chroma:
            h = min(h, 32)
        mx, my = (r.randint(0, 7), r.randint(0, 7)) if chroma else (r.randint(0, 3), r.randint(0, 3))
        y0, x0 = r.randint(8, 200 - 8 - h - 8), r.randint(8, 320 - 8 - w - 8)
        meta.append((chroma, wi, w, h, mx, my, y0, x0))
    c_o = oracle.hevcdsp(bd)
    out_o = out.copy()
    mcbuf = np.zeros((64 + 24) * 64, np.int16)
    for k, (chroma, wi, w, h, mx, my, y0, x0) in enumerate(meta):
        tab = c_o.put_hevc_epel if chroma else c_o.put_hevc_qpel
        tab[int(my != 0)][int(mx != 0)][wi](_i16p(out_o[k]), 128, _u8p(ref, y0 * stride + x0 * px), stride, h, mx, my, _i16p(mcbuf))
    d = Dev(prov.lib)
    try:
        p_ref, p_out = d.up(ref), d.up(out)
        jobs = [McJob(p_ref + y0 * stride + x0 * px, p_out + k * 8192, stride, 128, w, h, mx, my, chroma)
                for k, (chroma, wi, w, h, mx, my, y0, x0) in enumerate(meta)]
        assert prov.lib.mi355_hevc_mc_batch_dev(C.c_void_p(d.up_jobs(jobs)), n, bd, None) == 0
        out_g = d.down(p_out, out)
    finally:
        d.free()
    assert np.array_equal(out_g, out_o), "mc batch differs (bd %d)" % bd
    return n


def check_pred(prov, oracle, bd, seed, cells=(4, 6)):
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    cy, cx = cells
    pic = pixels(r, (cy * 64, cx * 64), bd)
    stride = pic.strides[0]
    n = cy * cx
    s1 = r.randint(-8192, 24575, (n, 64 * 64)).astype(np.int16)
    s2 = r.randint(-8192, 24575, (n, 64 * 64)).astype(np.int16)
    meta = []
    for k in range(n):
        chroma = r.randint(0, 1)
        wi = r.randint(0, 7)
        w = (EW if chroma else QW)[wi]
        h = [2, 4, 8, 16, 32, 64][r.randint(0, 5)]
        meta.append((chroma, wi, w, h, r.randint(0, 3), r.randint(0, 7), r.randint(-128, 127), r.randint(-128, 127),
                     r.randint(-128, 127), r.randint(-128, 127), (k // cx) * 64, (k % cx) * 64))
    c_o = oracle.hevcdsp(bd)
    pic_o = pic.copy()
    for k, (chroma, wi, w, h, kind, denom, w0, w1, o0, o1, y0, x0) in enumerate(meta):
        dp = _u8p(pic_o, y0 * stride + x0 * px)
        tabs = ((c_o.put_unweighted_pred_chroma, c_o.put_unweighted_pred_avg_chroma, c_o.weighted_pred_chroma, c_o.weighted_pred_avg_chroma)
                if chroma else (c_o.put_unweighted_pred, c_o.put_unweighted_pred_avg, c_o.weighted_pred, c_o.weighted_pred_avg))
        fn = tabs[kind][wi]
        if kind == 0:
            fn(dp, stride, _i16p(s1[k]), 128, h)
        elif kind == 1:
            fn(dp, stride, _i16p(s1[k]), _i16p(s2[k]), 128, h)
        elif kind == 2:
            fn(denom, w0, o0, dp, stride, _i16p(s1[k]), 128, h)
        else:
            fn(denom, w0, w1, o0, o1, dp, stride, _i16p(s1[k]), _i16p(s2[k]), 128, h)
    d = Dev(prov.lib)
    try:
        p_pic, p1, p2 = d.up(pic), d.up(s1), d.up(s2)
        jobs = [PredJob(p_pic + y0 * stride + x0 * px, p1 + k * 8192, p2 + k * 8192, stride, 128, w, h, kind, denom, w0, w1, o0, o1)
                for k, (chroma, wi, w, h, kind, denom, w0, w1, o0, o1, y0, x0) in enumerate(meta)]
        assert prov.lib.mi355_hevc_pred_batch_dev(C.c_void_p(d.up_jobs(jobs)), n, bd, None) == 0
        pic_g = d.down(p_pic, pic)
    finally:
        d.free()
    assert np.array_equal(pic_g, pic_o), "pred batch differs (bd %d)" % bd
    return n


def check_deblock(prov, oracle, bd, seed, cells=(7, 9)):
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    cy, cx = cells
    pic = pixels(r, (cy * 16, cx * 16), bd, smooth=True)
    pic[::5, ::3] = pixels(r, pic[::5, ::3].shape, bd)
    stride = pic.strides[0]
    meta = []
    for k in range(cy * cx):                 # one edge in the middle of each 16x16 cell: nothing overlaps
        y0, x0 = (k // cx) * 16 + 8, (k % cx) * 16 + 8
        horiz, chroma = r.randint(0, 1), r.randint(0, 1)
        step = r.randint(-10 << (bd - 8), 10 << (bd - 8))
        blk = pic[y0 - 8:y0 + 8, x0 - 8:x0 + 8].astype(np.int64)
        if horiz:
            blk[8:, :] += step
        else:
            blk[:, 8:] += step
        pic[y0 - 8:y0 + 8, x0 - 8:x0 + 8] = np.clip(blk, 0, (1 << bd) - 1)
        # pix = first sample on the q side of an 8-sample edge starting at the cell centre row/column - 4
        py, pxx = (y0, x0 - 4) if horiz else (y0 - 4, x0)
        meta.append((horiz, chroma, py, pxx, r.randint(0, 64), [r.randint(0, 24), r.randint(0, 24)],
                     [int(r.randint(0, 3) == 0), int(r.randint(0, 3) == 0)], [int(r.randint(0, 3) == 0), int(r.randint(0, 3) == 0)]))
    c_o = oracle.hevcdsp(bd)
    pic_o = pic.copy()
    for horiz, chroma, py, pxx, beta, tc, no_p, no_q in meta:
        tcv = np.array(tc, np.int32)
        npv, nqv = np.array(no_p, np.uint8), np.array(no_q, np.uint8)
        pix = _u8p(pic_o, py * stride + pxx * px)
        tcp = C.cast(tcv.ctypes.data, A.intp)
        if chroma:
            (c_o.hevc_h_loop_filter_chroma if horiz else c_o.hevc_v_loop_filter_chroma)(pix, stride, tcp, _u8p(npv), _u8p(nqv))
        else:
            (c_o.hevc_h_loop_filter_luma if horiz else c_o.hevc_v_loop_filter_luma)(pix, stride, beta, tcp, _u8p(npv), _u8p(nqv))
    d = Dev(prov.lib)
    try:
        p_pic = d.up(pic)
        jobs = []
        for horiz, chroma, py, pxx, beta, tc, no_p, no_q in meta:
            j = LfJob(p_pic + py * stride + pxx * px, stride, beta)
            j.tc[0], j.tc[1] = tc
            j.no_p[0], j.no_p[1] = no_p
            j.no_q[0], j.no_q[1] = no_q
            j.horizontal_edge, j.chroma = horiz, chroma
            jobs.append(j)
        assert prov.lib.mi355_hevc_deblock_batch_dev(C.c_void_p(d.up_jobs(jobs)), len(jobs), bd, None) == 0
        pic_g = d.down(p_pic, pic)
    finally:
        d.free()
    assert np.array_equal(pic_g, pic_o), "deblock batch differs (bd %d)" % bd
    return len(jobs)


def check_sao(prov, oracle, bd, seed, cells=(3, 4)):
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    cy, cx = cells
    src = pixels(r, (cy * 96, cx * 96), bd, smooth=True)
    src[::3, ::5] = pixels(r, src[::3, ::5].shape, bd)
    dst = np.full_like(src, 0x155 if bd > 8 else 0x55)
    stride = src.strides[0]
    meta = []
    for k in range(cy * cx):
        c_idx = r.randint(0, 2)
        w = [16, 32, 64][r.randint(0, 2)] >> (1 if c_idx else 0)
        h = [16, 32, 64][r.randint(0, 2)] >> (1 if c_idx else 0)
        meta.append(dict(y0=(k // cx) * 96 + 16, x0=(k % cx) * 96 + 16, c_idx=c_idx, w=w, h=h, cls=r.randint(0, 3), edge=r.randint(0, 1),
                         off=[0] + [r.randint(-7 << (bd - 8), 7 << (bd - 8)) for _ in range(4)], band=r.randint(0, 31), eo=r.randint(0, 3),
                         borders=[r.randint(0, 1) for _ in range(4)], ve=r.randint(0, 1), he=r.randint(0, 1), de=r.randint(0, 1)))
    c_o = oracle.hevcdsp(bd)
    dst_o = dst.copy()
    for m in meta:
        sao = A.SAOParams()
        for i in range(5):
            sao.offset_val[m["c_idx"]][i] = m["off"][i]
        sao.band_position[m["c_idx"]] = m["band"]
        sao.eo_class[m["c_idx"]] = m["eo"]
        bo = np.array(m["borders"], np.int32)
        off = m["y0"] * stride + m["x0"] * px
        if m["edge"]:
            c_o.sao_edge_filter[m["cls"]](_u8p(dst_o, off), _u8p(src, off), stride, C.byref(sao), C.cast(bo.ctypes.data, A.intp),
                                          m["w"], m["h"], m["c_idx"], m["ve"], m["he"], m["de"])
        else:
            c_o.sao_band_filter[m["cls"]](_u8p(dst_o, off), _u8p(src, off), stride, C.byref(sao), C.cast(bo.ctypes.data, A.intp),
                                          m["w"], m["h"], m["c_idx"])
    d = Dev(prov.lib)
    try:
        p_src, p_dst = d.up(src), d.up(dst)
        jobs = []
        for m in meta:
            off = m["y0"] * stride + m["x0"] * px
            j = SaoJob(p_dst + off, p_src + off, stride, m["w"], m["h"])
            for i in range(4):
                j.borders[i] = m["borders"][i]
            for i in range(5):
                j.offset_val[i] = m["off"][i]
            j.cls, j.edge, j.c_idx, j.eo_class, j.band_position = m["cls"], m["edge"], m["c_idx"], m["eo"], m["band"]
            j.vert_edge, j.horiz_edge, j.diag_edge = m["ve"], m["he"], m["de"]
            jobs.append(j)
        assert prov.lib.mi355_hevc_sao_batch_dev(C.c_void_p(d.up_jobs(jobs)), len(jobs), bd, None) == 0
        dst_g = d.down(p_dst, dst)
    finally:
        d.free()
    assert np.array_equal(dst_g, dst_o), "sao batch differs (bd %d)" % bd
    return len(jobs)


def sao_ctb_pieces(cx, cy, cw, chn, params, slice_addr, filter_edges):
    """what sao_filter_CTB (hevc_filter.c:188-314) derives for CTB (cx, cy): [(class, parameters of the owning CTB, vert, horiz, diag)]
    in the reference's order, and the CTB's picture-border flags"""
    here = cy * cw + cx
    has_l, has_u = cx > 0, cy > 0
    a_c = slice_addr[here]
    a_l = slice_addr[here - 1] if has_l else a_c
    a_u = slice_addr[here - cw] if has_u else a_c
    a_ul = slice_addr[here - cw - 1] if has_l and has_u else a_c
    f_c = filter_edges[here]
    f_l = filter_edges[here - 1] if has_l else 1
    f_u = filter_edges[here - cw] if has_u else 1
    vert, horiz, diag = [0] * 4, [0] * 4, [0] * 4
    if has_l:
        vert[0] = vert[2] = int(not f_c and a_c != a_l)
    if has_u:
        horiz[0] = horiz[1] = int(not f_c and a_c != a_u)
    if has_l and has_u:
        vert[1] = vert[3] = int(not f_u and a_u != a_ul)
        horiz[2] = horiz[3] = int(not f_l and a_l != a_ul)
        diag[0] = diag[3] = int(not f_c and a_c != a_ul)
        diag[1] = diag[2] = int(not f_l) if a_l > a_u else (int(not f_u) if a_l < a_u else 0)
    order = [0] + ([2] if has_l else []) + ([1] if has_u else []) + ([3] if has_l and has_u else [])
    return [(k, params[here - (k & 1) * cw - (k >> 1)], vert[k], horiz[k], diag[k]) for k in order], [int(cx == 0), int(cy == 0), int(cx == cw - 1), int(cy == chn - 1)]


def check_sao_ctbs(prov, oracle, bd, seed, sizes=((200, 150), (328, 264)), log2_ctb=6):
    """two pictures: 4 x 3 CTBs (every CTB touches a border or a slice edge somewhere) and 6 x 5 (interior CTBs take the whole-region forms)"""
    return sum(_check_sao_ctbs(prov, oracle, bd, seed + 7 * i, size, log2_ctb) for i, size in enumerate(sizes))


def _check_sao_ctbs(prov, oracle, bd, seed, size, log2_ctb):
    """mi355_hevc_sao_ctbs_dev (one job per CTB component: the up to four reference calls that make up the CTB's own samples) on a whole 4:2:0 picture with ragged
    last CTBs, random parameters (off / band / edge), two slices with and without filtering across their edge — against
    sao_filter_CTB restated over the oracle's table functions: per CTB in raster order, copy_CTB of the CTB shifted by 8 / 4, then
    the pieces"""
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    W, H = size
    ctb = 1 << log2_ctb
    cw, chn = -(-W // ctb), -(-H // ctb)
    planes = [pixels(r, (H >> (c > 0), W >> (c > 0)), bd, smooth=True) for c in range(3)]
    for pl in planes:
        pl[::3, ::5] = pixels(r, pl[::3, ::5].shape, bd)
    outs = [np.full_like(pl, 0x155 if bd > 8 else 0x55) for pl in planes]
    params = []
    for _ in range(cw * chn):
        params.append(dict(type=[r.randint(0, 2) for _ in range(3)], off=[[0] + [r.randint(-7 << (bd - 8), 7 << (bd - 8)) for _ in range(4)] for _ in range(3)],
                           band=[r.randint(0, 31) for _ in range(3)], eo=[r.randint(0, 3) for _ in range(3)]))
    first2 = r.randint(1, cw * chn - 1)                    # the second slice starts here (raster scan = tile scan)
    slice_addr = [0 if i < first2 else first2 for i in range(cw * chn)]
    filter_edges = [1 if i < first2 else r.randint(0, 1) for i in range(cw * chn)]
    if cw * chn > 2:
        filter_edges[first2] = 0
    c_o = oracle.hevcdsp(bd)
    exp = [o.copy() for o in outs]
    owner = {}                                             # (owner CTB, component) -> its pieces as the reference's calls produce them
    for cy in range(chn):
        for cx in range(cw):
            pieces, borders = sao_ctb_pieces(cx, cy, cw, chn, params, slice_addr, filter_edges)
            bo = np.array(borders, np.int32)
            for c in range(3):
                sh = 1 if c else 0
                size_c = ctb >> sh
                x0, y0 = cx * size_c, cy * size_c
                w, h = min(size_c, (W >> sh) - x0), min(size_c, (H >> sh) - y0)
                src, dst = planes[c], exp[c]
                stride = src.strides[0]
                xs, ys = (0 if borders[0] else 8 >> sh), (0 if borders[1] else 4 >> sh)
                cwid, chgt = (w + xs if borders[2] else w), (h + ys if borders[3] else h)
                dst[y0 - ys:y0 - ys + chgt, x0 - xs:x0 - xs + cwid] = src[y0 - ys:y0 - ys + chgt, x0 - xs:x0 - xs + cwid]      # copy_CTB
                off = y0 * stride + x0 * px
                for (k, p, ve, he, de) in pieces:
                    ox, oy = cx - (k >> 1), cy - (k & 1)
                    owner.setdefault((oy, ox, c), []).append(dict(cls=k, p=p, ve=ve, he=he, de=de, borders=borders, dx=(k >> 1) * size_c, dy=(k & 1) * size_c, w=w, h=h))
                    if p["type"][c] == 0:
                        continue
                    sao = A.SAOParams()
                    for i in range(5):
                        sao.offset_val[c][i] = p["off"][c][i]
                    sao.band_position[c], sao.eo_class[c] = p["band"][c], p["eo"][c]
                    if p["type"][c] == 2:
                        c_o.sao_edge_filter[k](_u8p(dst, off), _u8p(src, off), stride, C.byref(sao), C.cast(bo.ctypes.data, A.intp), w, h, c, ve, he, de)
                    else:
                        c_o.sao_band_filter[k](_u8p(dst, off), _u8p(src, off), stride, C.byref(sao), C.cast(bo.ctypes.data, A.intp), w, h, c)
    jobs = []
    for (oy, ox, c), pcs in sorted(owner.items()):
        sh = 1 if c else 0
        size_c = ctb >> sh
        stride = planes[c].strides[0]
        j = SaoCtbJob(0, 0, stride)
        j.c_idx, j.npieces = c, len(pcs)
        for n, m in enumerate(pcs):
            q, p = j.piece[n], m["p"]
            q.cls, q.type, q.eo_class, q.band_position, q.vert_edge, q.horiz_edge, q.diag_edge = m["cls"], p["type"][c], p["eo"][c], p["band"][c], m["ve"], m["he"], m["de"]
            q.borders = sum(b << e for e, b in enumerate(m["borders"]))
            q.dx, q.dy, q.width, q.height = m["dx"], m["dy"], m["w"], m["h"]
            for i in range(5):
                q.offset_val[i] = p["off"][c][i]
        jobs.append((c, oy * size_c * stride + ox * size_c * px, j))
    d = Dev(prov.lib)
    try:
        p_src, p_dst = [d.up(pl) for pl in planes], [d.up(o) for o in outs]
        arr = []
        for c, off, j in jobs:
            j.dst, j.src = p_dst[c] + off, p_src[c] + off
            arr.append(j)
        prov.lib.mi355_hevc_sao_ctbs_dev.restype = C.c_int
        assert prov.lib.mi355_hevc_sao_ctbs_dev(C.c_void_p(d.up_jobs(arr)), len(arr), bd, None) == 0
        got = [d.down(p_dst[c], outs[c]) for c in range(3)]
    finally:
        d.free()
    for c in range(3):
        assert np.array_equal(got[c], exp[c]), "CTB-level SAO differs in plane %d (bd %d)" % (c, bd)
    return len(jobs)


def check_edge_emu(prov, oracle, bd, seed, n=40):
    """mi355_edge_emu_batch_dev against the definition of emulated_edge_mc (videodsp_template.c:24-96: the window with every
    coordinate clamped to the plane)"""
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    w, h = 70, 41
    plane = pixels(r, (h, w + 6), bd)                      # 6 samples of row padding
    stride = plane.strides[0]
    meta = []
    for k in range(n):
        bw, bh = r.randint(1, 71), r.randint(1, 71)
        meta.append((bw, bh, r.randint(-bw - 3, w + 3), r.randint(-bh - 3, h + 3)))
    bufs = np.full((n, 71, 80), 0x155 if bd > 8 else 0x55, plane.dtype)
    exp = bufs.copy()
    for k, (bw, bh, sx, sy) in enumerate(meta):
        ys = np.clip(np.arange(sy, sy + bh), 0, h - 1)
        xs = np.clip(np.arange(sx, sx + bw), 0, w - 1)
        exp[k, :bh, :bw] = plane[:, :w][np.ix_(ys, xs)]
    d = Dev(prov.lib)
    try:
        p_pl, p_b = d.up(plane), d.up(bufs)
        jobs = [EdgeEmuJob(p_b + k * bufs.strides[0], p_pl + sy * stride + sx * px, bufs.strides[1], stride, bw, bh, sx, sy, w, h)
                for k, (bw, bh, sx, sy) in enumerate(meta)]
        prov.lib.mi355_edge_emu_batch_dev.restype = C.c_int
        assert prov.lib.mi355_edge_emu_batch_dev(C.c_void_p(d.up_jobs(jobs)), n, bd, None) == 0
        got = d.down(p_b, bufs)
    finally:
        d.free()
    assert np.array_equal(got, exp), "edge emulation batch differs (bd %d)" % bd
    return n


def check_mcpred(prov, oracle, bd, seed, cells=(4, 6)):
    """fused MC + prediction vs the oracle's put_hevc_qpel/epel followed by its (un)weighted prediction functions"""
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    cy, cx = cells
    ref0, ref1 = pixels(r, (200, 320), bd), pixels(r, (200, 320), bd)
    pic = pixels(r, (cy * 64, cx * 64), bd)
    rstride, stride = ref0.strides[0], pic.strides[0]
    n = cy * cx
    meta = []
    for k in range(n):
        chroma = r.randint(0, 1)
        wi = r.randint(0, 7)
        w = (EW if chroma else QW)[wi]
        h = [2, 4, 8, 16, 32, 64][r.randint(0, 5)]
        fmax = 7 if chroma else 3
        frac = [r.randint(0, fmax) if r.randint(0, 3) else 0 for _ in range(4)]
        pos = [(r.randint(8, 200 - 64 - 8), r.randint(8, 320 - 64 - 8)) for _ in range(2)]
        kind = r.randint(0, 3)
        # chroma == 2: both chroma planes of a block in one job (unweighted kinds): plane B reads the two references swapped
        # and writes 32 samples to the right of plane A in the block's 64x64 cell
        if chroma and kind < 2 and r.randint(0, 1):
            chroma = 2
        meta.append((chroma, wi, w, h, kind, r.randint(0, 7), r.randint(-128, 127), r.randint(-128, 127),
                     r.randint(-128, 127), r.randint(-128, 127), frac, pos, (k // cx) * 64, (k % cx) * 64))
    c_o = oracle.hevcdsp(bd)
    pic_o = pic.copy()
    mcbuf = np.zeros((64 + 24) * 64, np.int16)
    t0, t1 = np.zeros(64 * 64, np.int16), np.zeros(64 * 64, np.int16)
    for chroma, wi, w, h, kind, denom, w0, w1, o0, o1, frac, pos, y0, x0 in meta:
        tab = c_o.put_hevc_epel if chroma else c_o.put_hevc_qpel
        for plane in range(2 if chroma == 2 else 1):
            ra, rb = (ref1, ref0) if plane else (ref0, ref1)
            for t, ref, (sy, sx), (mx, my) in ((t0, ra, pos[0], frac[0:2]), (t1, rb, pos[1], frac[2:4])):
                tab[int(my != 0)][int(mx != 0)][wi](_i16p(t), 128, _u8p(ref, sy * rstride + sx * px), rstride, h, mx, my, _i16p(mcbuf))
            dp = _u8p(pic_o, y0 * stride + (x0 + 32 * plane) * px)
            tabs = ((c_o.put_unweighted_pred_chroma, c_o.put_unweighted_pred_avg_chroma, c_o.weighted_pred_chroma, c_o.weighted_pred_avg_chroma)
                    if chroma else (c_o.put_unweighted_pred, c_o.put_unweighted_pred_avg, c_o.weighted_pred, c_o.weighted_pred_avg))
            fn = tabs[kind][wi]
            if kind == 0:
                fn(dp, stride, _i16p(t0), 128, h)
            elif kind == 1:
                fn(dp, stride, _i16p(t0), _i16p(t1), 128, h)
            elif kind == 2:
                fn(denom, w0, o0, dp, stride, _i16p(t0), 128, h)
            else:
                fn(denom, w0, w1, o0, o1, dp, stride, _i16p(t0), _i16p(t1), 128, h)
    d = Dev(prov.lib)
    try:
        p_pic, p0, p1 = d.up(pic), d.up(ref0), d.up(ref1)
        jobs = []
        for chroma, wi, w, h, kind, denom, w0, w1, o0, o1, frac, pos, y0, x0 in meta:
            j = McPredJob(p0 + pos[0][0] * rstride + pos[0][1] * px, p1 + pos[1][0] * rstride + pos[1][1] * px, p_pic + y0 * stride + x0 * px,
                          rstride, rstride, stride, w, h, chroma, kind, frac[0], frac[1], frac[2], frac[3], denom)
            j.w0, j.w1, j.o0, j.o1 = w0, w1, o0, o1
            if chroma == 2:
                j.src0_b, j.src1_b = p1 + pos[0][0] * rstride + pos[0][1] * px, p0 + pos[1][0] * rstride + pos[1][1] * px
                j.dst_b = p_pic + y0 * stride + (x0 + 32) * px
            jobs.append(j)
        assert prov.lib.mi355_hevc_mcpred_batch_dev(C.c_void_p(d.up_jobs(jobs)), n, bd, None) == 0
        pic_g = d.down(p_pic, pic)
    finally:
        d.free()
    assert np.array_equal(pic_g, pic_o), "fused mc + pred batch differs (bd %d)" % bd
    return n


def check_intra(prov, oracle, bd, seed, cells=(5, 7)):
    """pred_planar / pred_dc / pred_angular of independent blocks, neighbour arrays in a device edge buffer"""
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    cy, cx = cells
    pic = pixels(r, (cy * 32, cx * 32), bd)
    stride = pic.strides[0]
    n = cy * cx
    edges = pixels(r, (n, 2, 72), bd)              # [job][top/left][-1 .. 2*size-1 (+ slack)]
    meta = []
    for k in range(n):
        log2 = r.randint(2, 5)
        kind = r.randint(0, 2)
        meta.append((log2, kind, r.randint(0, 2), r.randint(2, 34), (k // cx) * 32, (k % cx) * 32))
    h_o = oracle.hevcpred(bd)
    pic_o = pic.copy()
    for k, (log2, kind, c_idx, mode, y0, x0) in enumerate(meta):
        dp = _u8p(pic_o, y0 * stride + x0 * px)
        top, left = _u8p(edges, ((k * 2 + 0) * 72 + 1) * px), _u8p(edges, ((k * 2 + 1) * 72 + 1) * px)
        if kind == 0:
            h_o.pred_planar[log2 - 2](dp, top, left, stride // px)
        elif kind == 1:
            h_o.pred_dc(dp, top, left, stride // px, log2, c_idx)
        else:
            h_o.pred_angular[log2 - 2](dp, top, left, stride // px, c_idx, mode)
    d = Dev(prov.lib)
    try:
        p_pic, p_e = d.up(pic), d.up(edges)
        jobs = [IntraJob(p_pic + y0 * stride + x0 * px, p_e + ((k * 2 + 0) * 72 + 1) * px, p_e + ((k * 2 + 1) * 72 + 1) * px, stride,
                         log2, kind, c_idx, mode) for k, (log2, kind, c_idx, mode, y0, x0) in enumerate(meta)]
        assert prov.lib.mi355_hevc_intra_batch_dev(C.c_void_p(d.up_jobs(jobs)), n, bd, None) == 0
        pic_g = d.down(p_pic, pic)
    finally:
        d.free()
    assert np.array_equal(pic_g, pic_o), "intra batch differs (bd %d)" % bd
    return n



class Level(C.Structure):
    """mi355_hevc_level"""
    _fields_ = [("first_wg", C.c_uint32), ("mc0", C.c_uint32), ("n_mc", C.c_uint32), ("tu0", C.c_uint32), ("n_tu", C.c_uint32), ("in0", C.c_uint32),
                ("n_in", C.c_uint32), ("reserved", C.c_uint32)]


class CtbJob(C.Structure):
    _fields_ = [("dst", C.c_void_p * 3), ("stride", C.c_int32 * 3), ("width", C.c_uint16), ("height", C.c_uint16), ("log2_ctb_size", C.c_uint8),
                ("flags", C.c_uint8), ("reserved", C.c_uint8 * 2), ("first_mc", C.c_uint32), ("n_mc", C.c_uint32), ("first_tu", C.c_uint32),
                ("n_tu", C.c_uint32), ("reserved1", C.c_uint32)]


assert C.sizeof(CtbJob) == 64


def decoder_block(r, size):
    """coefficients of a size x size block as the DECODER leaves them (hevcdec.c:1062-1256): a last significant position of the diagonal scan,
    coefficients only in the 4x4 groups the scan reaches before that one's (and inside it up to the position's own diagonal); col_limit from it"""
    lx, ly = r.randint(0, size - 1), r.randint(0, size - 1)
    if r.randint(0, 2) == 0:                               # low-frequency blocks are the common case
        lx, ly = lx % 8, ly % 8
    if lx == 0 and ly == 0:
        lx = 1
    c = r.laplace_int(300, size * size, 32767).astype(np.int16)
    m = c.reshape(size, size)
    ys, xs = np.mgrid[0:size, 0:size]
    gl = (lx >> 2) + (ly >> 2)
    keep = ((xs >> 2) + (ys >> 2) < gl) | (((xs >> 2) == (lx >> 2)) & ((ys >> 2) == (ly >> 2)) & ((xs & 3) + (ys & 3) <= (lx & 3) + (ly & 3)))
    m[~keep] = 0
    m[ly, lx] = m[ly, lx] or 1
    mx = max(lx, ly)
    lim = lx + ly + 4
    lim = min(4, lim) if mx < 4 else (min(8, lim) if mx < 8 else (min(24, lim) if mx < 12 else lim))
    return c, lim


def check_recon_ctbs(prov, oracle, bd, seed, cells=(2, 3), promise_check=True, form="ctbs"):
    """mi355_hevc_recon_ctbs_dev — a coding tree block's prediction blocks and transform units in one workgroup — against the oracle's tables
    called block by block in the reference's order (hls_prediction_unit, then hls_transform_unit: hevcdec.c:1695-1885, :1238-1260): random
    partitions (squares 64..8, halves, quarter splits), every prediction kind with one or two references, chroma blocks alone and as pairs,
    transform trees 32..4 with every unit kind; some blocks leave samples uncovered (MI355_HEVC_CTB_PARTIAL)."""
    r = SplitMix64(seed)
    px = 2 if bd > 8 else 1
    cy, cx = cells
    H, W = cy * 64, cx * 64
    pic = [pixels(r, (H, W), bd), pixels(r, (H // 2, W // 2), bd), pixels(r, (H // 2, W // 2), bd)]
    # reference 0: rows a multiple of the matrix path's piece size; reference 1: not (those blocks take the general path)
    ry = [pixels(r, (H + 80, W + 80), bd), pixels(r, (H + 80, W + 84), bd)]
    rc = [[pixels(r, (H // 2 + 48, W // 2 + 48), bd) for _ in range(2)], [pixels(r, (H // 2 + 48, W // 2 + 44), bd) for _ in range(2)]]

    def split_pu(x, y, sz, out):
        if sz > 8 and r.randint(0, 99) < (100 if sz == 64 and r.randint(0, 3) else 45):
            for k in range(4):
                split_pu(x + (k & 1) * sz // 2, y + (k >> 1) * sz // 2, sz // 2, out)
            return
        m = r.randint(0, 7)
        if m == 0:
            out += [(x, y, sz, sz // 2), (x, y + sz // 2, sz, sz // 2)]
        elif m == 1:
            out += [(x, y, sz // 2, sz), (x + sz // 2, y, sz // 2, sz)]
        elif m == 2 and sz >= 16:
            out += [(x, y, sz, sz // 4), (x, y + sz // 4, sz, 3 * sz // 4)]
        elif m == 3 and sz >= 16:
            out += [(x, y, 3 * sz // 4, sz), (x + 3 * sz // 4, y, sz // 4, sz)]
        else:
            out.append((x, y, sz, sz))

    def split_tu(x, y, sz, out, mn):
        if sz > 32 or (sz > mn and r.randint(0, 99) < 40):
            for k in range(4):
                split_tu(x + (k & 1) * sz // 2, y + (k >> 1) * sz // 2, sz // 2, out, mn)
        elif r.randint(0, 3):
            out.append((x, y, sz))

    c_o = oracle.hevcdsp(bd)
    pic_o = [a.copy() for a in pic]
    mcbuf = np.zeros((64 + 24) * 64, np.int16)
    t0, t1 = np.zeros(64 * 64, np.int16), np.zeros(64 * 64, np.int16)
    mc_meta, tu_meta, ctbs, coefs, uniform_ctbs = [], [], [], [], []
    for cyi in range(cy):
        for cxi in range(cx):
            X, Y = cxi * 64, cyi * 64
            pus, first_mc, first_tu = [], len(mc_meta), len(tu_meta)
            uniform = r.randint(0, 2) == 0                   # a block of the measured shape: four 32x32 blocks of one reference, 32x32 units
            if uniform:
                pus = [(X + 32 * (k & 1), Y + 32 * (k >> 1), 32, 32) for k in range(4)]
            else:
                split_pu(X, Y, 64, pus)
            partial = (not uniform) and r.randint(0, 2) == 0
            if partial:
                pus = [q for q in pus if r.randint(0, 3)]
            for (x, y, w, h) in pus:
                kind = 0 if (uniform or r.randint(0, 1)) else r.randint(1, 3)
                refs = (0, 1) if (uniform or r.randint(0, 2)) else (1, 0)
                mv = [(r.randint(-96, 96), r.randint(-96, 96)) for _ in range(2)]          # quarter samples
                if r.randint(0, 3) == 0:
                    mv[0] = (mv[0][0] & ~3, mv[0][1])
                if r.randint(0, 3) == 0:
                    mv[0] = (mv[0][0], mv[0][1] & ~3)
                wts = (r.randint(0, 7), r.randint(-128, 127), r.randint(-128, 127), r.randint(-128, 127), r.randint(-128, 127))
                pair = kind < 2 and r.randint(0, 2) != 0
                for comp in ((0,), (1, 2)):
                    for c_idx in comp:
                        ch = c_idx != 0
                        bw, bh, bx, by = (w >> ch, h >> ch, x >> ch, y >> ch)
                        wi = (EW if ch else QW).index(bw)
                        tab = c_o.put_hevc_epel if ch else c_o.put_hevc_qpel
                        srcs = []
                        for t, ref, (mvx, mvy) in ((t0, refs[0], mv[0]), (t1, refs[1], mv[1])):
                            plane = rc[ref][c_idx - 1] if ch else ry[ref]
                            sx, sy = bx + (24 if ch else 40) + (mvx >> (3 if ch else 2)), by + (24 if ch else 40) + (mvy >> (3 if ch else 2))
                            fx, fy = (mvx & 7, mvy & 7) if ch else (mvx & 3, mvy & 3)
                            tab[int(fy != 0)][int(fx != 0)][wi](_i16p(t), 128, _u8p(plane, sy * plane.strides[0] + sx * px), plane.strides[0], bh, fx, fy, _i16p(mcbuf))
                            srcs.append((ref, sx, sy, fx, fy))
                        dp = _u8p(pic_o[c_idx], by * pic_o[c_idx].strides[0] + bx * px)
                        st = pic_o[c_idx].strides[0]
                        tabs = ((c_o.put_unweighted_pred_chroma, c_o.put_unweighted_pred_avg_chroma, c_o.weighted_pred_chroma, c_o.weighted_pred_avg_chroma)
                                if ch else (c_o.put_unweighted_pred, c_o.put_unweighted_pred_avg, c_o.weighted_pred, c_o.weighted_pred_avg))
                        fn = tabs[kind][wi]
                        denom, w0, w1, o0, o1 = wts
                        if kind == 0:
                            fn(dp, st, _i16p(t0), 128, bh)
                        elif kind == 1:
                            fn(dp, st, _i16p(t0), _i16p(t1), 128, bh)
                        elif kind == 2:
                            fn(denom, w0, o0, dp, st, _i16p(t0), 128, bh)
                        else:
                            fn(denom, w0, w1, o0, o1, dp, st, _i16p(t0), _i16p(t1), 128, bh)
                        mc_meta.append((c_idx, bx, by, bw, bh, kind, wts, srcs, pair))
            tus = []
            if uniform:
                tus = [(0, X + 32 * (k & 1), Y + 32 * (k >> 1), 32) for k in range(4)] + [(1, X // 2, Y // 2, 32), (2, X // 2, Y // 2, 32)]
            else:
                for c_idx in range(3):
                    leaves = []
                    split_tu(X >> (c_idx > 0), Y >> (c_idx > 0), 64 >> (c_idx > 0), leaves, 4)
                    tus += [(c_idx, x, y, sz) for (x, y, sz) in leaves]
            for (c_idx, x, y, sz) in tus:
                log2 = sz.bit_length() - 1
                kind = 0 if uniform else [0, 0, 0, 0, 1, 2, 3][r.randint(0, 6)]
                if kind >= 2 and sz != 4:
                    kind = 0
                lim = sz
                if kind == 0:
                    if sz >= 16 or r.randint(0, 1):
                        c, lim = decoder_block(r, sz)
                        if uniform and r.randint(0, 3) == 0:
                            c, lim = r.laplace_int(300, sz * sz, 32767).astype(np.int16), 32          # a dense block: no pruning
                    else:
                        lim = min(sz, [1, 2, 3, 4, 5, 7, 8][r.randint(0, 6)])
                        c = r.laplace_int(300, sz * sz, 32767).astype(np.int16)
                        c.reshape(sz, sz)[lim + 4:, :] = 0
                else:
                    c = r.laplace_int(300, sz * sz, 32767).astype(np.int16)
                    if kind == 1:
                        c[1:] = 0x1111
                blk = np.zeros(1024, np.int16)
                blk[:sz * sz] = c
                coefs.append(blk.copy())
                i = log2 - 2
                if kind == 0:
                    c_o.idct[i](_i16p(blk), lim)
                elif kind == 1:
                    c_o.idct_dc[i](_i16p(blk))
                elif kind == 2:
                    c_o.transform_4x4_luma(_i16p(blk))
                else:
                    c_o.dequant(_i16p(blk))
                st = pic_o[c_idx].strides[0]
                c_o.add_residual[i](_u8p(pic_o[c_idx], y * st + x * px), _i16p(blk), st)
                tu_meta.append((c_idx, x, y, log2, kind, lim))
            ctbs.append((X, Y, partial, first_mc, first_tu))
            # does the matrix path take every job of this block (include/mi355_hevc_batch.h)?  one reference whose rows are a multiple of the piece size,
            # sides multiples of 16 in the plane's samples, 16x16 / 32x32 inverse DCTs
            uniform_ctbs.append(all(m[5] == 0 and m[3] % 16 == 0 and m[4] % 16 == 0 and m[7][0][0] == 0 for m in mc_meta[first_mc:]) and
                                all(t[4] == 0 and t[3] >= 4 for t in tu_meta[first_tu:]))
    d = Dev(prov.lib)
    try:
        p_pic = [d.up(a) for a in pic]
        p_ry = [d.up(a) for a in ry]
        p_rc = [[d.up(a) for a in pl] for pl in rc]
        p_coef = d.up(np.stack(coefs))
        mc_jobs, ctb_first_mc = [], {}
        k = 0
        while k < len(mc_meta):
            c_idx, bx, by, bw, bh, kind, wts, srcs, pair = mc_meta[k]
            ctb_first_mc[k] = len(mc_jobs)
            plane_p = p_pic[c_idx]
            st = pic[c_idx].strides[0]

            def src_ptr(c, s):
                ref, sx, sy = s[0], s[1], s[2]
                plane = rc[ref][c - 1] if c else ry[ref]
                return (p_rc[ref][c - 1] if c else p_ry[ref]) + sy * plane.strides[0] + sx * px, plane.strides[0]
            (s0p, s0s), (s1p, s1s) = src_ptr(c_idx, srcs[0]), src_ptr(c_idx, srcs[1])
            j = McPredJob(s0p, s1p, plane_p + by * st + bx * px, s0s, s1s, st, bw, bh, 1 if c_idx else 0, kind, srcs[0][3], srcs[0][4], srcs[1][3], srcs[1][4], wts[0])
            j.w0, j.w1, j.o0, j.o1 = wts[1], wts[2], wts[3], wts[4]
            if c_idx == 1 and pair:
                # Cb and Cr of a block as one job: the same vector, strides and (unweighted) parameters
                srcs2 = mc_meta[k + 1][7]
                j.chroma = 2
                j.src0_b, j.src1_b = src_ptr(2, srcs2[0])[0], src_ptr(2, srcs2[1])[0]
                j.dst_b = p_pic[2] + by * st + bx * px
                ctb_first_mc[k + 1] = len(mc_jobs) + 1
                k += 1
            mc_jobs.append(j)
            k += 1
        ctb_first_mc[len(mc_meta)] = len(mc_jobs)
        tu_jobs = [TuJob(p_coef + i * 2048, p_pic[c_idx] + y * pic[c_idx].strides[0] + x * px, pic[c_idx].strides[0], log2, lim, kind, 0)
                   for i, (c_idx, x, y, log2, kind, lim) in enumerate(tu_meta)]
        jobs = []
        for i, (X, Y, partial, first_mc, first_tu) in enumerate(ctbs):
            nxt_mc = ctbs[i + 1][3] if i + 1 < len(ctbs) else len(mc_meta)
            nxt_tu = ctbs[i + 1][4] if i + 1 < len(ctbs) else len(tu_meta)
            j = CtbJob()
            for pl in range(3):
                sh = 1 if pl else 0
                j.dst[pl] = p_pic[pl] + (Y >> sh) * pic[pl].strides[0] + (X >> sh) * px
                j.stride[pl] = pic[pl].strides[0]
            j.width, j.height, j.log2_ctb_size = 64, 64, 6
            # without the flag the block is not read: every sample must then be covered (a dropped prediction block leaves a hole)
            j.flags = 1 if partial else 0
            j.first_mc, j.n_mc = ctb_first_mc[first_mc], ctb_first_mc[nxt_mc] - ctb_first_mc[first_mc]
            j.first_tu, j.n_tu = first_tu, nxt_tu - first_tu
            jobs.append(j)
        lib = prov.lib
        lib.mi355_hevc_recon_ctbs_dev.restype = C.c_int
        lib.mi355_hevc_recon_ctbs_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_void_p]
        lib.mi355_error_word_take.restype = C.c_uint
        p_jobs, p_mc, p_tu = d.up_jobs(jobs), d.up_jobs(mc_jobs), d.up_jobs(tu_jobs)
        if form == "levels":
            # the same jobs through mi355_hevc_recon_levels_dev: the prediction jobs as level 0, then the transform units (they add to what level 0 wrote) as
            # levels of 1 .. 5 units — a few hundred levels in one launch, each waiting for all before it
            lib.mi355_hevc_recon_levels_dev.restype = C.c_int
            lib.mi355_hevc_recon_levels_dev.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
            levels, wg, t = [Level(0, 0, len(mc_jobs), 0, 0, 0, 0, 0)], len(mc_jobs), 0
            while t < len(tu_jobs):
                n = min(r.randint(1, 5), len(tu_jobs) - t)
                levels.append(Level(wg, 0, 0, t, n, 0, 0, 0))
                wg += (n + 1) // 2
                t += n
            assert lib.mi355_hevc_recon_levels_dev(d.up_jobs(levels), len(levels), wg, p_mc, p_tu, None, None, None, bd, None) == 0
            assert lib.mi355_sync(None) == 0
            got = [d.down(p_pic[pl], pic[pl]) for pl in range(3)]
            for pl in range(3):
                bad = np.argwhere(got[pl] != pic_o[pl])
                assert bad.size == 0, "recon_levels: plane %d differs at %d samples, first (y, x) = %s (bd %d)" % (pl, len(bad), bad[0], bd)
            return len(levels)
        if promise_check:
            # the caller's promise (MI355_HEVC_RECON_UNIFORM) on a list that breaks it: the blocks of other shapes are left as they were and the device says so
            lib.mi355_error_word_take()
            assert lib.mi355_hevc_recon_ctbs_dev(p_jobs, len(jobs), p_mc, p_tu, bd, 1, None) == 0
            assert lib.mi355_sync(None) == -5, "a block outside the promised shapes must be reported (MI355_E_DEVICE_FAULT)"
            assert lib.mi355_error_word_take() == 2 and lib.mi355_sync(None) == 0
            got = [d.down(p_pic[pl], pic[pl]) for pl in range(3)]
            n_uniform = 0
            for (X, Y, partial, first_mc, first_tu), is_u in zip(ctbs, uniform_ctbs):
                for pl in range(3):
                    sh = 1 if pl else 0
                    sl = (slice(Y >> sh, (Y + 64) >> sh), slice(X >> sh, (X + 64) >> sh))
                    want = pic_o[pl][sl] if is_u else pic[pl][sl]
                    assert np.array_equal(got[pl][sl], want), "promise check: block (%d, %d) plane %d" % (X, Y, pl)
                n_uniform += is_u
            assert 0 < n_uniform < len(ctbs)
        assert lib.mi355_hevc_recon_ctbs_dev(p_jobs, len(jobs), p_mc, p_tu, bd, 0, None) == 0
        assert lib.mi355_sync(None) == 0
        got = [d.down(p_pic[pl], pic[pl]) for pl in range(3)]
    finally:
        d.free()
    for pl in range(3):
        bad = np.argwhere(got[pl] != pic_o[pl])
        assert bad.size == 0, "recon_ctbs: plane %d differs at %d samples, first (y, x) = %s (bd %d)" % (pl, len(bad), bad[0], bd)
    return len(mc_jobs) + len(tu_jobs)


class _LevelLib:
    """the library with its prediction / transform-unit batches routed through mi355_hevc_recon_level_dev (one launch for a dependency level's
    job kinds: here one kind at a time, the other two empty)"""

    def __init__(self, lib):
        self._lib = lib
        lib.mi355_hevc_recon_level_dev.restype = C.c_int
        lib.mi355_hevc_recon_level_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]

    def __getattr__(self, name):
        return getattr(self._lib, name)

    def mi355_hevc_mcpred_batch_dev(self, jobs, n, bd, stream):
        return self._lib.mi355_hevc_recon_level_dev(jobs, n, None, 0, None, None, None, 0, bd, stream)

    def mi355_hevc_residual_batch_dev(self, jobs, n, bd, stream):
        return self._lib.mi355_hevc_recon_level_dev(None, 0, jobs, n, None, None, None, 0, bd, stream)


class _LevelProv:
    def __init__(self, prov):
        self.lib = _LevelLib(prov.lib)


def check_level_mcpred(prov, oracle, bd, seed):
    return check_mcpred(_LevelProv(prov), oracle, bd, seed)


def check_level_residual(prov, oracle, bd, seed):
    return check_residual(_LevelProv(prov), oracle, bd, seed, cells=(5, 7))      # an odd number of units: the last workgroup holds one


def check_recon_levels(prov, oracle, bd, seed):
    return check_recon_ctbs(prov, oracle, bd, seed, promise_check=False, form="levels")


CHECKS = {"residual": check_residual, "mc": check_mc, "pred": check_pred, "deblock": check_deblock, "sao": check_sao,
          "intra": check_intra, "mcpred": check_mcpred, "sao_ctbs": check_sao_ctbs, "edge_emu": check_edge_emu,
          "level_mcpred": check_level_mcpred, "level_residual": check_level_residual, "recon_ctbs": check_recon_ctbs, "recon_levels": check_recon_levels}


def levels_wait_expiry_in_subprocess(provider_name):
    """mi355_hevc_recon_levels_dev's bounded wait (hevc_batch.hip: k_hevc_recon_levels): with the bound at zero (MI355_LEVELS_NAPS_MAX=0, read once per process: a process of its
    own) nobody counts itself done and every workgroup behind level 0 gives up at once — the entry point returns 0 (the launch was made), the wait behind it
    MI355_E_DEVICE_FAULT, the error word says MI355_ERR_WAIT_EXPIRED, and the word once taken the device is usable again."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, ctypes as C\n"
        "sys.path.insert(0, %r)\n"
        "import providers, hevc_batch\n"
        "prov = providers.%s()\n"
        "lib = prov.lib\n"
        "lib.mi355_error_word_take.restype = C.c_uint\n"
        "lib.mi355_error_word_take()\n"
        "try:\n"
        "    hevc_batch.check_recon_levels(prov, providers.oracle(), 8, 5)\n"
        "    print('RESULT no assertion')\n"
        "except AssertionError:\n"
        "    word = lib.mi355_error_word_take()\n"
        "    print('RESULT', word, lib.mi355_sync(None))\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), provider_name)
    env = dict(os.environ, MI355_LEVELS_NAPS_MAX="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert line[1:] == ["1", "0"], line
