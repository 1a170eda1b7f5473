"""checkasm-style differential cases for the 9- and 10-bit H.264 DSP tables (h264dsp.c:37-47 instantiates
the templates for BIT_DEPTH 8 / 9 / 10).  Same structure as cases_h264.py — every pointer of H264DSPContext /
H264QpelContext / H264ChromaContext / H264PredContext / VideoDSPContext is driven with seeded inputs through the C ABI and
`run_all(provider, bd, seed)` returns {case_name: bytes_of_every_output_buffer} — with 16-bit samples (`pixel` = uint16_t,
values below 1 << bd), 32-bit coefficients behind the tables' int16_t pointers, strides in bytes.

There is no CPU restatement for these tables: the checker is the reference itself (oracle/_ref/libref.so on the build
machine, golden sha1s made by it — tests/golden/h264dsp_hbd_ref_sha1.json — everywhere else)."""
import ctypes as C
from collections import OrderedDict

import numpy as np

import abi_ctypes as A
from cases_h264 import scan8
from rng import SplitMix64


def pp(a, off=0):
    """pointer to sample `off` of a uint16 array, as the tables' uint8_t *"""
    return C.cast(a.ctypes.data + 2 * off, A.u8p)


def pc(a, off=0):
    """pointer to coefficient `off` of an int32 array, as the tables' int16_t *"""
    return C.cast(a.ctypes.data + 4 * off, A.i16p)


def pi8(a):
    return C.cast(a.ctypes.data, A.i8p)


def pint(a):
    return C.cast(a.ctypes.data, A.intp)


WILD = False   # group "wild": samples anywhere in 16 bits, as transform-bypass streams can leave them in a plane (their residual adds do not clip)


def pix(r, shape, bd):
    return r.randint(0, 65535 if WILD else (1 << bd) - 1, shape).astype(np.uint16)


def edge_pixels(r, shape, axis, pos, bd):
    """smooth field with a step at the edge so the filter conditions fire often"""
    sc = 1 << (bd - 8)
    base = r.randint(40, 200) * sc
    a = base + r.randint(-6 * sc, 6 * sc, shape)
    step = r.randint(-12 * sc, 12 * sc)
    idx = [slice(None)] * 2
    idx[axis] = slice(pos, None)
    a[tuple(idx)] += step
    mask = r.randint(0, 7, shape) == 0
    if WILD:    # the same field lifted out of the bit depth's range (the conditions still fire), strays anywhere in 16 bits
        a = np.where(mask, r.randint(0, 65535, shape), a + r.randint(1 << bd, 60000))
        return np.clip(a, 0, 65535).astype(np.uint16)
    a = np.where(mask, r.randint(0, (1 << bd) - 1, shape), a)
    return np.clip(a, 0, (1 << bd) - 1).astype(np.uint16)


def coeffs(r, n, kind, bd):
    if kind == "small":
        return r.laplace_int(24 << (bd - 8), n, (2048 << (bd - 8)) - 1).astype(np.int32)
    if kind == "dconly":
        v = np.zeros(n, np.int32)
        v[0] = r.randint(-2047 << (bd - 8), 2047 << (bd - 8))
        return v
    return r.randint(-(1 << 18), 1 << 18, n).astype(np.int32)      # beyond 16 bits: the 32-bit coefficient type


def block_offsets(stride_bytes):
    # h264_slice.c:485-494 for pixel_shift 1: (4 * x << 1) + 4 * y * linesize
    off = np.zeros(48, np.int32)
    for i in range(16):
        x = (i & 1) + 2 * ((i >> 2) & 1)
        y = ((i >> 1) & 1) + 2 * (i >> 3)
        off[i] = off[16 + i] = off[32 + i] = 8 * x + 4 * y * stride_bytes
    return off


# ------------------------------------------------------------------ H264DSPContext
def cases_idct(c, r, out, bd):
    for name, fn, sz in (("idct_add", c.h264_idct_add, 4), ("idct8_add", c.h264_idct8_add, 8),
                         ("idct_dc_add", c.h264_idct_dc_add, 4), ("idct8_dc_add", c.h264_idct8_dc_add, 8)):
        for kind in ("small", "full", "dconly"):
            for align in range(0, 16, sz):
                for rep in range(2):
                    dst = pix(r, (24, 32), bd)
                    blk = np.zeros(64 + 16, np.int32)
                    blk[: sz * sz] = coeffs(r, sz * sz, kind, bd)
                    blk[sz * sz:] = 0x55
                    if fn:
                        fn(pp(dst, 4 * 32 + align), pc(blk), 64)
                        out["%s/%s/a%d/%d" % (name, kind, align, rep)] = dst.tobytes() + blk.tobytes()


def cases_idct_multi(c, r, out, bd, idc=1):
    stride = 48
    off = block_offsets(2 * stride)
    names = ("h264_idct_add16", "h264_idct_add16intra", "h264_idct8_add4", "h264_idct_add8") if idc == 1 else ("h264_idct_add8",)
    for name in names:
        fn = getattr(c, name)
        for rep in range(10):
            consistent = rep < 7
            nnzc = np.zeros(15 * 8, np.uint8)
            blk = np.zeros(16 * 48, np.int32)
            step = 4 if name == "h264_idct8_add4" else 1
            rng = range(16, 48) if name == "h264_idct_add8" else range(0, 16, step)
            for i in rng:
                per_plane = 4 if idc == 1 else 8
                if name == "h264_idct_add8" and (i & 15) >= per_plane:
                    continue
                n = 16 * step
                mode = r.randint(0, 3)
                if mode == 0:
                    nnz = 0
                    if name in ("h264_idct_add16intra", "h264_idct_add8") and r.randint(0, 1):
                        blk[i * 16] = r.randint(-2047, 2047) << (bd - 8)
                elif mode == 1:
                    nnz = 1
                    blk[i * 16] = r.randint(-2047, 2047) << (bd - 8)
                else:
                    nnz = r.randint(2, 16)
                    blk[i * 16: i * 16 + n] = coeffs(r, n, "small", bd)
                if not consistent:
                    nnz = r.randint(0, 2)
                    blk[i * 16: i * 16 + n] = coeffs(r, n, "small", bd) * (r.randint(0, 3) > 0)
                # 4:2:2: the second four blocks of a plane are counted at scan8[i + 4] (h264idct_template.c:216-238)
                k = i if (idc == 1 or (i & 15) < 4) else i + 4
                nnzc[scan8(k)] = nnz
            planes = [pix(r, (32, stride), bd) for _ in range(2)]
            if not fn:
                continue
            if name == "h264_idct_add8":
                arr = (A.u8p * 2)(pp(planes[0], 4 * stride + 8), pp(planes[1], 4 * stride + 8))
                fn(arr, pint(off), pc(blk), 2 * stride, C.cast(nnzc.ctypes.data, A.u8p))
            else:
                fn(pp(planes[0], 4 * stride + 16), pint(off), pc(blk), 2 * stride, C.cast(nnzc.ctypes.data, A.u8p))
            out["%s%s/%d" % (name, "" if idc == 1 else "_422", rep)] = planes[0].tobytes() + planes[1].tobytes() + blk.tobytes()


def cases_dc(c, r, out, bd, idc=1):
    for rep in range(18):
        qmul = int([16, 64, 208, 1024, 4096, 13 * 512][rep % 6])
        inp = coeffs(r, 16, "small", bd) if rep < 12 else r.randint(-32768, 32767, 16).astype(np.int32)
        if idc == 1:
            outb = np.full(256, 0x1234, np.int32)
            if c.h264_luma_dc_dequant_idct:
                c.h264_luma_dc_dequant_idct(pc(outb), pc(inp), qmul)
                out["luma_dc/%d" % rep] = outb.tobytes() + inp.tobytes()
            blk = np.full(64, 0x0777, np.int32)
            blk[[0, 16, 32, 48]] = inp[:4]
            if c.h264_chroma_dc_dequant_idct:
                c.h264_chroma_dc_dequant_idct(pc(blk), qmul)
                out["chroma_dc/%d" % rep] = blk.tobytes()
        else:
            blk = np.full(128, 0x0777, np.int32)
            blk[[0, 16, 32, 48, 64, 80, 96, 112]] = inp[:8]
            if c.h264_chroma_dc_dequant_idct:
                c.h264_chroma_dc_dequant_idct(pc(blk), qmul)
                out["chroma422_dc/%d" % rep] = blk.tobytes()


def cases_addpx(c, r, out, bd):
    for name, fn, sz in (("add_pixels4", c.h264_add_pixels4_clear, 4), ("add_pixels8", c.h264_add_pixels8_clear, 8)):
        for rep in range(4):
            dst = pix(r, (16, 32), bd)
            blk = r.randint(-(1 << bd) + 1, (1 << bd) - 1, sz * sz).astype(np.int32)
            if not fn:
                continue
            fn(pp(dst, 4 * 32 + 8), pc(blk), 64)
            out["%s/%d" % (name, rep)] = dst.tobytes() + blk.tobytes()


def cases_weight(c, r, out, bd):
    for idx, w in enumerate((16, 8, 4, 2)):
        for rep in range(10):
            h = [16, 8, 4, 2][r.randint(0, 3)] if w < 16 else [16, 8][r.randint(0, 1)]
            ld = r.randint(0, 7)
            wt, wt2, off = r.randint(-128, 127), r.randint(-128, 127), r.randint(-128, 127)
            if rep == 0:
                ld, wt, wt2, off = 5, 32, 32, 0
            fn = c.weight_h264_pixels_tab[idx]
            blk = pix(r, (20, 32), bd)
            if fn:
                fn(pp(blk, 2 * 32 + 8), 64, h, ld, wt, off)
                out["weight%d/%d" % (w, rep)] = blk.tobytes()
            fn = c.biweight_h264_pixels_tab[idx]
            dst, src = pix(r, (20, 32), bd), pix(r, (20, 32), bd)
            if fn:
                fn(pp(dst, 2 * 32 + 8), pp(src, 2 * 32 + 8), 64, h, ld, wt, wt2, off)
                out["biweight%d/%d" % (w, rep)] = dst.tobytes() + src.tobytes()


def cases_loopfilter(c, r, out, bd, idc=1):
    triples = []
    a, b, t = 255.0, 18.0, 25.0
    for _ in range(24):
        triples.append((int(a), int(b), int(t)))
        a, b, t = a * 0.88, b * 0.9, t * 0.88
    if idc == 1:
        specs = [
            ("h264_v_loop_filter_luma", 0, 16, 1), ("h264_h_loop_filter_luma", 1, 16, 1),
            ("h264_h_loop_filter_luma_mbaff", 1, 8, 1),
            ("h264_v_loop_filter_luma_intra", 0, 16, 0), ("h264_h_loop_filter_luma_intra", 1, 16, 0),
            ("h264_h_loop_filter_luma_mbaff_intra", 1, 8, 0),
            ("h264_v_loop_filter_chroma", 0, 8, 1), ("h264_h_loop_filter_chroma", 1, 8, 1),
            ("h264_h_loop_filter_chroma_mbaff", 1, 4, 1),
            ("h264_v_loop_filter_chroma_intra", 0, 8, 0), ("h264_h_loop_filter_chroma_intra", 1, 8, 0),
            ("h264_h_loop_filter_chroma_mbaff_intra", 1, 4, 0),
        ]
    else:   # 4:2:2: the horizontal-filter slots hold the 16-line forms (h264dsp.c:113-130)
        specs = [("h264_h_loop_filter_chroma", 1, 16, 1), ("h264_h_loop_filter_chroma_mbaff", 1, 8, 1),
                 ("h264_h_loop_filter_chroma_intra", 1, 16, 0), ("h264_h_loop_filter_chroma_mbaff_intra", 1, 8, 0)]
    stride = 32
    for name, vertical_edge, length, has_tc in specs:
        fn = getattr(c, name)
        for k, (al, be, t0) in enumerate(triples):
            buf = edge_pixels(r, (32, stride), 1 if vertical_edge else 0, 8, bd)
            tc = np.array([r.randint(-1, max(t0, 0)) for _ in range(4)], np.int8)
            if k % 5 == 0:
                tc[:] = t0
            off = 8 * stride + 8
            if not fn:
                continue
            if has_tc:
                fn(pp(buf, off), 2 * stride, al, be, pi8(tc))
            else:
                fn(pp(buf, off), 2 * stride, al, be)
            out["%s%s/%d" % (name, "" if idc == 1 else "_422", k)] = buf.tobytes()


# ------------------------------------------------------------------ qpel / chroma / videodsp
def cases_qpel(q, r, out, bd):
    mx = (1 << bd) - 1
    r2 = SplitMix64(0x0b7 + bd)          # its own stream for the out-of-range repetitions: the cases before them keep their values
    for tabname, tab, nsz in (("put", q.put_h264_qpel_pixels_tab, 4), ("avg", q.avg_h264_qpel_pixels_tab, 3)):
        for si in range(nsz):
            size = 16 >> si
            for pos in range(16):
                fn = tab[si][pos]
                for rep in range(5):
                    stride = 32
                    if rep == 0:
                        src = pix(r, (32, stride), bd)
                    elif rep >= 3:
                        # samples OUTSIDE the bit depth's range (transform-bypass streams leave them in a plane: add_pixels does not
                        # clip): the reference keeps the first pass of the 2-D positions in int16_t (+ the 10-bit bias) and wraps,
                        # h264qpel_template.c:119-146.  rep 3: a few times the largest value, rep 4: any 16-bit value
                        src = r2.randint(0, 4 * mx + 3 if rep == 3 else 65535, (32, stride)).astype(np.uint16)
                    elif rep == 1:
                        src = edge_pixels(r, (32, stride), 1, 11, bd)
                    else:
                        # extremes: columns / rows alternating between 0 and the largest value drive the unclipped sums
                        # (and the 10-bit `pad` of the hv pass, h264qpel_template.c:122) to their limits
                        hi = np.tile(np.array([mx, 0, mx, mx, 0, mx], np.int64), 6)[:32]
                        rows = [hi, mx - hi, hi, hi, mx - hi, hi]
                        src = np.stack([np.roll(rows[(y + pos) % 6], pos % 6) for y in range(32)]).astype(np.uint16)
                    dst = pix(r if rep < 3 else r2, (32, stride), bd)
                    keep = src.copy()
                    if not fn:
                        continue
                    fn(pp(dst, 4 * stride + (size if size < 16 else 0)), pp(src, 5 * stride + 5), 2 * stride)
                    assert (src == keep).all()
                    out["qpel_%s%d/%d/%d" % (tabname, size, pos, rep)] = dst.tobytes()


def cases_chroma(ch, r, out, bd):
    for tabname, tab in (("put", ch.put_h264_chroma_pixels_tab), ("avg", ch.avg_h264_chroma_pixels_tab)):
        for wi, w in enumerate((8, 4, 2)):
            fn = tab[wi]
            for rep in range(16):
                stride = 32
                h = [2, 4, 8, 16][r.randint(0, 3)]
                x, y = r.randint(0, 7), r.randint(0, 7)
                if rep < 3:
                    x, y = [(0, 0), (3, 0), (0, 5)][rep]
                src, dst = pix(r, (24, stride), bd), pix(r, (24, stride), bd)
                if not fn:
                    continue
                fn(pp(dst, 2 * stride + 8), pp(src, 2 * stride + 3), 2 * stride, h, x, y)
                out["chroma_%s%d/%d" % (tabname, w, rep)] = dst.tobytes()


def cases_videodsp(v, r, out, bd):
    if not v.emulated_edge_mc:
        return
    W, H, ls = 48, 40, 64
    plane = pix(r, (H, ls), bd)
    for rep in range(25):
        bw, bh = [(21, 21), (9, 9), (9, 17), (4, 4), (71, 71)][rep % 5]
        sx, sy = r.randint(-bw - 4, W + 4), r.randint(-bh - 4, H + 4)
        if rep % 4 == 0:
            sx, sy = r.randint(-bw + 1, W - 1), r.randint(-bh + 1, H - 1)
        buf = np.full((80, 96), 0xA5A5, np.uint16)
        base = plane.ctypes.data + 2 * (sy * ls + sx)
        v.emulated_edge_mc(pp(buf), C.cast(base, A.u8p), 2 * 96, 2 * ls, bw, bh, sx, sy, W, H)
        out["emu_edge/%d" % rep] = buf.tobytes()


# ------------------------------------------------------------------ intra prediction
def cases_pred(h, r, out, bd, idc=1):
    stride = 48
    if idc == 1:
        for mode in range(12):
            fn = h.pred4x4[mode]
            for rep in range(4):
                buf = pix(r, (24, stride), bd)
                tr_ext = pix(r, 8, bd)
                off = 8 * stride + 16
                if rep & 1:
                    tr_ext[:4] = buf[7, 16 + 3]
                    tr = pp(tr_ext)
                else:
                    tr = pp(buf, off + 4 - stride)
                if not fn:
                    continue
                fn(pp(buf, off), tr, 2 * stride)
                out["pred4x4/%d/%d" % (mode, rep)] = buf.tobytes()
        for mode in range(12):
            fn = h.pred8x8l[mode]
            for rep in range(4):
                buf = pix(r, (24, stride), bd)
                if not fn:
                    continue
                fn(pp(buf, 8 * stride + 16), (rep & 1) * 0x8000, (rep >> 1) * 0x4000, 2 * stride)
                out["pred8x8l/%d/%d" % (mode, rep)] = buf.tobytes()
        for mode in range(7):
            fn = h.pred16x16[mode]
            for rep in range(3):
                buf = pix(r, (32, stride), bd) if rep else edge_pixels(r, (32, stride), 1, 20, bd)
                if not fn:
                    continue
                fn(pp(buf, 8 * stride + 16), 2 * stride)
                out["pred16x16/%d/%d" % (mode, rep)] = buf.tobytes()
    for mode in range(11):
        fn = h.pred8x8[mode]
        for rep in range(3):
            buf = pix(r, (32, stride), bd) if rep else edge_pixels(r, (32, stride), 0, 4, bd)
            if not fn:
                continue
            fn(pp(buf, 8 * stride + 16), 2 * stride)
            out["pred8x8%s/%d/%d" % ("" if idc == 1 else "_422", mode, rep)] = buf.tobytes()


def cases_pred_add(h, r, out, bd, idc=1):
    stride = 48
    lim = (1 << bd) - 1
    if idc == 1:
        for name, tab, n in (("pred4x4_add", h.pred4x4_add, 4), ("pred8x8l_add", h.pred8x8l_add, 8)):
            for d in range(2):
                for rep in range(3):
                    buf = pix(r, (24, stride), bd)
                    blk = r.randint(-lim, lim, n * n).astype(np.int32)
                    if not tab[d]:
                        continue
                    tab[d](pp(buf, 8 * stride + 16), pc(blk), 2 * stride)
                    out["%s/%d/%d" % (name, d, rep)] = buf.tobytes() + blk.tobytes()
        for d in range(2):
            for rep in range(4):
                buf = pix(r, (24, stride), bd)
                blk = r.randint(-lim, lim, 64).astype(np.int32)
                if not h.pred8x8l_filter_add[d]:
                    continue
                h.pred8x8l_filter_add[d](pp(buf, 8 * stride + 16), pc(blk), (rep & 1) * 0x8000, (rep >> 1) * 0x4000, 2 * stride)
                out["pred8x8l_filter_add/%d/%d" % (d, rep)] = buf.tobytes() + blk.tobytes()
    sb = 2 * stride
    o16 = np.array([8 * (i & 1) + 16 * ((i >> 2) & 1) + (4 * ((i >> 1) & 1) + 8 * (i >> 3)) * sb for i in range(16)], np.int32)
    if idc == 1:
        specs = (("pred8x8_add", h.pred8x8_add, 4, np.array([0, 8, 4 * sb, 4 * sb + 8] + [0] * 12, np.int32)), ("pred16x16_add", h.pred16x16_add, 16, o16))
    else:   # pred8x16_*_add: blocks 0..3 at block_offset[0..3], 4..7 at block_offset[8..11] (h264pred_template.c:1326-1354)
        o = np.zeros(16, np.int32)
        o[:4] = [0, 8, 4 * sb, 4 * sb + 8]
        o[8:12] = [8 * sb, 8 * sb + 8, 12 * sb, 12 * sb + 8]
        specs = (("pred8x16_add", h.pred8x8_add, 8, o),)
    for name, tab, nblk, offs in specs:
        for d in (1, 2):
            for rep in range(3):
                buf = pix(r, (32, stride), bd)
                blk = r.randint(-lim, lim, 16 * nblk).astype(np.int32)
                if not tab[d]:
                    continue
                tab[d](pp(buf, 8 * stride + 16), C.cast(offs.ctypes.data, A.intp), pc(blk), 2 * stride)
                out["%s/%d/%d" % (name, d, rep)] = buf.tobytes() + blk.tobytes()


GROUPS = ("idct", "idct_multi", "dc", "addpx", "weight", "loopfilter", "qpel", "chroma", "videodsp", "pred", "pred_add", "422", "wild")


def run_group(provider, group, bd, seed=0x2640):
    global WILD
    if group == "wild":     # the sample-reading entries once more on out-of-range planes (the 2-D qpel positions have theirs in "qpel")
        out = OrderedDict()
        WILD = True
        try:
            for g in ("idct", "addpx", "weight", "loopfilter", "chroma", "videodsp", "pred", "pred_add"):
                for k, v in run_group(provider, g, bd, seed + 77).items():
                    out[g + "." + k] = v
        finally:
            WILD = False
        return out
    out = OrderedDict()
    r = SplitMix64(seed + 1000 * bd + sum(ord(ch) for ch in group))
    if group in ("idct", "idct_multi", "dc", "addpx", "weight", "loopfilter"):
        c = provider.h264dsp(bd, 1)
        {"idct": cases_idct, "idct_multi": cases_idct_multi, "dc": cases_dc, "addpx": cases_addpx, "weight": cases_weight,
         "loopfilter": cases_loopfilter}[group](c, r, out, bd)
    elif group == "qpel":
        cases_qpel(provider.h264qpel(bd), r, out, bd)
    elif group == "chroma":
        cases_chroma(provider.h264chroma(bd), r, out, bd)
    elif group == "videodsp":
        cases_videodsp(provider.videodsp(bd), r, out, bd)
    elif group == "pred":
        cases_pred(provider.h264pred(bd, 1), r, out, bd)
    elif group == "pred_add":
        cases_pred_add(provider.h264pred(bd, 1), r, out, bd)
    elif group == "422":
        c = provider.h264dsp(bd, 2)
        cases_idct_multi(c, r, out, bd, idc=2)
        cases_dc(c, r, out, bd, idc=2)
        cases_loopfilter(c, r, out, bd, idc=2)
        h = provider.h264pred(bd, 2)
        cases_pred(h, r, out, bd, idc=2)
        cases_pred_add(h, r, out, bd, idc=2)
    return out


def run_all(provider, bd, seed=0x2640, groups=None):
    out = OrderedDict()
    for g in groups or GROUPS:
        for k, v in run_group(provider, g, bd, seed).items():
            out["%s:%s" % (g, k)] = v
    return out
