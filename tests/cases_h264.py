"""checkasm-style differential cases for the H.264 DSP tables.

`run_all(provider, seed)` drives every pointer of H264DSPContext / H264QpelContext /
H264ChromaContext / H264PredContext / VideoDSPContext with seeded inputs through
the C ABI and returns {case_name: bytes_of_every_output_buffer}.  Buffer shapes,
strides, alignment sweeps and parameter ranges follow tests/checkasm/h264dsp.c,
h264qpel.c, h264pred.c of the reference; functions the reference has no checkasm
for (chroma MC, weight/biweight, videodsp) get the same treatment.

Outputs include the guard bands around every destination and the coefficient
blocks (which must be cleared identically), like checkasm's memcmp of whole buffers.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

import abi_ctypes as A
from rng import SplitMix64


def p8(a, off=0):
    return C.cast(a.ctypes.data + off, A.u8p)


def p16(a, off=0):
    return C.cast(a.ctypes.data + 2 * off, A.i16p)


def pi8(a):
    return C.cast(a.ctypes.data, A.i8p)


def pint(a):
    return C.cast(a.ctypes.data, A.intp)


def scan8(i):
    p, b = i >> 4, i & 15
    x = (b & 1) + 2 * ((b >> 2) & 1)
    y = ((b >> 1) & 1) + 2 * (b >> 3)
    return 4 + x + 8 * (1 + y + 5 * p)


def _coeffs(r, n, kind):
    if kind == "small":
        return r.laplace_int(24, n, 2047).astype(np.int16)
    if kind == "dconly":
        v = np.zeros(n, np.int16)
        v[0] = r.randint(-2047, 2047)
        return v
    return r.randint(-32768, 32767, n).astype(np.int16)  # full range: exercises int16 wrap


# ------------------------------------------------------------------ H264DSPContext
def cases_idct(c, r, out):
    for name, fn, sz in (("idct_add", c.h264_idct_add, 4), ("idct8_add", c.h264_idct8_add, 8),
                         ("idct_dc_add", c.h264_idct_dc_add, 4), ("idct8_dc_add", c.h264_idct8_dc_add, 8)):
        for kind in ("small", "full", "dconly"):
            for align in range(0, 16, sz):
                for rep in range(3):
                    dst = r.u8((24, 32))
                    blk = np.zeros(64 + 16, np.int16)
                    blk[: sz * sz] = _coeffs(r, sz * sz, kind)
                    blk[sz * sz:] = 0x55
                    if fn:   # inputs are drawn even when a provider leaves the slot empty (keeps streams in step)
                        fn(p8(dst, 4 * 32 + align), p16(blk), 32)
                        out["%s/%s/a%d/%d" % (name, kind, align, rep)] = dst.tobytes() + blk.tobytes()


def _block_offsets(stride):
    # h264_slice.c:485-494: 4*x + 4*y*linesize for the 16 luma blocks, then chroma (4:2:0: 4 blocks/plane)
    off = np.zeros(48, np.int32)
    for i in range(16):
        x = (i & 1) + 2 * ((i >> 2) & 1)
        y = ((i >> 1) & 1) + 2 * (i >> 3)
        off[i] = 4 * x + 4 * y * stride
        off[16 + i] = off[32 + i] = 4 * x + 4 * y * stride
    return off


def cases_idct_multi(c, r, out):
    stride = 48
    off = _block_offsets(stride)
    for name in ("h264_idct_add16", "h264_idct_add16intra", "h264_idct8_add4", "h264_idct_add8"):
        fn = getattr(c, name)
        for rep in range(12):
            consistent = rep < 8
            nnzc = np.zeros(15 * 8, np.uint8)
            blk = np.zeros(16 * 48, np.int16)
            step = 4 if name == "h264_idct8_add4" else 1
            rng = range(16, 48) if name == "h264_idct_add8" else range(0, 16, step)
            for i in rng:
                if name == "h264_idct_add8" and (i & 15) >= 4:
                    continue
                n = 16 * step
                mode = r.randint(0, 3)
                if mode == 0:
                    nnz = 0
                    if name in ("h264_idct_add16intra", "h264_idct_add8") and r.randint(0, 1):
                        blk[i * 16] = r.randint(-2047, 2047)
                elif mode == 1:
                    nnz = 1
                    blk[i * 16] = r.randint(-2047, 2047)
                else:
                    nnz = r.randint(2, 16)
                    blk[i * 16: i * 16 + n] = _coeffs(r, n, "small")
                if not consistent:   # adversarial: counts that disagree with the block contents
                    nnz = r.randint(0, 2)
                    blk[i * 16: i * 16 + n] = _coeffs(r, n, "small") * (r.randint(0, 3) > 0)
                nnzc[scan8(i)] = nnz
            planes = [r.u8((24, stride)) for _ in range(2)]
            if not fn:
                continue
            if name == "h264_idct_add8":
                arr = (A.u8p * 2)(p8(planes[0], 4 * stride + 8), p8(planes[1], 4 * stride + 8))
                fn(arr, pint(off), p16(blk), stride, p8(nnzc))
            else:
                fn(p8(planes[0], 4 * stride + 16), pint(off), p16(blk), stride, p8(nnzc))
            out["%s/%d" % (name, rep)] = planes[0].tobytes() + planes[1].tobytes() + blk.tobytes()


def cases_dc(c, r, out):
    for rep in range(24):
        qmul = int([16, 64, 208, 1024, 4096, 13 * 512][rep % 6])
        inp = _coeffs(r, 16, "small" if rep < 16 else "full")
        outb = np.full(256, 0x1234, np.int16)
        if c.h264_luma_dc_dequant_idct:
            c.h264_luma_dc_dequant_idct(p16(outb), p16(inp), qmul)
            out["luma_dc/%d" % rep] = outb.tobytes() + inp.tobytes()
        blk = np.full(64, 0x0777, np.int16)
        blk[[0, 16, 32, 48]] = _coeffs(r, 4, "small" if rep < 16 else "full")
        if c.h264_chroma_dc_dequant_idct:
            c.h264_chroma_dc_dequant_idct(p16(blk), qmul)
            out["chroma_dc/%d" % rep] = blk.tobytes()


def cases_addpx(c, r, out):
    for name, fn, sz in (("add_pixels4", c.h264_add_pixels4_clear, 4), ("add_pixels8", c.h264_add_pixels8_clear, 8)):
        for rep in range(4):
            dst = r.u8((16, 32))
            blk = r.randint(-255, 255, sz * sz).astype(np.int16)
            if not fn:
                continue
            fn(p8(dst, 4 * 32 + 8), p16(blk), 32)
            out["%s/%d" % (name, rep)] = dst.tobytes() + blk.tobytes()


def cases_weight(c, r, out):
    for idx, w in enumerate((16, 8, 4, 2)):
        for rep in range(10):
            h = [16, 8, 4, 2][r.randint(0, 3)] if w < 16 else [16, 8][r.randint(0, 1)]
            ld = r.randint(0, 7)
            wt, wt2, off = r.randint(-128, 127), r.randint(-128, 127), r.randint(-128, 127)
            if rep == 0:
                ld, wt, wt2, off = 5, 32, 32, 0
            fn = c.weight_h264_pixels_tab[idx]
            blk = r.u8((20, 32))
            if fn:
                fn(p8(blk, 2 * 32 + 8), 32, h, ld, wt, off)
                out["weight%d/%d" % (w, rep)] = blk.tobytes()
            fn = c.biweight_h264_pixels_tab[idx]
            dst, src = r.u8((20, 32)), r.u8((20, 32))
            if fn:
                fn(p8(dst, 2 * 32 + 8), p8(src, 2 * 32 + 8), 32, h, ld, wt, wt2, off)
                out["biweight%d/%d" % (w, rep)] = dst.tobytes() + src.tobytes()


def _edge_pixels(r, shape, axis, pos):
    """smooth field with a step at the edge so the filter conditions fire often"""
    base = r.randint(40, 200)
    a = base + r.randint(-6, 6, shape)
    step = r.randint(-12, 12)
    idx = [slice(None)] * 2
    idx[axis] = slice(pos, None)
    a[tuple(idx)] += step
    mask = r.randint(0, 7, shape) == 0       # sprinkle outliers
    a = np.where(mask, r.randint(0, 255, shape), a)
    return np.clip(a, 0, 255).astype(np.uint8)


def cases_loopfilter(c, r, out):
    # 36 (alpha,beta,tc0) triples decaying from (255,18,25): checkasm/h264dsp.c:317-340
    triples = []
    a, b, t = 255.0, 18.0, 25.0
    for _ in range(36):
        triples.append((int(a), int(b), int(t)))
        a, b, t = a * 0.9, b * 0.92, t * 0.9
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
    stride = 32
    for name, vertical_edge, length, has_tc in specs:
        fn = getattr(c, name)
        for k, (al, be, t0) in enumerate(triples):
            buf = _edge_pixels(r, (24, stride), 1 if vertical_edge else 0, 8)
            tc = np.array([r.randint(-1, max(t0, 0)) for _ in range(4)], np.int8)
            if k % 5 == 0:
                tc[:] = t0
            # edge at (8,8): v-filter spans columns 8..8+len, h-filter rows 8..8+len
            off = 8 * stride + 8
            if not fn:
                continue
            if has_tc:
                fn(p8(buf, off), stride, al, be, pi8(tc))
            else:
                fn(p8(buf, off), stride, al, be)
            out["%s/%d" % (name, k)] = buf.tobytes()


def cases_startcode(c, r, out):
    if not c.startcode_find_candidate:
        return
    for rep in range(6):
        # the reference scans eight bytes at a time and may look up to 7 bytes past `size` (its callers pad their
        # buffers with zeros, AV_INPUT_BUFFER_PADDING_SIZE): give it that padding, or the result of a buffer without
        # a zero byte depends on whatever follows it in memory
        buf = np.zeros(308, np.uint8)
        buf[:300] = r.randint(1, 255, 300).astype(np.uint8)
        if rep:
            buf[r.randint(0, 299)] = 0
        out["startcode/%d" % rep] = bytes([c.startcode_find_candidate(p8(buf), 300) & 0xFF])


# ------------------------------------------------------------------ qpel / chroma / videodsp
def cases_qpel(q, r, out):
    for tabname, tab, nsz in (("put", q.put_h264_qpel_pixels_tab, 4), ("avg", q.avg_h264_qpel_pixels_tab, 3)):
        for si in range(nsz):
            size = 16 >> si
            for pos in range(16):
                fn = tab[si][pos]
                for rep in range(2):
                    stride = 32
                    src = r.u8((32, stride)) if rep == 0 else _edge_pixels(r, (32, stride), rep & 1, 11)
                    dst = r.u8((32, stride))
                    keep = src.copy()
                    if not fn:
                        continue
                    fn(p8(dst, 4 * stride + (size if size < 16 else 0)), p8(src, 5 * stride + 5), stride)
                    assert (src == keep).all()
                    out["qpel_%s%d/%d/%d" % (tabname, size, pos, rep)] = dst.tobytes()


def cases_qpel_extreme(q, r, out):
    """Adversarial inputs for the centre (hv) positions: sample patterns that drive the unclipped first-pass sums to
    their extremes (rows of 255 0 255 255 0 255 ... give +10710, their complement -2550), stacked so that the second pass
    reaches +-475 000 — the values where a 16-bit formulation of the second pass would have to wrap or saturate."""
    base = np.array([255, 0, 255, 255, 0, 255], np.uint8)
    hi = np.tile(base, 6)[:32]                  # horizontal sum at the pattern's phase: 10710
    lo = 255 - hi                               # -2550
    patterns = {
        "max": [hi, lo, hi, hi, lo, hi],        # vertical taps (1,-5,20,20,-5,1) all aligned with the signs
        "min": [lo, hi, lo, lo, hi, lo],
        "flat255": [np.full(32, 255, np.uint8)] * 6,
        "checker": [hi, lo] * 3,
    }
    for pname, rows in patterns.items():
        for phase in range(6):
            src = np.zeros((32, 32), np.uint8)
            for y in range(32):
                src[y] = np.roll(rows[(y + phase) % 6], phase)
            for tabname, tab, sizes in (("put", q.put_h264_qpel_pixels_tab, (0, 1, 2)), ("avg", q.avg_h264_qpel_pixels_tab, (0,))):
                for si in sizes:
                    size = 16 >> si
                    for pos in (6, 9, 10, 11, 14, 5, 7, 13, 15):      # every position that involves the hv plane, + diagonals
                        fn = tab[si][pos]
                        if not fn:
                            continue
                        dst = r.u8((32, 32))
                        fn(p8(dst, 4 * 32 + (size if size < 16 else 0)), p8(src, 5 * 32 + 5), 32)
                        out["qpelx_%s_%s%d/%d/%d" % (pname, tabname, size, pos, phase)] = dst.tobytes()


def cases_chroma(ch, r, out):
    for tabname, tab in (("put", ch.put_h264_chroma_pixels_tab), ("avg", ch.avg_h264_chroma_pixels_tab)):
        for wi, w in enumerate((8, 4, 2)):
            fn = tab[wi]
            for rep in range(24):
                stride = 32
                h = [2, 4, 8, 16][r.randint(0, 3)]
                x, y = r.randint(0, 7), r.randint(0, 7)
                if rep < 3:
                    x, y = [(0, 0), (3, 0), (0, 5)][rep]
                src, dst = r.u8((24, stride)), r.u8((24, stride))
                if not fn:
                    continue
                fn(p8(dst, 2 * stride + 8), p8(src, 2 * stride + 3), stride, h, x, y)
                out["chroma_%s%d/%d" % (tabname, w, rep)] = dst.tobytes()


def cases_videodsp(v, r, out):
    if not v.emulated_edge_mc:
        return
    W, H, ls = 48, 40, 64
    plane = r.u8((H, ls))
    for rep in range(40):
        bw, bh = [(21, 21), (9, 9), (9, 17), (4, 4), (71, 71)][rep % 5]
        sx, sy = r.randint(-bw - 4, W + 4), r.randint(-bh - 4, H + 4)
        if rep % 4 == 0:   # at least partly inside like every real caller
            sx, sy = r.randint(-bw + 1, W - 1), r.randint(-bh + 1, H - 1)
        buf = np.full((80, 96), 0xA5, np.uint8)
        base = plane.ctypes.data + sy * ls + sx
        v.emulated_edge_mc(p8(buf), C.cast(base, A.u8p), 96, ls, bw, bh, sx, sy, W, H)
        out["emu_edge/%d" % rep] = buf.tobytes()


# ------------------------------------------------------------------ intra prediction
def cases_pred(h, r, out):
    stride = 48
    for mode in range(12):
        fn = h.pred4x4[mode]
        for rep in range(4):
            buf = r.u8((24, stride))
            tr_ext = r.u8(8)
            off = 8 * stride + 16
            if rep & 1:   # caller-synthesised top-right (h264_mb.c:675-689)
                tr_ext[:4] = buf[7, 16 + 3]
                tr = p8(tr_ext)
            else:
                tr = p8(buf, off + 4 - stride)
            if not fn:
                continue
            fn(p8(buf, off), tr, stride)
            out["pred4x4/%d/%d" % (mode, rep)] = buf.tobytes()
    for mode in range(12):
        fn = h.pred8x8l[mode]
        for rep in range(4):
            buf = r.u8((24, stride))
            if not fn:
                continue
            fn(p8(buf, 8 * stride + 16), (rep & 1) * 0x8000, (rep >> 1) * 0x4000, stride)
            out["pred8x8l/%d/%d" % (mode, rep)] = buf.tobytes()
    for mode in range(11):
        fn = h.pred8x8[mode]
        for rep in range(3):
            buf = r.u8((24, stride)) if rep else _edge_pixels(r, (24, stride), 0, 4)
            if not fn:
                continue
            fn(p8(buf, 8 * stride + 16), stride)
            out["pred8x8/%d/%d" % (mode, rep)] = buf.tobytes()
    for mode in range(7):
        fn = h.pred16x16[mode]
        for rep in range(3):
            buf = r.u8((32, stride)) if rep else _edge_pixels(r, (32, stride), 1, 20)
            if not fn:
                continue
            fn(p8(buf, 8 * stride + 16), stride)
            out["pred16x16/%d/%d" % (mode, rep)] = buf.tobytes()


def cases_pred_add(h, r, out):
    """lossless prediction + residual (transform bypass): h264pred_template.c:1127-1354"""
    stride = 48
    for name, tab, n in (("pred4x4_add", h.pred4x4_add, 4), ("pred8x8l_add", h.pred8x8l_add, 8)):
        for d in range(2):
            for rep in range(3):
                buf = r.u8((24, stride))
                blk = r.randint(-255, 255, n * n).astype(np.int16)
                if not tab[d]:
                    continue
                tab[d](p8(buf, 8 * stride + 16), p16(blk), stride)
                out["%s/%d/%d" % (name, d, rep)] = buf.tobytes() + blk.tobytes()
    for d in range(2):
        for rep in range(4):
            buf = r.u8((24, stride))
            blk = r.randint(-255, 255, 64).astype(np.int16)
            if not h.pred8x8l_filter_add[d]:
                continue
            h.pred8x8l_filter_add[d](p8(buf, 8 * stride + 16), p16(blk), (rep & 1) * 0x8000, (rep >> 1) * 0x4000, stride)
            out["pred8x8l_filter_add/%d/%d" % (d, rep)] = buf.tobytes() + blk.tobytes()
    for name, tab, nblk, w in (("pred8x8_add", h.pred8x8_add, 4, 8), ("pred16x16_add", h.pred16x16_add, 16, 16)):
        offs = np.array([4 * (i & 1) + 8 * ((i >> 2) & 1) + (4 * ((i >> 1) & 1) + 8 * (i >> 3)) * stride for i in range(16)], np.int32)
        if nblk == 4:
            offs = np.array([0, 4, 4 * stride, 4 * stride + 4] + [0] * 12, np.int32)
        for d in (1, 2):
            for rep in range(3):
                buf = r.u8((32, stride))
                blk = r.randint(-255, 255, 16 * nblk).astype(np.int16)
                if not tab[d]:
                    continue
                tab[d](p8(buf, 8 * stride + 16), C.cast(offs.ctypes.data, A.intp), p16(blk), stride)
                out["%s/%d/%d" % (name, d, rep)] = buf.tobytes() + blk.tobytes()


# ------------------------------------------------------------------ 4:2:2 variants (chroma_format_idc == 2)
def cases_dsp422(c, r, out):
    """h264_idct_add8 = ff_h264_idct_add8_422, chroma422_dc_dequant_idct, 16-line horizontal chroma edge filters
    (h264dsp.c:80-84, :88-91, :113-130)"""
    stride = 48
    off = _block_offsets(stride)
    fn = c.h264_idct_add8
    for rep in range(12):
        consistent = rep < 8
        nnzc = np.zeros(15 * 8, np.uint8)
        blk = np.zeros(16 * 48, np.int16)
        for j in (1, 2):
            for i in range(j * 16, j * 16 + 8):
                k = i if i < j * 16 + 4 else i + 4        # where the block is counted and placed
                mode = r.randint(0, 3)
                if mode == 0:
                    nnz = 0
                    if r.randint(0, 1):
                        blk[i * 16] = r.randint(-2047, 2047)
                elif mode == 1:
                    nnz = 1
                    blk[i * 16] = r.randint(-2047, 2047)
                else:
                    nnz = r.randint(2, 16)
                    blk[i * 16: i * 16 + 16] = _coeffs(r, 16, "small")
                if not consistent:
                    nnz = r.randint(0, 2)
                    blk[i * 16: i * 16 + 16] = _coeffs(r, 16, "small") * (r.randint(0, 3) > 0)
                nnzc[scan8(k)] = nnz
        planes = [r.u8((24, stride)) for _ in range(2)]
        if fn:
            arr = (A.u8p * 2)(p8(planes[0], 4 * stride + 8), p8(planes[1], 4 * stride + 8))
            fn(arr, pint(off), p16(blk), stride, p8(nnzc))
            out["422/h264_idct_add8/%d" % rep] = planes[0].tobytes() + planes[1].tobytes() + blk.tobytes()
    for rep in range(24):
        qmul = int([16, 64, 208, 1024, 4096, 13 * 512][rep % 6])
        blk = np.full(128, 0x0777, np.int16)
        blk[[0, 16, 32, 48, 64, 80, 96, 112]] = _coeffs(r, 8, "small" if rep < 16 else "full")
        if c.h264_chroma_dc_dequant_idct:
            c.h264_chroma_dc_dequant_idct(p16(blk), qmul)
            out["422/chroma_dc/%d" % rep] = blk.tobytes()
    triples = []
    a, b, t = 255.0, 18.0, 25.0
    for _ in range(36):
        triples.append((int(a), int(b), int(t)))
        a, b, t = a * 0.9, b * 0.92, t * 0.9
    stride = 32
    for name, has_tc in (("h264_h_loop_filter_chroma", 1), ("h264_h_loop_filter_chroma_mbaff", 1),
                         ("h264_h_loop_filter_chroma_intra", 0), ("h264_h_loop_filter_chroma_mbaff_intra", 0)):
        fn = getattr(c, name)
        for k, (al, be, t0) in enumerate(triples):
            buf = _edge_pixels(r, (32, stride), 1, 8)
            tc = np.array([r.randint(-1, max(t0, 0)) for _ in range(4)], np.int8)
            if k % 5 == 0:
                tc[:] = t0
            if not fn:
                continue
            if has_tc:
                fn(p8(buf, 8 * stride + 8), stride, al, be, pi8(tc))
            else:
                fn(p8(buf, 8 * stride + 8), stride, al, be)
            out["422/%s/%d" % (name, k)] = buf.tobytes()


def cases_pred422(h, r, out):
    """pred8x8[] = the pred8x16_* functions, pred8x8_add[] = pred8x16_*_add (h264pred.c:470-531, :558-565)"""
    stride = 48
    for mode in range(11):
        fn = h.pred8x8[mode]
        for rep in range(3):
            buf = r.u8((32, stride)) if rep else _edge_pixels(r, (32, stride), 0, 4)
            if not fn:
                continue
            fn(p8(buf, 8 * stride + 16), stride)
            out["422/pred8x8/%d/%d" % (mode, rep)] = buf.tobytes()
    offs = np.array([4 * (i & 1) + 8 * ((i >> 2) & 1) + (4 * ((i >> 1) & 1) + 8 * (i >> 3)) * stride for i in range(16)], np.int32)
    for d in (1, 2):
        for rep in range(3):
            buf = r.u8((32, stride))
            blk = r.randint(-255, 255, 16 * 8).astype(np.int16)
            if not h.pred8x8_add[d]:
                continue
            h.pred8x8_add[d](p8(buf, 8 * stride + 16), C.cast(offs.ctypes.data, A.intp), p16(blk), stride)
            out["422/pred8x8_add/%d/%d" % (d, rep)] = buf.tobytes() + blk.tobytes()


GROUPS = OrderedDict([
    ("idct", ("h264dsp", cases_idct)),
    ("idct_multi", ("h264dsp", cases_idct_multi)),
    ("dc", ("h264dsp", cases_dc)),
    ("addpx", ("h264dsp", cases_addpx)),
    ("weight", ("h264dsp", cases_weight)),
    ("loopfilter", ("h264dsp", cases_loopfilter)),
    ("startcode", ("h264dsp", cases_startcode)),
    ("qpel", ("h264qpel", cases_qpel)),
    ("chroma", ("h264chroma", cases_chroma)),
    ("videodsp", ("videodsp", cases_videodsp)),
    ("pred", ("h264pred", cases_pred)),
    ("pred_add", ("h264pred", cases_pred_add)),
    ("dsp422", ("h264dsp", cases_dsp422, 8, 2)),
    ("pred422", ("h264pred", cases_pred422, 8, 2)),
    ("qpel_extreme", ("h264qpel", cases_qpel_extreme)),
])


def run_group(provider, group, seed=0x264):
    table, fn, *args = GROUPS[group]
    ctx = getattr(provider, table)(*args)      # optional (bit_depth, chroma_format_idc)
    out = OrderedDict()
    # the stream depends only on (seed, group), never on which pointers a provider fills
    fn(ctx, SplitMix64(seed * 1000003 + list(GROUPS).index(group)), out)
    return out


def run_all(provider, seed=0x264, groups=None):
    res = OrderedDict()
    for g in (groups or GROUPS):
        res.update(run_group(provider, g, seed))
    return res
