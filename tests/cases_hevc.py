"""checkasm-style differential cases for HEVCDSPContext / HEVCPredContext,
bit depths 8 and 10.  Parameter ranges follow tests/checkasm/hevc_idct.c, hevc_add_res.c,
hevc_mc.c of the reference; deblocking, SAO and intra prediction have no checkasm in the
reference and get the same treatment here.  Inputs are drawn even when a provider leaves a slot
empty so that every provider sees the same stream."""
import ctypes as C
from collections import OrderedDict

import numpy as np

import abi_ctypes as A
from rng import SplitMix64

QW = [4, 8, 12, 16, 24, 32, 48, 64]
EW = [2, 4, 6, 8, 12, 16, 24, 32]


def p8(a, off=0):
    return C.cast(a.ctypes.data + off, A.u8p)


def p16(a, off=0):
    return C.cast(a.ctypes.data + 2 * off, A.i16p)


def pint(a):
    return C.cast(a.ctypes.data, A.intp)


def pixels(r, shape, bd, smooth=False):
    if smooth:
        base = r.randint(40 << (bd - 8), 200 << (bd - 8))
        a = base + r.randint(-3 << (bd - 8), 3 << (bd - 8), shape)
        a = np.clip(a, 0, (1 << bd) - 1)
    else:
        a = r.randint(0, (1 << bd) - 1, shape)
    return a.astype(np.uint16 if bd > 8 else np.uint8)


def cases_residual(c, r, out, bd):
    px = 2 if bd > 8 else 1
    for i, size in enumerate((4, 8, 16, 32)):
        for rep in range(3):
            dst = pixels(r, (size + 8, size + 16), bd)
            res = (r.randint(-32768, 32767, size * size) >> 3).astype(np.int16)
            if c.add_residual[i]:
                c.add_residual[i](p8(dst, (4 * (size + 16) + 8) * px), p16(res), (size + 16) * px)
                out["add_residual%d/%d" % (size, rep)] = dst.tobytes()
    for rep in range(4):
        blk = r.randint(-32768, 32767, 16).astype(np.int16)
        if c.dequant:
            b = blk.copy()
            c.dequant(p16(b))
            out["dequant/%d" % rep] = b.tobytes()
        if c.transform_4x4_luma:
            b = blk.copy()
            c.transform_4x4_luma(p16(b))
            out["dst4/%d" % rep] = b.tobytes()


def cases_idct(c, r, out, bd):
    for i, size in enumerate((4, 8, 16, 32)):
        for rep in range(8):
            coef = r.randint(-32768, 32767, size * size).astype(np.int16)
            if rep < 2:
                lim = size                      # checkasm: full range, col_limit = block size
            else:
                lim = [1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 20, 24, 28, 31][r.randint(0, 15)]
                lim = min(lim, size)
                coef = r.laplace_int(200, size * size, 32767).astype(np.int16)
                if rep < 6:                     # consistent: nothing outside the top-left lim x lim
                    m = coef.reshape(size, size)
                    m[lim:, :] = 0
                    m[:, lim:] = 0
            if c.idct[i]:
                b = coef.copy()
                c.idct[i](p16(b), lim)
                out["idct%d/%d" % (size, rep)] = b.tobytes()
        for rep in range(3):
            coef = np.full(size * size, 0x1111, np.int16)
            coef[0] = r.randint(-32768, 32767)
            if c.idct_dc[i]:
                c.idct_dc[i](p16(coef))
                out["idct_dc%d/%d" % (size, rep)] = coef.tobytes()


def cases_mc(c, r, out, bd):
    px = 2 if bd > 8 else 1
    sstride = 96
    for tab, widths, maxf, name in ((c.put_hevc_qpel, QW, 3, "qpel"), (c.put_hevc_epel, EW, 7, "epel")):
        for wi, w in enumerate(widths):
            for (vy, vx) in ((0, 0), (0, 1), (1, 0), (1, 1)):
                for rep in range(2):
                    h = [2, 4, 6, 8, 12, 16, 24, 32, 48, 64][r.randint(0, 9)]
                    if name == "qpel":
                        h = max(4, h)
                    else:
                        h = min(32, h)
                    mx = r.randint(1, maxf) if vx else 0
                    my = r.randint(1, maxf) if vy else 0
                    src = pixels(r, (h + 16, sstride), bd)
                    dst = np.full((h + 2, 64), 0x2222, np.int16)
                    mcbuf = np.zeros((64 + 24) * 64, np.int16)
                    fn = tab[vy][vx][wi]
                    if fn:
                        fn(p16(dst), 128, p8(src, (8 * sstride + 8) * px), sstride * px, h, mx, my, p16(mcbuf))
                        out["%s%d/%d%d/%d" % (name, w, vy, vx, rep)] = dst.tobytes()


def cases_pred(c, r, out, bd):
    px = 2 if bd > 8 else 1
    for tabs, widths, name in (((c.put_unweighted_pred, c.put_unweighted_pred_avg, c.weighted_pred, c.weighted_pred_avg), QW, "luma"),
                               ((c.put_unweighted_pred_chroma, c.put_unweighted_pred_avg_chroma, c.weighted_pred_chroma,
                                 c.weighted_pred_avg_chroma), EW, "chroma")):
        for wi, w in enumerate(widths):
            for rep in range(2):
                h = [2, 4, 8, 16, 32, 64][r.randint(0, 5)]
                s1 = r.randint(-8192, 24575, (h, 64)).astype(np.int16)
                s2 = r.randint(-8192, 24575, (h, 64)).astype(np.int16)
                denom = r.randint(0, 7)
                w0, w1 = r.randint(-128, 127), r.randint(-128, 127)
                o0, o1 = r.randint(-128, 127), r.randint(-128, 127)
                for k, fn in enumerate(t[wi] for t in tabs):
                    dst = pixels(r, (h + 2, 80), bd)
                    if not fn:
                        continue
                    d = p8(dst, (80 + 8) * px)
                    if k == 0:
                        fn(d, 80 * px, p16(s1), 128, h)
                    elif k == 1:
                        fn(d, 80 * px, p16(s1), p16(s2), 128, h)
                    elif k == 2:
                        fn(denom, w0, o0, d, 80 * px, p16(s1), 128, h)
                    else:
                        fn(denom, w0, w1, o0, o1, d, 80 * px, p16(s1), p16(s2), 128, h)
                    out["pred_%s%d/%d/%d" % (name, w, k, rep)] = dst.tobytes()


def cases_deblock(c, r, out, bd):
    px = 2 if bd > 8 else 1
    stride = 32
    for name, horiz_edge, luma in (("hevc_h_loop_filter_luma", 1, 1), ("hevc_v_loop_filter_luma", 0, 1),
                                   ("hevc_h_loop_filter_chroma", 1, 0), ("hevc_v_loop_filter_chroma", 0, 0),
                                   ("hevc_h_loop_filter_luma_c", 1, 1), ("hevc_v_loop_filter_chroma_c", 0, 0)):
        fn = getattr(c, name)
        for rep in range(24):
            buf = pixels(r, (24, stride), bd, smooth=rep % 4 != 3)
            step = r.randint(-10 << (bd - 8), 10 << (bd - 8))
            if horiz_edge:
                buf[8:, :] = np.clip(buf[8:, :].astype(np.int64) + step, 0, (1 << bd) - 1)
            else:
                buf[:, 8:] = np.clip(buf[:, 8:].astype(np.int64) + step, 0, (1 << bd) - 1)
            beta = r.randint(0, 64)
            tc = np.array([r.randint(0, 24), r.randint(0, 24)], np.int32)
            no_p = np.array([r.randint(0, 3) == 0, r.randint(0, 3) == 0], np.uint8)
            no_q = np.array([r.randint(0, 3) == 0, r.randint(0, 3) == 0], np.uint8)
            if not fn:
                continue
            pix = p8(buf, (8 * stride + 8) * px)
            if luma:
                fn(pix, stride * px, beta, pint(tc), p8(no_p), p8(no_q))
            else:
                fn(pix, stride * px, pint(tc), p8(no_p), p8(no_q))
            out["%s/%d" % (name, rep)] = buf.tobytes()


def cases_sao(c, r, out, bd):
    px = 2 if bd > 8 else 1
    stride = 96
    for cls in range(4):
        for kind in ("band", "edge"):
            fn = (c.sao_band_filter if kind == "band" else c.sao_edge_filter)[cls]
            for rep in range(10):
                c_idx = r.randint(0, 2)
                w = [16, 32, 64][r.randint(0, 2)] >> (1 if c_idx else 0)
                h = [16, 32, 64][r.randint(0, 2)] >> (1 if c_idx else 0)
                src = pixels(r, (h + 24, stride), bd, smooth=True)
                src[::3, ::5] = pixels(r, src[::3, ::5].shape, bd)
                dst = src.copy()
                dst[:] = (0x155 if bd > 8 else 0x55)
                sao = A.SAOParams()
                for k in range(5):
                    sao.offset_val[c_idx][k] = 0 if k == 0 else r.randint(-7 << (bd - 8), 7 << (bd - 8))
                sao.band_position[c_idx] = r.randint(0, 31)
                sao.eo_class[c_idx] = r.randint(0, 3)
                borders = np.array([r.randint(0, 1) for _ in range(4)], np.int32)
                ve, he, de = r.randint(0, 1), r.randint(0, 1), r.randint(0, 1)
                if not fn:
                    continue
                off = (12 * stride + 16) * px
                if kind == "band":
                    fn(p8(dst, off), p8(src, off), stride * px, C.byref(sao), pint(borders), w, h, c_idx)
                else:
                    fn(p8(dst, off), p8(src, off), stride * px, C.byref(sao), pint(borders), w, h, c_idx, ve, he, de)
                out["sao_%s%d/%d" % (kind, cls, rep)] = dst.tobytes()


def cases_intra(hp, r, out, bd):
    px = 2 if bd > 8 else 1
    for i, size in enumerate((4, 8, 16, 32)):
        stride = 48
        for rep in range(2 + 33):
            c_idx = r.randint(0, 1)
            edge_t = pixels(r, 2 * size + 4, bd, smooth=rep & 1)
            edge_l = pixels(r, 2 * size + 4, bd, smooth=rep & 1)
            top, left = p8(edge_t, px), p8(edge_l, px)       # element -1 is addressable
            buf = pixels(r, (size + 4, stride), bd)
            dst = p8(buf, (2 * stride + 8) * px)
            if rep == 0:
                if hp.pred_planar[i]:
                    hp.pred_planar[i](dst, top, left, stride)   # NB: these take the stride in SAMPLES (hevcpred_template.c:31,349)
                    out["planar%d" % size] = buf.tobytes()
            elif rep == 1:
                if hp.pred_dc:
                    hp.pred_dc(dst, top, left, stride, i + 2, c_idx)
                    out["dc%d" % size] = buf.tobytes()
            else:
                mode = rep                                   # 2..34
                if hp.pred_angular[i]:
                    hp.pred_angular[i](dst, top, left, stride, c_idx, mode)
                    out["angular%d/%d" % (size, mode)] = buf.tobytes()


def cases_pcm(c, r, out, bd):
    """put_pcm (hevcdsp_template.c:28-41): coding-block sizes 8..32 luma / 4..16 chroma, every legal PCM depth,
    starting at an arbitrary bit position; one case runs into the end of the buffer (position clamp)"""
    px = 2 if bd > 8 else 1
    for size in (4, 8, 16, 32):
        for rep in range(3):
            pcm_bd = [1, bd, r.randint(1, bd)][rep]
            nbits = size * size * pcm_bd
            start = r.randint(0, 64)
            data = r.u8((start + nbits + 7) // 8 + 16)           # 16: the reader's 32-bit window may look past the last level
            short = rep == 2 and size == 8
            gb = A.GetBitContext()
            gb.buffer = data.ctypes.data
            gb.buffer_end = data.ctypes.data + len(data) - 16
            gb.index = start
            gb.size_in_bits = (start + nbits - (5 * pcm_bd if short else 0))
            gb.size_in_bits_plus8 = gb.size_in_bits + 8
            dst = pixels(r, (size + 8, size + 16), bd)
            if c.put_pcm:
                c.put_pcm(p8(dst, (4 * (size + 16) + 8) * px), (size + 16) * px, size, C.byref(gb), pcm_bd)
                out["put_pcm%d/%d" % (size, rep)] = dst.tobytes() + np.int32(gb.index).tobytes()


GROUPS = OrderedDict([
    ("residual", ("hevcdsp", cases_residual)), ("idct", ("hevcdsp", cases_idct)), ("mc", ("hevcdsp", cases_mc)),
    ("pred", ("hevcdsp", cases_pred)), ("deblock", ("hevcdsp", cases_deblock)), ("sao", ("hevcdsp", cases_sao)),
    ("intra", ("hevcpred", cases_intra)), ("pcm", ("hevcdsp", cases_pcm)),
])
DEPTHS = (8, 10)


def run_group(provider, group, bd, seed=0x265):
    table, fn = GROUPS[group]
    ctx = getattr(provider, table)(bd)
    out = OrderedDict()
    fn(ctx, SplitMix64(seed * 1000003 + list(GROUPS).index(group) * 16 + bd), out, bd)
    return OrderedDict(("bd%d/%s" % (bd, k), v) for k, v in out.items())


def run_all(provider, seed=0x265, depths=DEPTHS):
    res = OrderedDict()
    for bd in depths:
        for g in GROUPS:
            res.update(run_group(provider, g, bd, seed))
    return res
