"""Differential cases for the individual swscale inner loops; the reference
has no checkasm for libswscale, so shapes and ranges come from how swscale() calls them:
15-bit intermediates (0..32767 after hScale, may be negative only through filter undershoot),
horizontal filters normalised to 1<<14, vertical ones to 1<<12."""
import ctypes as C
from collections import OrderedDict

import numpy as np

from rng import SplitMix64
import sws_support as S


def _ptr_array(rows):
    arr = (C.c_void_p * len(rows))(*[r.ctypes.data for r in rows])
    return arr


def _norm_filter(r, n, size, one):
    """taps summing exactly to `one` with small negative lobes at both ends (sum of |taps| <= 1.25 x one),
    like initFilter's bicubic output: keeps every filtered value inside the range the reference's
    "clip only if bit 8 is set" shortcut (output.c:963) handles, i.e. inside its LUT"""
    lobe = one // 16 if size >= 3 else 0
    w = r.randint(1, 1000, (n, size)).astype(np.float64)
    f = np.floor(w / w.sum(axis=1, keepdims=True) * (one + 2 * lobe)).astype(np.int64)
    f[:, 0] -= lobe
    f[:, -1] -= lobe
    f[:, size // 2] += one - f.sum(axis=1)
    return f.astype(np.int16)


class Funcs:
    """Adapter: one object per implementation exposing the seven inner loops with identical
    argument lists.  `luts` is a sws_support.Luts."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix

    def _f(self, base, restype=None):
        f = getattr(self.lib, self.prefix + base)
        f.restype = restype
        return f

    def hscale(self, dst, dstW, src, filt, pos, size):
        self._f("hscale8to15")(C.c_void_p(dst.ctypes.data), dstW, C.c_void_p(src.ctypes.data), C.c_void_p(filt.ctypes.data),
                               C.c_void_p(pos.ctypes.data), size)

    def planeX(self, filt, rows, dest, dstW, dither, offset):
        self._f("yuv2planeX_8")(C.c_void_p(filt.ctypes.data), len(rows), _ptr_array(rows), C.c_void_p(dest.ctypes.data), dstW,
                                C.c_void_p(dither.ctypes.data), offset)

    def plane1(self, row, dest, dstW, dither, offset):
        self._f("yuv2plane1_8")(C.c_void_p(row.ctypes.data), C.c_void_p(dest.ctypes.data), dstW, C.c_void_p(dither.ctypes.data), offset)

    def rgbX(self, luts, lf, lrows, cf, urows, vrows, dest, dstW):
        self._f("yuv2rgb24_X")(C.byref(luts), C.c_void_p(lf.ctypes.data), _ptr_array(lrows), len(lrows), C.c_void_p(cf.ctypes.data),
                               _ptr_array(urows), _ptr_array(vrows), len(urows), C.c_void_p(dest.ctypes.data), dstW)

    def rgb2(self, luts, lrows, urows, vrows, dest, dstW, ya, uva):
        self._f("yuv2rgb24_2")(C.byref(luts), _ptr_array(lrows), _ptr_array(urows), _ptr_array(vrows), C.c_void_p(dest.ctypes.data), dstW, ya, uva)

    def rgb1(self, luts, lrow, urows, vrows, dest, dstW, uva):
        self._f("yuv2rgb24_1")(C.byref(luts), C.c_void_p(lrow.ctypes.data), _ptr_array(urows), _ptr_array(vrows),
                               C.c_void_p(dest.ctypes.data), dstW, uva)

    def c24(self, luts, dstW, planes, y0, h, dst):
        src = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
        strides = (C.c_int * 3)(*[p.strides[0] for p in planes])
        return self._f("yuv2rgb_c_24_rgb", C.c_int)(C.byref(luts), dstW, src, strides, y0, h, C.c_void_p(dst.ctypes.data), dst.strides[0])


class RefFuncs(Funcs):
    """The reference's static functions through the pointers two live contexts hold (oracle/ref_sws_glue.c):
    an rgb24 context for hScale / packed output, a planar one for yuv2plane*.  Its LUTs are the context's own."""

    def __init__(self, ref):
        self.ref, self.lib = ref, ref.lib
        self.c_rgb = ref.open("down2_128x96", dst_fmt=1)
        self.c_yuv = ref.open("down2_128x96", dst_fmt=0)
        for n in ("hscale", "planeX", "plane1", "packedX", "packed2", "packed1"):
            getattr(self.lib, "ref_sws_" + n).restype = None
        d = S.Desc()
        assert self.lib.ref_sws_describe(C.c_void_p(self.c_rgb), C.byref(d)) == 0
        self.luts = S.Context.from_desc(d).desc.luts

    def hscale(self, dst, dstW, src, filt, pos, size):
        self.lib.ref_sws_hscale(C.c_void_p(self.c_rgb), C.c_void_p(dst.ctypes.data), dstW, C.c_void_p(src.ctypes.data),
                                C.c_void_p(filt.ctypes.data), C.c_void_p(pos.ctypes.data), size)

    def planeX(self, filt, rows, dest, dstW, dither, offset):
        self.lib.ref_sws_planeX(C.c_void_p(self.c_yuv), C.c_void_p(filt.ctypes.data), len(rows), _ptr_array(rows),
                                C.c_void_p(dest.ctypes.data), dstW, C.c_void_p(dither.ctypes.data), offset)

    def plane1(self, row, dest, dstW, dither, offset):
        self.lib.ref_sws_plane1(C.c_void_p(self.c_yuv), C.c_void_p(row.ctypes.data), C.c_void_p(dest.ctypes.data), dstW,
                                C.c_void_p(dither.ctypes.data), offset)

    def rgbX(self, luts, lf, lrows, cf, urows, vrows, dest, dstW):
        self.lib.ref_sws_packedX(C.c_void_p(self.c_rgb), C.c_void_p(lf.ctypes.data), _ptr_array(lrows), len(lrows),
                                 C.c_void_p(cf.ctypes.data), _ptr_array(urows), _ptr_array(vrows), len(urows),
                                 C.c_void_p(dest.ctypes.data), dstW)

    def rgb2(self, luts, lrows, urows, vrows, dest, dstW, ya, uva):
        self.lib.ref_sws_packed2(C.c_void_p(self.c_rgb), _ptr_array(lrows), _ptr_array(urows), _ptr_array(vrows),
                                 C.c_void_p(dest.ctypes.data), dstW, ya, uva)

    def rgb1(self, luts, lrow, urows, vrows, dest, dstW, uva):
        self.lib.ref_sws_packed1(C.c_void_p(self.c_rgb), C.c_void_p(lrow.ctypes.data), _ptr_array(urows), _ptr_array(vrows),
                                 C.c_void_p(dest.ctypes.data), dstW, uva)

    c24 = None   # static in yuv2rgb.c and only reachable through sws_scale: covered by the whole-picture cases


def run_functions(fn, luts, seed=S.SEED):
    r = SplitMix64(seed)
    out = OrderedDict()
    # a19
    for rep, (dstW, size) in enumerate(((64, 8), (100, 7), (352, 1), (96, 4), (130, 2), (48, 11))):
        srcW = dstW * 2 + 16
        src = r.u8(srcW + 16)
        if rep == 3:
            src[:] = 255                                  # saturates: the FFMIN(…, 32767) branch
        filt = _norm_filter(r, dstW, size, 1 << 14)
        if rep == 3:
            filt[:, size // 2] += 300
        pos = np.sort(r.randint(0, srcW - size, dstW)).astype(np.int32)
        dst = np.full(dstW + 8, 0x1111, np.int16)
        if fn.hscale:
            fn.hscale(dst, dstW, src, filt, pos, size)
            out["hscale/%d" % rep] = dst.tobytes()
    # a20
    dither = (r.randint(0, 127, 8)).astype(np.uint8)
    for rep, (dstW, size) in enumerate(((64, 8), (100, 4), (352, 2), (50, 7), (64, 1))):
        rows = [r.randint(-600, 33000, dstW + 8).clip(-32768, 32767).astype(np.int16) for _ in range(size)]
        filt = _norm_filter(r, 1, size, 1 << 12)[0]
        dest = np.full(dstW + 8, 0x5A, np.uint8)
        off = [0, 3, 0, 3, 5][rep]
        fn.planeX(filt, rows, dest, dstW, dither, off)
        out["planeX/%d" % rep] = dest.tobytes()
        dest = np.full(dstW + 8, 0x5A, np.uint8)
        fn.plane1(rows[0], dest, dstW, dither, off)
        out["plane1/%d" % rep] = dest.tobytes()
    for rep, (dstW, ls, cs) in enumerate(((64, 8, 1), (100, 1, 4), (352, 4, 4), (50, 7, 4), (64, 2, 1), (96, 8, 8))):
        hw = (dstW + 1) >> 1
        wide = rep == 5        # out-of-range sums: exercises the "clip only if bit 8 is set" rule
        lo, hi = (-2000, 34000) if wide else (0, 32767)
        lrows = [r.randint(lo, hi, dstW + 8).clip(-32768, 32767).astype(np.int16) for _ in range(ls)]
        urows = [r.randint(lo, hi, hw + 8).clip(-32768, 32767).astype(np.int16) for _ in range(cs)]
        vrows = [r.randint(lo, hi, hw + 8).clip(-32768, 32767).astype(np.int16) for _ in range(cs)]
        lf = _norm_filter(r, 1, ls, 1 << 12)[0]
        cf = _norm_filter(r, 1, cs, 1 << 12)[0]
        dest = np.full(dstW * 3 + 16, 0x5A, np.uint8)
        fn.rgbX(luts, lf, lrows, cf, urows, vrows, dest, dstW)
        out["rgbX/%d" % rep] = dest.tobytes()
    for rep, dstW in enumerate((64, 100, 352, 50)):
        hw = (dstW + 1) >> 1
        lrows = [r.randint(0, 32767, dstW + 8).astype(np.int16) for _ in range(2)]
        urows = [r.randint(0, 32767, hw + 8).astype(np.int16) for _ in range(2)]
        vrows = [r.randint(0, 32767, hw + 8).astype(np.int16) for _ in range(2)]
        ya, uva = r.randint(0, 4095), r.randint(0, 4095)
        dest = np.full(dstW * 3 + 16, 0x5A, np.uint8)
        fn.rgb2(luts, lrows, urows, vrows, dest, dstW, ya, uva)
        out["rgb2/%d" % rep] = dest.tobytes()
        for k, uv in enumerate((0, 1000, 2048, 4000)):
            dest = np.full(dstW * 3 + 16, 0x5A, np.uint8)
            fn.rgb1(luts, lrows[0], urows, vrows, dest, dstW, uv)
            out["rgb1/%d/%d" % (rep, k)] = dest.tobytes()
    # a22 through the slice interface
    if fn.c24:
        for rep, (w, h, y0, sh) in enumerate(((64, 16, 0, 16), (70, 12, 4, 8), (22, 8, 2, 4), (352, 8, 0, 8))):
            planes = [r.u8((h, w + 4)), r.u8((h // 2, w // 2 + 4)), r.u8((h // 2, w // 2 + 4))]
            dst = np.full((h, w * 3 + 10), 0x5A, np.uint8)
            # src pointers address the first line of the slice (SwsFunc contract)
            sl = [planes[0][y0:], planes[1][y0 // 2:], planes[2][y0 // 2:]]
            n = fn.c24(luts, w, sl, y0, sh, dst)
            assert n == sh
            out["c24/%d" % rep] = dst.tobytes()
    return out
