"""Picture-level HEVC deblocking cases (driver half): a picture plus the frame-level arrays the reference's
slice decoder leaves behind (bS arrays, qp_y_tab, is_pcm, per-CTB offsets), as seeded synthetic data, run through a
backend's deblocking driver.  Backends: the reference's own hevc_filter.c (oracle/_ref/libhevcfilterref.so), the oracle
(oracle_hevc_deblock_picture), the product (mi355_hevc_deblock_pictures_dev, device pointers)."""
import ctypes as C

import numpy as np

from rng import SplitMix64


class DBParams(C.Structure):
    _fields_ = [("beta_offset", C.c_int32), ("tc_offset", C.c_int32)]


class LfPicture(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("linesize", C.c_int32 * 3), ("width", C.c_int32), ("height", C.c_int32),
                ("log2_ctb_size", C.c_int32), ("log2_min_cb_size", C.c_int32), ("log2_min_pu_size", C.c_int32),
                ("min_cb_width", C.c_int32), ("min_pu_width", C.c_int32), ("min_pu_height", C.c_int32),
                ("ctb_width", C.c_int32), ("bs_width", C.c_int32),
                ("vertical_bs", C.c_void_p), ("horizontal_bs", C.c_void_p), ("qp_y_tab", C.c_void_p), ("is_pcm", C.c_void_p),
                ("deblock", C.c_void_p), ("pcmf", C.c_int32), ("cb_qp_offset", C.c_int32), ("cr_qp_offset", C.c_int32)]


CASES = {
    # name: (width, height, bit depth, log2 ctb, pcmf, seed)
    "p8_64":      (192, 128, 8, 6, 0, 1),
    "p10_64":     (256, 136, 10, 6, 0, 2),        # height not a multiple of the CTB size
    "p9_32_pcm":  (200, 104, 9, 5, 1, 3),         # width not a multiple of 16 / of the CTB size; pcm / bypass masks
    "p10_16_pcm": (176, 144, 10, 4, 1, 4),
    "p8_tiny":    (16, 16, 8, 4, 0, 5),
    "p8_wide":    (416, 24, 8, 6, 1, 6),
}


class Case:
    def __init__(self, name):
        w, h, bd, l2ctb, pcmf, seed = CASES[name]
        r = SplitMix64(0x265000 + seed)
        self.name, self.w, self.h, self.bd, self.l2ctb, self.pcmf = name, w, h, bd, l2ctb, pcmf
        dt = np.uint8 if bd == 8 else np.uint16
        mx = (1 << bd) - 1
        self.planes = []
        for c in range(3):
            pw, ph = (w, h) if c == 0 else (w // 2, h // 2)
            # piecewise-smooth content with steps at block boundaries: the filters' decisions go every way
            yy, xx = np.mgrid[0:ph, 0:pw]
            base = (mx // 3) + (xx * r.randint(-2, 2)) // 4 + (yy * r.randint(-2, 2)) // 4
            blk = (8 if c == 0 else 4)
            steps = r.randint(-(6 << (bd - 8)), 6 << (bd - 8), ((ph + blk - 1) // blk, (pw + blk - 1) // blk))
            a = base + np.kron(steps, np.ones((blk, blk), np.int64))[:ph, :pw] + r.randint(-2, 2, (ph, pw))
            hard = r.uniform((ph, pw)) < 0.02
            a = np.where(hard, r.randint(0, mx, (ph, pw)), a)
            stride = (pw * dt().itemsize + 31) // 32 * 32 + 32
            buf = np.zeros((ph + 2, stride), np.uint8)      # a guard row above and below
            buf[1:ph + 1, :pw * dt().itemsize] = np.clip(a, 0, mx).astype(dt).view(np.uint8).reshape(ph, -1)
            self.planes.append(buf)
        self.l2cb, self.l2pu = 3, 2
        self.min_cb_w, self.min_cb_h = w >> 3, (h + 7) >> 3
        self.min_pu_w, self.min_pu_h = w >> 2, h >> 2
        self.ctb_w, self.ctb_h = (w + (1 << l2ctb) - 1) >> l2ctb, (h + (1 << l2ctb) - 1) >> l2ctb
        self.bs_w, bs_h = w >> 3, h >> 3
        n = 2 * self.bs_w * (bs_h + 1)
        pick = np.array([0, 0, 1, 1, 2, 2, 2], np.uint8)
        self.vbs = pick[r.randint(0, 6, n)]
        self.hbs = pick[r.randint(0, 6, n)]
        # no strength outside the picture (the decoder never sets one there; the driver would read side information and
        # filter samples beyond the last row)
        self.vbs[(h >> 2) * self.bs_w:] = 0
        self.hbs[(h * self.bs_w) >> 2:] = 0
        self.qp = r.randint(18, 46, self.min_cb_w * self.min_cb_h).astype(np.int8)
        self.is_pcm = (r.uniform(self.min_pu_w * self.min_pu_h) < 0.15).astype(np.uint8)
        self.db = np.zeros((self.ctb_w * self.ctb_h, 2), np.int32)
        self.db[:, 0] = 2 * r.randint(-6, 6, self.ctb_w * self.ctb_h)
        self.db[:, 1] = 2 * r.randint(-6, 6, self.ctb_w * self.ctb_h)
        self.cb_off, self.cr_off = int(r.randint(-6, 6)), int(r.randint(-6, 6))

    def descriptor(self, ptr):
        """ptr(array) -> address the backend can use (host address, or a device copy)"""
        d = LfPicture()
        for c in range(3):
            d.data[c] = ptr(self.planes[c]) + self.planes[c].shape[1]      # skip the guard row
            d.linesize[c] = self.planes[c].shape[1]
        d.width, d.height, d.log2_ctb_size = self.w, self.h, self.l2ctb
        d.log2_min_cb_size, d.log2_min_pu_size = self.l2cb, self.l2pu
        d.min_cb_width, d.min_pu_width, d.min_pu_height = self.min_cb_w, self.min_pu_w, self.min_pu_h
        d.ctb_width, d.bs_width = self.ctb_w, self.bs_w
        d.vertical_bs, d.horizontal_bs, d.qp_y_tab, d.is_pcm, d.deblock = ptr(self.vbs), ptr(self.hbs), ptr(self.qp), ptr(self.is_pcm), ptr(self.db)
        d.pcmf, d.cb_qp_offset, d.cr_qp_offset = self.pcmf, self.cb_off, self.cr_off
        return d


def run_host(fn, name):
    """fn(byref(descriptor), bit_depth) with host pointers; returns the three planes (guards included)"""
    c = Case(name)
    d = c.descriptor(lambda a: a.ctypes.data)
    rc = fn(C.byref(d), c.bd)
    assert rc in (0, None), rc
    return [p.copy() for p in c.planes], c


def run_device(lib, name, npics=1):
    """the product: descriptors and every array on the device"""
    c = Case(name)
    lib.mi355_malloc.restype = C.c_void_p
    lib.mi355_malloc.argtypes = [C.c_size_t]
    allocs = []

    def up(a):
        a = np.ascontiguousarray(a)
        p = lib.mi355_malloc(max(a.nbytes, 16))
        assert p
        assert lib.mi355_memcpy_h2d(C.c_void_p(p), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)) == 0
        allocs.append(p)
        return p
    descs = (LfPicture * npics)()
    planes_dev = []
    for i in range(npics):
        dev_planes = {}

        def ptr(a, dev_planes=dev_planes):
            p = up(a)
            dev_planes[id(a)] = p
            return p
        d = c.descriptor(ptr)
        C.memmove(C.byref(descs, i * C.sizeof(LfPicture)), C.byref(d), C.sizeof(LfPicture))
        planes_dev.append([dev_planes[id(pl)] for pl in c.planes])
    d_desc = up(np.frombuffer(bytes(descs), np.uint8))
    lib.mi355_hevc_deblock_pictures_dev.restype = C.c_int
    lib.mi355_hevc_deblock_pictures_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    assert lib.mi355_hevc_deblock_pictures_dev(d_desc, npics, c.w, c.h, c.bd, None) == 0
    assert lib.mi355_sync(None) == 0
    outs = []
    for i in range(npics):
        got = []
        for k, pl in enumerate(c.planes):
            o = np.zeros_like(pl)
            assert lib.mi355_memcpy_d2h(C.c_void_p(o.ctypes.data), C.c_void_p(planes_dev[i][k]), C.c_size_t(o.nbytes)) == 0
            got.append(o)
        outs.append(got)
    for p in allocs:
        lib.mi355_free(C.c_void_p(p))
    return outs, c
