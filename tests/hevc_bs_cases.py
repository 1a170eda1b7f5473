"""Boundary-strength cases: a random power-of-two tiling of a picture into the blocks the reference calls
ff_hevc_deblocking_boundary_strengths for, a random motion field / cbf_luma map, one reference-list pair.
 * reference: the function itself, once per block (oracle/_ref/libhevcfilterref.so: ref_hevc_boundary_strengths);
 * oracle / product: the per-cell formulation of include/mi355_hevc_batch.h from the edge marks derived from the same tiling."""
import ctypes as C

import numpy as np

from rng import SplitMix64

MVF_DT = np.dtype([("mv", "<i2", (2, 2)), ("ref_idx", "i1", 2), ("pred_flag", "i1", 2), ("is_intra", "u1"), ("pad", "u1", 3)])
assert MVF_DT.itemsize == 16
L_BLOCK, T_BLOCK, L_INNER, T_INNER = 1, 2, 4, 8


class BsPicture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("log2_min_pu_size", C.c_int32), ("log2_min_tb_size", C.c_int32),
                ("min_pu_width", C.c_int32), ("min_tb_width", C.c_int32), ("bs_width", C.c_int32), ("reserved", C.c_int32),
                ("tab_mvf", C.c_void_p), ("cbf_luma", C.c_void_p), ("edge_flags", C.c_void_p), ("ref_poc", (C.c_int32 * 16) * 2),
                ("vertical_bs", C.c_void_p), ("horizontal_bs", C.c_void_p)]


CASES = {"b128x96": (128, 96, 32, 1), "b256x64": (256, 64, 32, 2), "b64x64_16": (64, 64, 16, 3), "b96x160": (96, 160, 32, 4), "b32x32": (32, 32, 32, 5)}


class Case:
    def __init__(self, name):
        w, h, root, seed = CASES[name]
        r = SplitMix64(0xB5000 + seed)
        self.w, self.h = w, h
        cw, chh = w >> 2, h >> 2
        # ---- motion field on the 4x4 grid: per 8x8 region one field, or two halves
        pool = np.zeros(12, MVF_DT)
        for k in range(12):
            kind = r.randint(0, 9)
            if kind == 0:
                pool[k]["is_intra"] = 1
                continue
            lists = (1, 0) if kind < 4 else ((0, 1) if kind < 6 else (1, 1))
            pool[k]["pred_flag"] = lists
            pool[k]["ref_idx"] = (r.randint(0, 2), r.randint(0, 2))
            for l in range(2):
                pool[k]["mv"][l] = (int(np.array([0, 2, 5, -3, 9])[r.randint(0, 4)]), int(np.array([0, 3, -4, 7])[r.randint(0, 3)]))
        mvf = np.zeros((chh, cw), MVF_DT)
        for y in range(0, chh, 2):
            for x in range(0, cw, 2):
                a, b = pool[r.randint(0, 11)], pool[r.randint(0, 11)]
                pat = r.randint(0, 3)
                mvf[y:y + 2, x:x + 2] = a
                if pat == 1:
                    mvf[y:y + 2, x + 1] = b
                elif pat == 2:
                    mvf[y + 1, x:x + 2] = b
        self.mvf = mvf
        self.cbf = (r.uniform((chh, cw)) < 0.3).astype(np.uint8)
        self.ref_poc = np.array([[8, 4, 8] + [0] * 13, [8, 12, 4] + [0] * 13], np.int32)
        # ---- tiling
        blocks = []

        def split(x0, y0, size):
            if size > 4 and r.uniform() < (0.75 if size > 8 else 0.35):
                hs = size // 2
                for dy in (0, hs):
                    for dx in (0, hs):
                        split(x0 + dx, y0 + dy, hs)
            else:
                blocks.append((x0, y0, size.bit_length() - 1))
        for y0 in range(0, h, root):
            for x0 in range(0, w, root):
                split(x0, y0, root)
        self.blocks = np.array(blocks, np.int32)
        fl = np.zeros((chh, cw), np.uint8)
        for x0, y0, l2 in blocks:
            size = 1 << l2
            cx, cy, n = x0 >> 2, y0 >> 2, size >> 2
            if x0 > 0 and not (x0 & 7):
                fl[cy:cy + n, cx] |= L_BLOCK
            if y0 > 0 and not (y0 & 7):
                fl[cy, cx:cx + n] |= T_BLOCK
            if size > 4 and not mvf[cy, cx]["is_intra"]:
                for j in range(8, size, 8):
                    fl[cy + (j >> 2), cx:cx + n] |= T_INNER
                    fl[cy:cy + n, cx + (j >> 2)] |= L_INNER
        self.flags = fl
        self.bs_w = w >> 3
        self.nbs = 2 * self.bs_w * ((h >> 3) + 1)

    def descriptor(self, ptr, vbs, hbs):
        d = BsPicture()
        d.width, d.height, d.log2_min_pu_size, d.log2_min_tb_size = self.w, self.h, 2, 2
        d.min_pu_width, d.min_tb_width, d.bs_width = self.w >> 2, self.w >> 2, self.bs_w
        d.tab_mvf, d.cbf_luma, d.edge_flags = ptr(self.mvf), ptr(self.cbf), ptr(self.flags)
        for l in range(2):
            for i in range(16):
                d.ref_poc[l][i] = int(self.ref_poc[l, i])
        d.vertical_bs, d.horizontal_bs = vbs, hbs
        return d


def run_reference(lib, name):
    c = Case(name)
    v, h = np.zeros(c.nbs, np.uint8), np.zeros(c.nbs, np.uint8)
    d = c.descriptor(lambda a: a.ctypes.data, v.ctypes.data, h.ctypes.data)
    lib.ref_hevc_boundary_strengths.restype = C.c_int
    assert lib.ref_hevc_boundary_strengths(C.byref(d), C.c_void_p(c.blocks.ctypes.data), len(c.blocks)) == 0
    return v, h, c


def run_oracle(lib, name):
    c = Case(name)
    v, h = np.full(c.nbs, 0, np.uint8), np.full(c.nbs, 0, np.uint8)
    d = c.descriptor(lambda a: a.ctypes.data, v.ctypes.data, h.ctypes.data)
    lib.oracle_hevc_boundary_strengths.restype = None
    lib.oracle_hevc_boundary_strengths(C.byref(d))
    return v, h, c


def run_device(lib, name, npics=2):
    c = Case(name)
    lib.mi355_malloc.restype = C.c_void_p
    lib.mi355_malloc.argtypes = [C.c_size_t]
    allocs = []

    def up(a):
        a = np.ascontiguousarray(a)
        p = lib.mi355_malloc(max(a.nbytes, 16))
        assert p and lib.mi355_memcpy_h2d(C.c_void_p(p), C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)) == 0
        allocs.append(p)
        return p
    descs = (BsPicture * npics)()
    outs = []
    for i in range(npics):
        dv, dh = up(np.full(c.nbs, 0xEE, np.uint8)), up(np.full(c.nbs, 0xEE, np.uint8))      # dirty: the kernel writes every grid cell
        d = c.descriptor(up, dv, dh)
        C.memmove(C.byref(descs, i * C.sizeof(BsPicture)), C.byref(d), C.sizeof(BsPicture))
        outs.append((dv, dh))
    d_desc = up(np.frombuffer(bytes(descs), np.uint8))
    lib.mi355_hevc_boundary_strengths_dev.restype = C.c_int
    lib.mi355_hevc_boundary_strengths_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    assert lib.mi355_hevc_boundary_strengths_dev(d_desc, npics, c.w, c.h, None) == 0
    assert lib.mi355_sync(None) == 0
    res = []
    for dv, dh in outs:
        v, h = np.zeros(c.nbs, np.uint8), np.zeros(c.nbs, np.uint8)
        lib.mi355_memcpy_d2h(C.c_void_p(v.ctypes.data), C.c_void_p(dv), C.c_size_t(c.nbs))
        lib.mi355_memcpy_d2h(C.c_void_p(h.ctypes.data), C.c_void_p(dh), C.c_size_t(c.nbs))
        res.append((v, h))
    for p in allocs:
        lib.mi355_free(C.c_void_p(p))
    return res, c


def grid_mask(c):
    """which entries of the two arrays the per-cell formulation defines (every cell side on the 8x8 grid)"""
    mv, mh = np.zeros(c.nbs, bool), np.zeros(c.nbs, bool)
    for y in range(0, c.h, 4):
        for x in range(0, c.w, 8):
            mv[(x >> 3) + (y >> 2) * c.bs_w] = True
    for y in range(0, c.h, 8):
        for x in range(0, c.w, 4):
            mh[(x + y * c.bs_w) >> 2] = True
    return mv, mh
