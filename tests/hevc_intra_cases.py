"""HEVC intra prediction at wrapper level (HEVCPredContext.intra_pred[], hevcpred_template.c:31-334):
a picture, its motion field's is_intra map, the PPS's min_tb_addr_zs table and a list of transform blocks (position, size,
plane, mode, lc->na availability flags).  Blocks are grouped into launches whose members neither read nor write what
another member writes; running the launches one after the other equals running all blocks in list order, which is what the
host backends do.  Backends: the reference's own intra_pred (oracle/_ref/libhevcfilterref.so: ref_hevc_intra_pred_blocks),
the oracle (oracle_hevc_intra_pred_blocks), the product (mi355_hevc_intra_pred_blocks_dev, device pointers)."""
import ctypes as C

import numpy as np

from rng import SplitMix64

MVF_DT = np.dtype([("mv", "<i2", (2, 2)), ("ref_idx", "i1", 2), ("pred_flag", "i1", 2), ("is_intra", "u1"), ("pad", "u1", 3)])
BL, L, UL, U, UR = 1, 2, 4, 8, 16


class IntraPicture(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("linesize", C.c_int32 * 3), ("width", C.c_int32), ("height", C.c_int32),
                ("hshift", C.c_int32), ("vshift", C.c_int32), ("log2_min_pu_size", C.c_int32), ("log2_min_tb_size", C.c_int32),
                ("min_pu_width", C.c_int32), ("min_pu_height", C.c_int32), ("min_tb_width", C.c_int32),
                ("constrained_intra_pred", C.c_int32), ("strong_intra_smoothing", C.c_int32), ("reserved", C.c_int32),
                ("tab_mvf", C.c_void_p), ("min_tb_addr_zs", C.c_void_p)]


class IntraBlock(C.Structure):
    _fields_ = [("pic", C.c_int32), ("x0", C.c_uint16), ("y0", C.c_uint16), ("log2_size", C.c_uint8), ("c_idx", C.c_uint8),
                ("mode", C.c_uint8), ("cand", C.c_uint8)]


assert C.sizeof(IntraPicture) == 104 and C.sizeof(IntraBlock) == 12

CASES = {
    # name: (width, height, bit depth, log2 ctb, log2 min pu, constrained, strong, smooth content, intra share, blocks, seed)
    "i8_noise":        (128, 96, 8, 6, 2, 0, 0, 0, 1.0, 260, 1),
    "i8_smooth_strong": (192, 128, 8, 6, 2, 0, 1, 1, 1.0, 260, 2),
    "i10_strong":      (136, 72, 10, 6, 2, 0, 1, 1, 1.0, 220, 4),      # sizes not multiples of the CTB
    "i9_ctb16":        (80, 48, 9, 4, 2, 0, 0, 0, 1.0, 160, 4),
    "i8_cip":          (128, 96, 8, 6, 2, 1, 1, 1, 0.6, 320, 3),        # constrained intra: substitution paths
    "i10_cip_pu8":     (136, 104, 10, 5, 3, 1, 0, 0, 0.5, 320, 6),       # min PU 8: blocks off the PU grid
    "i8_cip_sparse":   (96, 64, 8, 5, 2, 1, 1, 1, 0.15, 260, 11),        # few intra neighbours: long substitution runs
}


def zscan_table(w, h, l2ctb, l2tb):
    """pps->min_tb_addr_zs for one tile (hevc_ps.c:1153-1167): CTBs in raster order, z-order inside a CTB"""
    tw, th = w >> l2tb, h >> l2tb
    ctb_w = (w + (1 << l2ctb) - 1) >> l2ctb
    d = l2ctb - l2tb
    y, x = np.mgrid[0:th, 0:tw]
    val = ((y >> d) * ctb_w + (x >> d)) << (2 * d)
    for i in range(d):
        m = 1 << i
        val = val + np.where(x & m, m * m, 0) + np.where(y & m, 2 * m * m, 0)
    return val.astype(np.int32).reshape(-1), tw


class Case:
    def __init__(self, name):
        w, h, bd, l2ctb, l2pu, cip, strong, smooth, share, nblocks, seed = CASES[name]
        r = SplitMix64(0x18A000 + seed)
        self.name, self.w, self.h, self.bd, self.l2ctb, self.l2pu, self.cip, self.strong = name, w, h, bd, l2ctb, l2pu, cip, strong
        dt = np.uint8 if bd == 8 else np.uint16
        mx = (1 << bd) - 1
        self.planes = []
        for c in range(3):
            pw, ph = (w, h) if c == 0 else (w // 2, h // 2)
            if smooth:
                yy, xx = np.mgrid[0:ph, 0:pw]
                a = (mx // 4) + (xx * int(r.randint(0, 5 << (bd - 8)))) // 8 + (yy * int(r.randint(0, 5 << (bd - 8)))) // 8 + r.randint(-1, 1, (ph, pw))
                steps = r.randint(-(20 << (bd - 8)), 20 << (bd - 8), ((ph + 95) // 96, (pw + 95) // 96))     # a few large plateaus
                a = a + np.kron(steps, np.ones((96, 96), np.int64))[:ph, :pw]
            else:
                a = r.randint(0, mx, (ph, pw))
            stride = (pw * dt().itemsize + 31) // 32 * 32 + 32
            buf = np.zeros((ph + 2, stride), np.uint8)          # a guard row above and below
            buf[:] = 0xA5
            buf[1:ph + 1, :pw * dt().itemsize] = np.clip(a, 0, mx).astype(dt).view(np.uint8).reshape(ph, -1)
            self.planes.append(buf)
        self.l2tb = 2
        self.min_pu_w, self.min_pu_h = w >> l2pu, h >> l2pu
        self.mvf = np.zeros(self.min_pu_w * self.min_pu_h, MVF_DT)
        self.mvf["is_intra"] = (r.uniform(self.min_pu_w * self.min_pu_h) < share).astype(np.uint8)
        self.mvf["pad"] = r.randint(0, 255, (self.min_pu_w * self.min_pu_h, 3))      # must be ignored
        self.zs, self.min_tb_w = zscan_table(w, h, l2ctb, self.l2tb)
        # ---- blocks
        ctb = 1 << l2ctb
        blocks = []
        while len(blocks) < nblocks:
            c_idx = int(r.randint(0, 2))
            l2 = int(r.randint(2, 5))
            n = 1 << l2
            nl = n << (1 if c_idx else 0)                       # extent in luma samples
            if nl > ctb or nl > 64:
                continue
            x0 = int(r.randint(0, w // nl - 1)) * nl if w >= nl else -1
            y0 = int(r.randint(0, h // nl - 1)) * nl if h >= nl else -1
            if x0 < 0 or y0 < 0 or x0 + nl > w or y0 + nl > h:
                continue
            if self.cip:
                # the block itself is intra (the decoder would not predict it otherwise)
                self.mvf["is_intra"].reshape(self.min_pu_h, self.min_pu_w)[y0 >> l2pu:((y0 + nl - 1) >> l2pu) + 1, x0 >> l2pu:((x0 + nl - 1) >> l2pu) + 1] = 1
            mode = int(r.randint(0, 34))
            x0b, y0b = x0 & (ctb - 1), y0 & (ctb - 1)
            # ff_hevc_set_neighbour_available (hevc_mvs.c:42-60) for one slice, one tile; CTB-level flags from the position
            left, up = x0 > 0, y0 > 0
            up_left = left and up
            if x0b + nl == ctb:
                up_right = (y0 >= ctb) and (x0 + nl < w) and not y0b
            else:
                up_right = up
            up_right = up_right and (x0 + nl) < w
            bottom_left = left and (y0 + nl) < h
            cand = (BL if bottom_left else 0) | (L if left else 0) | (UL if up_left else 0) | (U if up else 0) | (UR if up_right else 0)
            if r.uniform() < 0.3:
                cand &= int(r.randint(0, 31))                   # slice / tile boundaries take neighbours away
            blocks.append((x0, y0, l2, c_idx, mode, cand))
        self.blocks = blocks
        self.launches = self._group(blocks)
        self.order = [i for g in self.launches for i in g]

    def _rects(self, blk):
        x0, y0, l2, c_idx, _, _ = blk
        sh = 1 if c_idx else 0
        x, y, n = x0 >> sh, y0 >> sh, 1 << l2
        wr = (x, y, x + n, y + n)
        rd = [(x - 1, y - 1, x, y + 2 * n), (x - 1, y - 1, x + 2 * n, y)]
        return c_idx, wr, rd

    def _group(self, blocks):
        def hit(a, b):
            return a[0] < b[2] and b[0] < a[2] and a[1] < b[3] and b[1] < a[3]
        launches, members, cur, cur_r = [], [], [], []
        for i, blk in enumerate(blocks):
            c, wr, rd = self._rects(blk)
            clash = any(c == c2 and (hit(wr, w2) or any(hit(wr, q) for q in r2) or any(hit(q, w2) for q in rd)) for c2, w2, r2 in cur_r)
            if clash:
                launches.append(cur)
                cur, cur_r = [], []
            cur.append(i)
            cur_r.append((c, wr, rd))
        if cur:
            launches.append(cur)
        return launches

    def descriptor(self, ptr):
        d = IntraPicture()
        for c in range(3):
            d.data[c] = ptr(self.planes[c]) + self.planes[c].shape[1]
            d.linesize[c] = self.planes[c].shape[1]
        d.width, d.height, d.hshift, d.vshift = self.w, self.h, 1, 1
        d.log2_min_pu_size, d.log2_min_tb_size = self.l2pu, self.l2tb
        d.min_pu_width, d.min_pu_height, d.min_tb_width = self.min_pu_w, self.min_pu_h, self.min_tb_w
        d.constrained_intra_pred, d.strong_intra_smoothing = self.cip, self.strong
        d.tab_mvf, d.min_tb_addr_zs = ptr(self.mvf), ptr(self.zs)
        return d

    def block_array(self, pic, idx):
        arr = (IntraBlock * len(idx))()
        for k, i in enumerate(idx):
            x0, y0, l2, c_idx, mode, cand = self.blocks[i]
            arr[k] = IntraBlock(pic, x0, y0, l2, c_idx, mode, cand)
        return arr


def run_host(fn, name):
    """fn(pictures, blocks, n, bit_depth) with host pointers; all blocks in launch order"""
    c = Case(name)
    d = c.descriptor(lambda a: a.ctypes.data)
    arr = c.block_array(0, c.order)
    rc = fn(C.byref(d), arr, len(arr), c.bd)
    assert rc in (0, None), rc
    return [p.copy() for p in c.planes], c


class TuJob(C.Structure):                 # mi355_hevc_tu_job (include/mi355_hevc_batch.h)
    _fields_ = [("coeffs", C.c_void_p), ("dst", C.c_void_p), ("dst_stride", C.c_int32), ("log2_size", C.c_uint8),
                ("col_limit", C.c_uint8), ("kind", C.c_uint8), ("reserved", C.c_uint8)]


def run_device(lib, name, npics=1, residual=None):
    """the product: everything on the device, one call per launch group covering all pictures.
    residual: None = predictions only; "split" = every group's predictions, then the transform units of its blocks
    (mi355_hevc_residual_batch_dev); "fused" = both in one launch per group (mi355_hevc_intra_recon_blocks_dev).  Units: random
    coefficients, kinds idct / idct_dc / bypass (4x4 luma also dst / skip), about a third of the blocks without one."""
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
    descs = (IntraPicture * npics)()
    planes_dev = []
    for i in range(npics):
        dev = {}

        def ptr(a, dev=dev):
            p = up(a)
            dev[id(a)] = p
            return p
        d = c.descriptor(ptr)
        C.memmove(C.byref(descs, i * C.sizeof(IntraPicture)), C.byref(d), C.sizeof(IntraPicture))
        planes_dev.append([dev[id(pl)] for pl in c.planes])
    d_desc = up(np.frombuffer(bytes(descs), np.uint8))
    fn = lib.mi355_hevc_intra_pred_blocks_dev
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    # one block array for the whole job: per launch group, the group's blocks for every picture
    chunks, spans, pos = [], [], 0
    for g in c.launches:
        for i in range(npics):
            chunks.append(bytes(c.block_array(i, g)))
        spans.append((pos, len(g) * npics))
        pos += len(g) * npics
    d_blocks = up(np.frombuffer(b"".join(chunks), np.uint8))
    if residual:
        assert C.sizeof(TuJob) == 24
        r = SplitMix64(0x7E51D + sum(map(ord, name)))
        total = sum(cnt for _, cnt in spans)
        coef = r.randint(-400, 400, (total, 1024)).astype(np.int16)
        coef[r.uniform(total) < 0.5, 16:] = 0                        # sparse units too
        d_coef = up(coef)
        tus = (TuJob * total)()
        k = 0
        for g in c.launches:
            for i in range(npics):
                for b in g:
                    x0, y0, l2, c_idx, _, _ = c.blocks[b]
                    sh = 1 if c_idx else 0
                    px = 2 if c.bd > 8 else 1
                    stride = c.planes[c_idx].shape[1]
                    dst = planes_dev[i][c_idx] + stride + (y0 >> sh) * stride + (x0 >> sh) * px
                    kinds = [0, 1, 4] + ([2, 3] if l2 == 2 and c_idx == 0 else [])     # MI355_HEVC_TU_IDCT, _IDCT_DC, _BYPASS, _DST4, _SKIP
                    kind = kinds[int(r.randint(0, len(kinds) - 1))]
                    has = r.uniform() < 0.7
                    tus[k] = TuJob(d_coef + k * 2048 if has else None, dst, stride, l2, int(r.randint(1, 1 << l2)), kind, 0)
                    k += 1
        d_tus = up(np.frombuffer(bytes(tus), np.uint8))
        fused = lib.mi355_hevc_intra_recon_blocks_dev
        fused.restype = C.c_int
        fused.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        res = lib.mi355_hevc_residual_batch_dev
        res.restype = C.c_int
        res.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    for start, cnt in spans:
        if residual == "fused":
            assert fused(d_desc, d_blocks + start * C.sizeof(IntraBlock), d_tus + start * 24, cnt, c.bd, None) == 0
            continue
        assert fn(d_desc, d_blocks + start * C.sizeof(IntraBlock), cnt, c.bd, None) == 0
        if residual == "split":
            # the units that exist, packed (the batch entry point takes no empty jobs)
            live = [tus[j] for j in range(start, start + cnt) if tus[j].coeffs]
            if live:
                arr = (TuJob * len(live))(*live)
                d_live = up(np.frombuffer(bytes(arr), np.uint8))
                assert res(d_live, len(live), c.bd, None) == 0
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
