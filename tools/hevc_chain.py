"""BASELINE.json config 3 as ONE back-to-back chain on a batch of pictures (bench.py's `extra`): 10-bit 3840x2160, CTB 64,
per CTB four 32x32 luma + two 32x32 chroma transform units (75 % with non-zeros in the top-left 8x8 only), 32x32
uni-predicted prediction units, deblocking of the whole picture FROM ITS FRAME-LEVEL ARRAYS (bS 1 on every 32x32 TU / PU
edge, bS 2 on 10 %, QP ~ U{22..37}: beta / tc derived on the device by mi355_hevc_deblock_pictures_dev), SAO on every
CTB (50 % edge, 25 % band, 25 % off).  Job arrays are numpy records laid out like the C structs; nothing loops per job.
Stages are enqueued one after the other on the null stream, each reading what the previous one wrote:
  fused MC + put_unweighted_pred -> recon;  idct32 + add_residual -> recon;  deblock (V then H) in place;  SAO recon -> out."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hevc_filter_cases as HFC  # noqa: E402

W, H, BD, PX = 3840, 2160, 10, 2
TU_DT = np.dtype([("coeffs", "<u8"), ("dst", "<u8"), ("dst_stride", "<i4"), ("log2_size", "u1"), ("col_limit", "u1"), ("kind", "u1"), ("rsv", "u1")])
MP_DT = np.dtype([("src0", "<u8"), ("src1", "<u8"), ("dst", "<u8"), ("s0", "<i4"), ("s1", "<i4"), ("ds", "<i4"), ("width", "u1"), ("height", "u1"),
                  ("chroma", "u1"), ("kind", "u1"), ("mx0", "u1"), ("my0", "u1"), ("mx1", "u1"), ("my1", "u1"), ("denom", "u1"), ("rsv", "u1", 3),
                  ("w0", "<i2"), ("w1", "<i2"), ("o0", "<i2"), ("o1", "<i2")])
SAO_DT = np.dtype([("dst", "<u8"), ("src", "<u8"), ("stride", "<i4"), ("width", "<i4"), ("height", "<i4"), ("borders", "<i4", 4),
                   ("offset_val", "<i4", 5), ("cls", "u1"), ("edge", "u1"), ("c_idx", "u1"), ("eo_class", "u1"), ("band_position", "u1"),
                   ("vert_edge", "u1"), ("horiz_edge", "u1"), ("diag_edge", "u1")])
assert TU_DT.itemsize == 24 and MP_DT.itemsize == 56 and SAO_DT.itemsize == 72
BYTES_PER_CTB = 73984


class Chain:
    def __init__(self, lib, pictures, distinct=2, seed=0x265):
        self.lib, self.P = lib, pictures
        lib.mi355_malloc.restype = C.c_void_p
        lib.mi355_malloc.argtypes = [C.c_size_t]
        lib.mi355_free.argtypes = [C.c_void_p]
        for f in ("mi355_memcpy_h2d", "mi355_memcpy_d2d"):
            getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.bufs = []
        rng = np.random.default_rng(seed)
        P, G = pictures, min(distinct, pictures)
        ls, cs = W * PX, (W // 2) * PX
        ysz, csz = ls * H, cs * (H // 2)
        # surfaces: reference (read by MC), recon (MC/pred + residual + deblock, in place), out (SAO)
        self.ref_y, self.ref_c = self.replicated(rng.integers(0, 1 << BD, (G, H, W), dtype=np.uint16), P), \
            self.replicated(rng.integers(0, 1 << BD, (G, 2, H // 2, W // 2), dtype=np.uint16), P)
        self.rec_y, self.rec_c = self.alloc(P * ysz), self.alloc(P * 2 * csz)
        self.out_y, self.out_c = self.alloc(P * ysz), self.alloc(P * 2 * csz)
        pic = np.arange(P, dtype=np.uint64)
        # ---- prediction units: one 32x32 luma block + its two 16x16 chroma blocks, vectors U[-64, 63] quarter samples
        by, bx = np.meshgrid(np.arange(H // 32), np.arange(W // 32), indexing="ij")
        n32 = by.size
        mvx, mvy = rng.integers(-64, 64, (P, n32)), rng.integers(-64, 64, (P, n32))
        x = np.clip(bx.reshape(-1)[None, :] * 32 + (mvx >> 2), 8, W - 40)
        y = np.clip(by.reshape(-1)[None, :] * 32 + (mvy >> 2), 8, H - 40)
        mp = np.zeros((P, n32, 3), MP_DT)
        mp["src0"][:, :, 0] = self.ref_y + pic[:, None] * ysz + (y * ls + x * PX).astype(np.uint64)
        mp["dst"][:, :, 0] = self.rec_y + pic[:, None] * ysz + (by.reshape(-1) * 32 * ls + bx.reshape(-1) * 32 * PX).astype(np.uint64)[None, :]
        mp["s0"][:, :, 0] = mp["ds"][:, :, 0] = ls
        mp["width"][:, :, 0] = mp["height"][:, :, 0] = 32
        mp["mx0"][:, :, 0], mp["my0"][:, :, 0] = mvx & 3, mvy & 3
        for pl in range(2):
            k = 1 + pl
            mp["src0"][:, :, k] = self.ref_c + (pic[:, None] * 2 + pl) * csz + ((y // 2) * cs + (x // 2) * PX).astype(np.uint64)
            mp["dst"][:, :, k] = self.rec_c + (pic[:, None] * 2 + pl) * csz + (by.reshape(-1) * 16 * cs + bx.reshape(-1) * 16 * PX).astype(np.uint64)[None, :]
            mp["s0"][:, :, k] = mp["ds"][:, :, k] = cs
            mp["width"][:, :, k] = mp["height"][:, :, k] = 16
            mp["chroma"][:, :, k] = 1
            mp["mx0"][:, :, k], mp["my0"][:, :, k] = mvx & 7, mvy & 7
        self.n_mp, self.d_mp = mp.size, self.up(mp)
        # ---- transform units: 32x32, all coded; 75 % carry non-zeros in the top-left 8x8 only (col_limit 12)
        ncy, ncx = np.meshgrid(np.arange(H // 64), np.arange(W // 64), indexing="ij")
        n64 = ncy.size
        n_tu = P * (n32 + 2 * n64)
        sparse = rng.random(n_tu) < 0.75
        coef = np.zeros((G * (n32 + 2 * n64), 32, 32), np.int16)          # coefficient blocks of the distinct pictures, replicated
        sp = sparse[:coef.shape[0]]
        coef[:, :8, :8] = np.clip(np.rint(rng.laplace(0, 64, (coef.shape[0], 8, 8))), -32767, 32767).astype(np.int16)
        dn = np.flatnonzero(~sp)
        coef[dn] = np.clip(np.rint(rng.laplace(0, 64, (len(dn), 32, 32))), -32767, 32767).astype(np.int16)
        per_pic = n32 + 2 * n64
        d_coef = self.replicated(coef.reshape(G, per_pic, 1024), P)
        sparse = np.tile(sp.reshape(G, per_pic), ((P + G - 1) // G, 1))[:P]
        tu = np.zeros((P, per_pic), TU_DT)
        tu["coeffs"] = d_coef + (pic[:, None] * per_pic + np.arange(per_pic, dtype=np.uint64)[None, :]) * 2048
        tu["dst"][:, :n32] = self.rec_y + pic[:, None] * ysz + (by.reshape(-1) * 32 * ls + bx.reshape(-1) * 32 * PX).astype(np.uint64)[None, :]
        tu["dst_stride"][:, :n32] = ls
        for pl in range(2):
            sl = slice(n32 + pl * n64, n32 + (pl + 1) * n64)
            tu["dst"][:, sl] = self.rec_c + (pic[:, None] * 2 + pl) * csz + (ncy.reshape(-1) * 32 * cs + ncx.reshape(-1) * 32 * PX).astype(np.uint64)[None, :]
            tu["dst_stride"][:, sl] = cs
        tu["log2_size"] = 5
        tu["col_limit"] = np.where(sparse, 12, 32)
        tu = tu.reshape(-1)
        tu = tu[np.argsort(tu["col_limit"], kind="stable")]                # the bridge bins its list by pruning class
        self.n_tu, self.d_tu = tu.size, self.up(tu)
        # ---- deblocking from frame-level arrays: bS on the 32x32 grid (TU and PU edges coincide), QP per 8x8 block
        bs_w, bs_h = W >> 3, H >> 3
        nbs = 2 * bs_w * (bs_h + 1)
        descs = (HFC.LfPicture * P)()
        for g in range(G):
            v = np.zeros(nbs, np.uint8)
            h = np.zeros(nbs, np.uint8)
            vv = v[:(H >> 2) * bs_w].reshape(H >> 2, bs_w)                 # [(y >> 2), (x >> 3)]
            vv[:, 4::4] = np.where(rng.random((H >> 2, (bs_w - 1) // 4)) < 0.1, 2, 1)     # x = 32, 64, ...
            hh = h[:(H * bs_w) >> 2].reshape(H >> 3, 2 * bs_w)           # index (x + y * bs_w) >> 2 for y multiple of 8
            hh[4::4, :] = np.where(rng.random(hh[4::4, :].shape) < 0.1, 2, 1)                # y = 32, 64, ...
            qp = rng.integers(22, 38, (H >> 3) * (W >> 3)).astype(np.int8)
            db = np.zeros(((W + 63) // 64) * ((H + 63) // 64), HFC.DBParams)
            dv, dh, dq, dd = self.up(v), self.up(h), self.up(qp), self.up(np.zeros(db.size * 2, np.int32))
            for p in range(g, P, G):
                d = descs[p]
                d.data[0], d.data[1], d.data[2] = self.rec_y + p * ysz, self.rec_c + (2 * p) * csz, self.rec_c + (2 * p + 1) * csz
                d.linesize[0], d.linesize[1], d.linesize[2] = ls, cs, cs
                d.width, d.height, d.log2_ctb_size, d.log2_min_cb_size, d.log2_min_pu_size = W, H, 6, 3, 2
                d.min_cb_width, d.min_pu_width, d.min_pu_height, d.ctb_width, d.bs_width = W >> 3, W >> 2, H >> 2, (W + 63) // 64, bs_w
                d.vertical_bs, d.horizontal_bs, d.qp_y_tab, d.is_pcm, d.deblock = dv, dh, dq, None, dd
        self.d_lf = self.up(np.frombuffer(bytes(descs), np.uint8))
        # ---- SAO: the class-0 region of every interior CTB, luma and both chroma planes
        cy, cx = np.meshgrid(np.arange(1, H // 64 - 1), np.arange(1, W // 64 - 1), indexing="ij")
        nct = cy.size
        sao = np.zeros((P, 3, nct), SAO_DT)
        kind = rng.random((P, 3, nct))
        sao["edge"] = kind < 0.5
        sao["offset_val"][..., 1:] = np.where((kind < 0.75)[..., None], rng.integers(-7, 8, (P, 3, nct, 4)) << (BD - 8), 0)
        sao["eo_class"] = rng.integers(0, 4, (P, 3, nct))
        sao["band_position"] = rng.integers(0, 32, (P, 3, nct))
        o_y = (cy.reshape(-1) * 64 * ls + cx.reshape(-1) * 64 * PX).astype(np.uint64)
        o_c = (cy.reshape(-1) * 32 * cs + cx.reshape(-1) * 32 * PX).astype(np.uint64)
        sao["src"][:, 0], sao["dst"][:, 0] = self.rec_y + pic[:, None] * ysz + o_y, self.out_y + pic[:, None] * ysz + o_y
        sao["stride"][:, 0], sao["width"][:, 0], sao["height"][:, 0] = ls, 64, 64
        for pl in range(2):
            sao["src"][:, 1 + pl] = self.rec_c + (pic[:, None] * 2 + pl) * csz + o_c
            sao["dst"][:, 1 + pl] = self.out_c + (pic[:, None] * 2 + pl) * csz + o_c
            sao["stride"][:, 1 + pl], sao["width"][:, 1 + pl], sao["height"][:, 1 + pl] = cs, 32, 32
            sao["c_idx"][:, 1 + pl] = 1 + pl
        self.n_sao, self.d_sao = sao.size, self.up(sao)
        self.ctbs = P * ((W + 63) // 64) * ((H + 63) // 64)
        lib.mi355_hevc_deblock_pictures_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]

    def alloc(self, n):
        p = self.lib.mi355_malloc(int(n) + 64)
        assert p, "device allocation of %d bytes failed" % n
        self.bufs.append(p)
        return p

    def up(self, a):
        a = np.ascontiguousarray(a)
        p = self.alloc(a.nbytes)
        assert self.lib.mi355_memcpy_h2d(p, a.ctypes.data, a.nbytes) == 0
        return p

    def replicated(self, a, n):
        """a[0..G) uploaded once and copied on the device to n items"""
        a = np.ascontiguousarray(a)
        G, item = a.shape[0], a.nbytes // a.shape[0]
        base = self.alloc(n * item)
        assert self.lib.mi355_memcpy_h2d(base, a.ctypes.data, a.nbytes) == 0
        done = G
        while done < n:
            k = min(done, n - done)
            assert self.lib.mi355_memcpy_d2d(base + done * item, base, k * item) == 0
            done += k
        return base

    def run(self):
        L = self.lib
        assert L.mi355_hevc_mcpred_batch_dev(C.c_void_p(self.d_mp), self.n_mp, BD, None) == 0
        assert L.mi355_hevc_residual_batch_dev(C.c_void_p(self.d_tu), self.n_tu, BD, None) == 0
        assert L.mi355_hevc_deblock_pictures_dev(C.c_void_p(self.d_lf), self.P, W, H, BD, None) == 0
        assert L.mi355_hevc_sao_batch_dev(C.c_void_p(self.d_sao), self.n_sao, BD, None) == 0

    def free(self):
        for p in self.bufs:
            self.lib.mi355_free(p)
        self.bufs = []


def measure(lib, pictures=64, steps=3):
    lib.mi355_event_create.restype = C.c_void_p
    lib.mi355_event_elapsed_ms.restype = C.c_float
    ch = Chain(lib, pictures)
    try:
        ch.run()
        lib.mi355_sync(None)
        e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
        lib.mi355_event_record(C.c_void_p(e0), None)
        for _ in range(steps):
            ch.run()
        lib.mi355_event_record(C.c_void_p(e1), None)
        lib.mi355_sync(None)
        ms = lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1)) / steps
        return {"name": "config3_hevc_2160p10_chain", "pictures_per_step": pictures, "ms_per_step": ms, "pictures_per_s": pictures / ms * 1e3,
                "ctb_per_s": ch.ctbs / ms * 1e3, "macroblock_equivalents_per_s": 16 * ch.ctbs / ms * 1e3,
                "algorithmic_bytes_per_ctb": BYTES_PER_CTB, "fraction_of_hbm_roofline": ch.ctbs * BYTES_PER_CTB / (ms * 1e-3) / 8e12,
                "note": "fused MC+pred, idct32+add_residual, picture-level deblocking (beta / tc derived on the device from bS arrays and "
                        "qp_y_tab), SAO: one back-to-back sequence, each stage reading what the previous one wrote; launch gaps included"}
    finally:
        ch.free()


if __name__ == "__main__":
    import json
    import libav_amd
    print(json.dumps(measure(libav_amd.load(0), int(sys.argv[1]) if len(sys.argv) > 1 else 64)))
