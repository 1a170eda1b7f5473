"""BASELINE.json config 3 as ONE back-to-back chain on a batch of pictures (bench.py's `extra`): 10-bit 3840x2160, CTB 64,
per CTB four 32x32 luma + two 32x32 chroma transform units (75 % with non-zeros in the top-left 8x8 only), 32x32
uni-predicted prediction units, deblocking of the whole picture FROM ITS FRAME-LEVEL ARRAYS (bS 1 on every 32x32 TU / PU
edge, bS 2 on 10 %, QP ~ U{22..37}: beta / tc derived on the device by mi355_hevc_deblock_pictures_dev), SAO of EVERY
CTB (50 % edge, 25 % band, 25 % off; one job per CTB component = the up to four reference calls that make up the CTB's own
samples, with their border classes: mi355_hevc_sao_ctbs_dev).  Prediction blocks lie where their vectors put them: those that reach over a picture border go
through mi355_edge_emu_batch_dev first (emulated_edge_mc, as luma_mc / chroma_mc do, hevcdec.c:1555, 1613-1630).  Job arrays are numpy records laid out like the C structs; nothing loops per job.
Stages are enqueued one after the other on the null stream, each reading what the previous one wrote:
  edge emulation -> scratch;  fused MC + put_unweighted_pred -> recon;  idct32 + add_residual -> recon;  deblock (V then H) in place;
  SAO recon -> out."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hevc_filter_cases as HFC  # noqa: E402

W, H, BD, PX = 3840, 2160, 10, 2
TU_DT = np.dtype([("coeffs", "<u8"), ("dst", "<u8"), ("dst_stride", "<i4"), ("log2_size", "u1"), ("col_limit", "u1"), ("kind", "u1"), ("rsv", "u1")])
MP_DT = np.dtype([("src0", "<u8"), ("src1", "<u8"), ("dst", "<u8"), ("s0", "<i4"), ("s1", "<i4"), ("ds", "<i4"), ("width", "u1"), ("height", "u1"),
                  ("chroma", "u1"), ("kind", "u1"), ("mx0", "u1"), ("my0", "u1"), ("mx1", "u1"), ("my1", "u1"), ("denom", "u1"), ("rsv", "u1", 3),
                  ("w0", "<i2"), ("w1", "<i2"), ("o0", "<i2"), ("o1", "<i2"), ("src0_b", "<u8"), ("src1_b", "<u8"), ("dst_b", "<u8")])
SAO_DT = np.dtype([("dst", "<u8"), ("src", "<u8"), ("stride", "<i4"), ("width", "<i4"), ("height", "<i4"), ("borders", "<i4", 4),
                   ("offset_val", "<i4", 5), ("cls", "u1"), ("edge", "u1"), ("c_idx", "u1"), ("eo_class", "u1"), ("band_position", "u1"),
                   ("vert_edge", "u1"), ("horiz_edge", "u1"), ("diag_edge", "u1")])
PIECE_DT = np.dtype([("offset_val", "<i4", 5), ("cls", "u1"), ("type", "u1"), ("eo_class", "u1"), ("band_position", "u1"), ("vert_edge", "u1"),
                     ("horiz_edge", "u1"), ("diag_edge", "u1"), ("borders", "u1"), ("dx", "<i2"), ("dy", "<i2"), ("width", "<i2"), ("height", "<i2")])
FCTB_DT = np.dtype([("pic", "<i4"), ("x0", "<u2"), ("y0", "<u2"), ("sao", "<u4", 3)])
SAOC_DT = np.dtype([("dst", "<u8"), ("src", "<u8"), ("stride", "<i4"), ("c_idx", "u1"), ("npieces", "u1"), ("rsv", "u1", 2), ("piece", PIECE_DT, 4)])
EE_DT = np.dtype([("dst", "<u8"), ("src", "<u8"), ("dst_stride", "<i4"), ("src_stride", "<i4"), ("block_w", "<i4"), ("block_h", "<i4"),
                  ("src_x", "<i4"), ("src_y", "<i4"), ("w", "<i4"), ("h", "<i4")])
CTB_DT = np.dtype([("dst", "<u8", 3), ("stride", "<i4", 3), ("width", "<u2"), ("height", "<u2"), ("log2_ctb_size", "u1"), ("flags", "u1"), ("rsv", "u1", 2),
                   ("first_mc", "<u4"), ("n_mc", "<u4"), ("first_tu", "<u4"), ("n_tu", "<u4"), ("rsv1", "<u4")])
assert CTB_DT.itemsize == 64
assert TU_DT.itemsize == 24 and MP_DT.itemsize == 80 and SAO_DT.itemsize == 72 and PIECE_DT.itemsize == 36 and SAOC_DT.itemsize == 168 and EE_DT.itemsize == 48
BYTES_PER_CTB = 73984


class Chain:
    def __init__(self, lib, pictures, distinct=2, seed=0x265, width=W, height=H, bd=BD, fused=True, filter_fused=None):
        """width, height: multiples of 64 x 16 at least (the bench: 3840 x 2160); the parity tests run smaller pictures.  self.host keeps
        what the CPU side (oracle/ref_hevc_chain.c: the reference's own functions) needs to decode the same pictures."""
        W, H, BD, PX = width, height, bd, (2 if bd > 8 else 1)
        self.W, self.H, self.BD, self.PX = W, H, BD, PX
        self.fused = fused
        # MI355_CHAIN_FILTER_FUSED=1: deblocking + SAO of a coding tree block in one workgroup (mi355_hevc_filter_ctbs_dev: the deblocked picture never leaves the chip) instead
        # of the picture-level deblocking launches and the SAO launch.  Measured in round 6 (profiles/r06_experiments.md): 1.7 - 1.8 ms against 0.53 + 0.78 ms — not the default
        self.filter_fused = bool(os.environ.get("MI355_CHAIN_FILTER_FUSED")) if filter_fused is None else filter_fused
        self.recon_flags = 0 if os.environ.get("MI355_CHAIN_NO_PROMISE") else 1      # MI355_HEVC_RECON_UNIFORM: this generator makes 32x32 one-reference blocks and 32x32 units only
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
        sdt = np.uint16 if BD > 8 else np.uint8
        h_ref_y, h_ref_c = rng.integers(0, 1 << BD, (G, H, W), dtype=sdt), rng.integers(0, 1 << BD, (G, 2, H // 2, W // 2), dtype=sdt)
        self.ref_y, self.ref_c = self.replicated(h_ref_y, P), self.replicated(h_ref_c, P)
        self.ysz, self.csz, self.ls, self.cs = ysz, csz, ls, cs
        self.host = {"ref_y": h_ref_y, "ref_c": h_ref_c, "G": G}
        self.rec_y, self.rec_c = self.alloc(P * ysz), self.alloc(P * 2 * csz)
        self.out_y, self.out_c = self.alloc(P * ysz), self.alloc(P * 2 * csz)
        pic = np.arange(P, dtype=np.uint64)
        # ---- prediction units: one 32x32 luma block + its two 16x16 chroma blocks, vectors U[-64, 63] quarter samples
        by, bx = np.meshgrid(np.arange(H // 32), np.arange(W // 32), indexing="ij")
        n32 = by.size
        mvx, mvy = rng.integers(-64, 64, (P, n32)), rng.integers(-64, 64, (P, n32))
        self.host["mv"] = np.stack([mvx, mvy], axis=2).astype(np.int32)
        x = bx.reshape(-1)[None, :] * 32 + (mvx >> 2)                    # where the vectors point: up to 16 samples outside the picture
        y = by.reshape(-1)[None, :] * 32 + (mvy >> 2)
        mp = np.zeros((P, n32, 3), MP_DT)
        ee = []

        def plane_jobs(k, base, plane_bytes, plane_idx, stride, xs, ys, bw, fx, fy, before, after, pw, ph):
            """component k of every prediction block: source = the reference plane, or — for a window that reaches over a picture
            border (the decoder's test, hevcdec.c:1546-1566 / 1600-1640) — a scratch window filled by an edge-emulation job"""
            eb = np.where(fx != 0, before, 0) if before == 3 else np.full(fx.shape, before)          # qpel: extra only in a filtered direction
            et = np.where(fy != 0, before, 0) if before == 3 else np.full(fy.shape, before)
            ea_x = np.where(fx != 0, after, 0) if before == 3 else np.full(fx.shape, after)
            ea_y = np.where(fy != 0, after, 0) if before == 3 else np.full(fy.shape, after)
            if before == 3:
                cross = (xs < eb) | (ys < et) | (xs >= pw - bw - ea_x) | (ys >= ph - bw - ea_y)
            else:                                                          # chroma_mc compares y with EPEL_EXTRA_AFTER (its own quirk)
                cross = (xs < before) | (ys < after) | (xs >= pw - bw - after) | (ys >= ph - bw - after)
            src = base + plane_idx * plane_bytes + (ys * stride + xs * PX).astype(np.int64).astype(np.uint64)
            nz = int(cross.sum())
            if nz:
                est = 80 * PX                                             # EDGE_EMU_BUFFER_STRIDE samples
                buf = self.alloc(nz * 80 * est)
                slot = np.cumsum(cross.reshape(-1)).reshape(cross.shape) - 1
                j = np.zeros(nz, EE_DT)
                c = cross
                ex, ey = (eb + ea_x)[c], (et + ea_y)[c]
                j["dst"] = buf + slot[c].astype(np.uint64) * np.uint64(80 * est)
                j["src"] = (src[c].astype(np.int64) - (et[c] * stride + eb[c] * PX)).astype(np.uint64)
                j["dst_stride"], j["src_stride"] = est, stride
                j["block_w"], j["block_h"] = bw + ex, bw + ey
                j["src_x"], j["src_y"], j["w"], j["h"] = xs[c] - eb[c], ys[c] - et[c], pw, ph
                ee.append(j)
                src = src.copy()
                src[c] = j["dst"] + (et[c] * est + eb[c] * PX).astype(np.uint64)
                mp["s0"][:, :, k] = np.where(c, est, stride)
            else:
                mp["s0"][:, :, k] = stride
            mp["src0"][:, :, k] = src
        plane_jobs(0, self.ref_y, ysz, pic[:, None], ls, x, y, 32, mvx & 3, mvy & 3, 3, 4, W, H)
        mp["dst"][:, :, 0] = self.rec_y + pic[:, None] * ysz + (by.reshape(-1) * 32 * ls + bx.reshape(-1) * 32 * PX).astype(np.uint64)[None, :]
        mp["ds"][:, :, 0] = ls
        mp["width"][:, :, 0] = mp["height"][:, :, 0] = 32
        mp["mx0"][:, :, 0], mp["my0"][:, :, 0] = mvx & 3, mvy & 3
        for pl in range(2):
            k = 1 + pl
            # chroma_mc: the chroma vector is the luma vector in eighth samples (4:2:0)
            plane_jobs(k, self.ref_c, csz, pic[:, None] * 2 + pl, cs, bx.reshape(-1)[None, :] * 16 + (mvx >> 3), by.reshape(-1)[None, :] * 16 + (mvy >> 3),
                       16, mvx & 7, mvy & 7, 1, 2, W // 2, H // 2)
            mp["dst"][:, :, k] = self.rec_c + (pic[:, None] * 2 + pl) * csz + (by.reshape(-1) * 16 * cs + bx.reshape(-1) * 16 * PX).astype(np.uint64)[None, :]
            mp["ds"][:, :, k] = cs
            mp["width"][:, :, k] = mp["height"][:, :, k] = 16
            mp["chroma"][:, :, k] = 1
            mp["mx0"][:, :, k], mp["my0"][:, :, k] = mvx & 7, mvy & 7
        ee = np.concatenate(ee) if ee else np.zeros(0, EE_DT)
        self.n_ee, self.d_ee = ee.size, (self.up(ee) if ee.size else 0)
        # the two chroma blocks of a prediction unit as ONE job (chroma = 2: same vector, strides and — unweighted — parameters; the edge test
        # of a block is the same in both planes, so both read the picture or both read their windows)
        assert (mp["s0"][:, :, 1] == mp["s0"][:, :, 2]).all()
        pairs = mp[:, :, 1].copy()
        pairs["chroma"] = 2
        pairs["src0_b"], pairs["dst_b"] = mp["src0"][:, :, 2], mp["dst"][:, :, 2]
        mp_luma = mp[:, :, 0].copy()
        mp = np.concatenate([mp[:, :, 0].reshape(-1), pairs.reshape(-1)])
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
        self.host["coef"], self.host["col_limit"] = coef.reshape(G, per_pic, 1024), np.where(sp, 12, 32).astype(np.uint8).reshape(G, per_pic)
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
        if fused:
            self.fused_lists(mp_luma, pairs, tu, bx, by, n32, n64)
        tu = tu.reshape(-1)
        tu = tu[np.argsort(tu["col_limit"], kind="stable")]                # the bridge bins its list by pruning class
        self.n_tu, self.d_tu = tu.size, self.up(tu)
        # ---- deblocking from frame-level arrays: bS on the 32x32 grid (TU and PU edges coincide), QP per 8x8 block
        bs_w, bs_h = W >> 3, H >> 3
        nbs = 2 * bs_w * (bs_h + 1)
        descs = (HFC.LfPicture * P)()
        self.host["bs"] = []
        for g in range(G):
            v = np.zeros(nbs, np.uint8)
            h = np.zeros(nbs, np.uint8)
            vv = v[:(H >> 2) * bs_w].reshape(H >> 2, bs_w)                 # [(y >> 2), (x >> 3)]
            vv[:, 4::4] = np.where(rng.random((H >> 2, (bs_w - 1) // 4)) < 0.1, 2, 1)     # x = 32, 64, ...
            hh = h[:(H * bs_w) >> 2].reshape(H >> 3, 2 * bs_w)           # index (x + y * bs_w) >> 2 for y multiple of 8
            hh[4::4, :] = np.where(rng.random(hh[4::4, :].shape) < 0.1, 2, 1)                # y = 32, 64, ...
            qp = rng.integers(22, 38, (H >> 3) * (W >> 3)).astype(np.int8)
            db = np.zeros(((W + 63) // 64) * ((H + 63) // 64), HFC.DBParams)
            self.host["bs"].append((v, h, qp))
            dv, dh, dq, dd = self.up(v), self.up(h), self.up(qp), self.up(np.zeros(db.size * 2, np.int32))
            for p in range(g, P, G):
                d = descs[p]
                d.data[0], d.data[1], d.data[2] = self.rec_y + p * ysz, self.rec_c + (2 * p) * csz, self.rec_c + (2 * p + 1) * csz
                d.linesize[0], d.linesize[1], d.linesize[2] = ls, cs, cs
                d.width, d.height, d.log2_ctb_size, d.log2_min_cb_size, d.log2_min_pu_size = W, H, 6, 3, 2
                d.min_cb_width, d.min_pu_width, d.min_pu_height, d.ctb_width, d.bs_width = W >> 3, W >> 2, H >> 2, (W + 63) // 64, bs_w
                d.vertical_bs, d.horizontal_bs, d.qp_y_tab, d.is_pcm, d.deblock = dv, dh, dq, None, dd
        self.d_lf = self.up(np.frombuffer(bytes(descs), np.uint8))
        # ---- SAO: every CTB, luma and both chroma planes: one job per CTB component (copy + the pieces with their border classes);
        # parameters per CTB and component: 50 % edge, 25 % band, 25 % off; one slice, no tiles (no unfilterable edges)
        ncx, ncy = (W + 63) // 64, (H + 63) // 64
        cy, cx = np.meshgrid(np.arange(ncy), np.arange(ncx), indexing="ij")
        kind = rng.random((P, 3, ncy, ncx))
        typ = np.where(kind < 0.5, 2, np.where(kind < 0.75, 1, 0)).astype(np.uint8)
        offs = (rng.integers(-7, 8, (P, 3, ncy, ncx, 4)) << (BD - 8)).astype(np.int32)
        eo = rng.integers(0, 4, (P, 3, ncy, ncx)).astype(np.uint8)
        band = rng.integers(0, 32, (P, 3, ncy, ncx)).astype(np.uint8)
        self.host["sao"] = (typ, offs, eo, band)
        sao = np.zeros((P, 3, ncy, ncx), SAOC_DT)
        for c in range(3):
            sz_c, st_c = (64, ls) if c == 0 else (32, cs)
            wc, hc = (W, H) if c == 0 else (W // 2, H // 2)
            o = (cy * sz_c * st_c + cx * sz_c * PX).astype(np.uint64)[None]
            src_b = (self.rec_y + pic * ysz) if c == 0 else (self.rec_c + (pic * 2 + (c - 1)) * csz)
            dst_b = (self.out_y + pic * ysz) if c == 0 else (self.out_c + (pic * 2 + (c - 1)) * csz)
            sao["src"][:, c], sao["dst"][:, c] = src_b[:, None, None] + o, dst_b[:, None, None] + o
            sao["stride"][:, c] = st_c
            sao["c_idx"][:, c] = c
            # the owner's samples = class 0 of the CTB itself, class 2 of the CTB to its right, class 1 of the one below, class 3 of the one
            # below-right, all with the OWNER's parameters (one slice, no tiles: no unfilterable edges)
            npieces = np.zeros((ncy, ncx), np.int64)
            for k in (0, 2, 1, 3):
                px_, py_ = cx + (k >> 1), cy + (k & 1)                    # the CTB the reference makes this call for
                have = (px_ < ncx) & (py_ < ncy)
                for i in np.unique(npieces[have]):
                    yy, xx = np.nonzero(have & (npieces == i))
                    pc = sao["piece"][:, c, yy, xx, int(i)]
                    pc["cls"] = k
                    pc["type"], pc["eo_class"], pc["band_position"] = typ[:, c, yy, xx], eo[:, c, yy, xx], band[:, c, yy, xx]
                    pc["offset_val"][..., 1:] = offs[:, c, yy, xx]
                    qx, qy = px_[yy, xx], py_[yy, xx]
                    pc["borders"] = ((qx == 0) | ((qy == 0) << 1) | ((qx == ncx - 1) << 2) | ((qy == ncy - 1) << 3))[None]
                    pc["dx"], pc["dy"] = ((k >> 1) * sz_c), ((k & 1) * sz_c)
                    pc["width"], pc["height"] = np.minimum(sz_c, wc - qx * sz_c)[None], np.minimum(sz_c, hc - qy * sz_c)[None]
                    sao["piece"][:, c, yy, xx, int(i)] = pc
                npieces = npieces + have
            sao["npieces"][:, c] = npieces[None].astype(np.uint8)
        self.n_sao, self.d_sao = sao.size, self.up(sao)
        # the fused form's block list: picture, position, the three components' SAO jobs
        fctb = np.zeros((P, ncy, ncx), FCTB_DT)
        fctb["pic"] = np.arange(P, dtype=np.int32)[:, None, None]
        fctb["x0"], fctb["y0"] = (cx * 64)[None], (cy * 64)[None]
        for c in range(3):
            fctb["sao"][..., c] = ((np.arange(P)[:, None, None] * 3 + c) * ncy + cy[None]) * ncx + cx[None]
        self.n_fctb, self.d_fctb = fctb.size, self.up(fctb)
        self.lib.mi355_hevc_filter_ctbs_dev.restype = C.c_int
        self.lib.mi355_hevc_filter_ctbs_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        # credited units: the CTBs the chain codes COMPLETELY (prediction + residual of luma and chroma: H // 64 rows; VERDICT r3: the 34th,
        # partial CTB row of 2160 lines gets luma blocks for its first 32 lines only and no chroma units — filtered and SAO'd, not credited)
        self.ctbs = P * (W // 64) * (H // 64)
        lib.mi355_hevc_deblock_pictures_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]

    def fused_lists(self, mp_luma, pairs, tu, bx, by, n32, n64):
        """the same prediction jobs and transform units listed per coding tree block for mi355_hevc_recon_ctbs_dev: a block's luma prediction
        blocks, its chroma pairs; its luma transform units, its Cb and Cr units"""
        W, H, PX, P = self.W, self.H, self.PX, self.P
        ncx, ncy = W // 64, (H + 63) // 64
        ctb32 = ((by // 2) * ncx + (bx // 2)).reshape(-1)
        order = np.argsort(ctb32, kind="stable")
        cnt = np.bincount(ctb32, minlength=ncx * ncy)
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        c_sorted = ctb32[order]
        rank = np.arange(n32) - start[c_sorted]
        comb = np.zeros((P, 2 * n32), MP_DT)
        comb[:, 2 * start[c_sorted] + rank] = mp_luma[:, order]
        comb[:, 2 * start[c_sorted] + cnt[c_sorted] + rank] = pairs[:, order]
        has_c = (np.arange(ncx * ncy) // ncx) < (H // 64)              # the block has its two chroma units
        ntu = cnt + 2 * has_c
        tstart = np.concatenate([[0], np.cumsum(ntu)[:-1]])
        per_pic = n32 + 2 * n64
        tus = np.zeros((P, per_pic), TU_DT)
        tus[:, tstart[c_sorted] + rank] = tu[:, :n32][:, order]
        cc = np.flatnonzero(has_c)                                       # = the raster index of the 64x64 chroma units
        assert len(cc) == n64
        tus[:, tstart[cc] + cnt[cc]] = tu[:, n32:n32 + n64]
        tus[:, tstart[cc] + cnt[cc] + 1] = tu[:, n32 + n64:]
        live = np.flatnonzero(cnt > 0)
        ctb = np.zeros((P, len(live)), CTB_DT)
        cy, cx = live // ncx, live % ncx
        pic = np.arange(P, dtype=np.uint64)[:, None]
        ctb["dst"][:, :, 0] = self.rec_y + pic * self.ysz + (cy * 64 * self.ls + cx * 64 * PX).astype(np.uint64)[None]
        for pl in range(2):
            ctb["dst"][:, :, 1 + pl] = self.rec_c + (pic * 2 + pl) * self.csz + (cy * 32 * self.cs + cx * 32 * PX).astype(np.uint64)[None]
        ctb["stride"][:, :, 0], ctb["stride"][:, :, 1], ctb["stride"][:, :, 2] = self.ls, self.cs, self.cs
        ctb["width"], ctb["height"] = 64, np.minimum(64, H - 64 * cy)[None]
        ctb["log2_ctb_size"] = 6
        ctb["flags"] = np.where(cnt[live] < 4, 1, 0)[None]             # MI355_HEVC_CTB_PARTIAL: rows of the block that no prediction block covers
        ctb["first_mc"] = (np.arange(P) * 2 * n32)[:, None] + 2 * start[live][None]
        ctb["n_mc"] = 2 * cnt[live][None]
        ctb["first_tu"] = (np.arange(P) * per_pic)[:, None] + tstart[live][None]
        ctb["n_tu"] = ntu[live][None]
        self.n_ctb, self.d_ctb = ctb.size, self.up(ctb)
        self.d_mp_ctb, self.d_tu_ctb = self.up(comb), self.up(tus)
        self.lib.mi355_hevc_recon_ctbs_dev.restype = C.c_int
        self.lib.mi355_hevc_recon_ctbs_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_void_p]

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

    def run(self, stream=None, turn_stage=-1, wait_ev=None, record_ev=None):
        """turn_stage (with several chains on their own streams): stage 1 (MC + prediction), 2 (transform + residual) or 4 (SAO) waits for `wait_ev` and is followed by `record_ev`"""
        L = self.lib

        def around(stage, launch):
            if stage == turn_stage and wait_ev is not None:
                assert L.mi355_stream_wait_event(stream, wait_ev) == 0
            assert launch() == 0
            if stage == turn_stage and record_ev is not None:
                L.mi355_event_record(record_ev, stream)
        if self.n_ee:
            assert L.mi355_edge_emu_batch_dev(C.c_void_p(self.d_ee), self.n_ee, self.BD, stream) == 0
        if self.fused:
            # prediction + residual of a coding tree block in one workgroup: the block's samples leave for the picture once
            around(1, lambda: L.mi355_hevc_recon_ctbs_dev(C.c_void_p(self.d_ctb), self.n_ctb, C.c_void_p(self.d_mp_ctb), C.c_void_p(self.d_tu_ctb), self.BD, self.recon_flags, stream))
        else:
            around(1, lambda: L.mi355_hevc_mcpred_batch_dev(C.c_void_p(self.d_mp), self.n_mp, self.BD, stream))
            around(2, lambda: L.mi355_hevc_residual_batch_dev(C.c_void_p(self.d_tu), self.n_tu, self.BD, stream))
        if self.filter_fused:
            if os.environ.get("MI355_FT_SKIP_V"):          # developer experiment: vertical edges in the picture (MI355_DEBLOCK_DIRS=1), horizontal edges + SAO fused
                around(3, lambda: L.mi355_hevc_deblock_pictures_dev(C.c_void_p(self.d_lf), self.P, self.W, self.H, self.BD, stream))
            around(4, lambda: L.mi355_hevc_filter_ctbs_dev(C.c_void_p(self.d_lf), C.c_void_p(self.d_fctb), self.n_fctb, C.c_void_p(self.d_sao), 6, self.BD, stream))
        else:
            around(3, lambda: L.mi355_hevc_deblock_pictures_dev(C.c_void_p(self.d_lf), self.P, self.W, self.H, self.BD, stream))
            around(4, lambda: L.mi355_hevc_sao_ctbs_dev(C.c_void_p(self.d_sao), self.n_sao, self.BD, stream))

    def free(self):
        for p in self.bufs:
            self.lib.mi355_free(p)
        self.bufs = []


def measure_pipelines(lib, pictures=64, pipelines=2, steps=6, turn_stage=-1):
    """the same 64 pictures as `pipelines` chains of pictures / pipelines each on their own streams (developer experiment: profiles/r05_experiments.md 13)"""
    import time
    lib.mi355_stream_create.restype = C.c_void_p
    chains = [Chain(lib, pictures // pipelines, seed=0x265 + i, fused=not os.environ.get("MI355_CHAIN_SPLIT")) for i in range(pipelines)]
    streams = [C.c_void_p(lib.mi355_stream_create()) for _ in range(pipelines)]
    try:
        lib.mi355_stream_wait_event.restype = C.c_int
        lib.mi355_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
        lib.mi355_event_record.argtypes = [C.c_void_p, C.c_void_p]
        lib.mi355_event_create.restype = C.c_void_p
        turn = [C.c_void_p(lib.mi355_event_create()) for _ in range(pipelines)]
        started = [False]

        def step():
            for i, (ch, st) in enumerate(zip(chains, streams)):
                ch.run(st, turn_stage, turn[(i - 1) % pipelines] if (turn_stage > 0 and (i > 0 or started[0])) else None, turn[i] if turn_stage > 0 else None)
                started[0] = True

        def sync():
            for st in streams:
                lib.mi355_sync(st)
        step(); step(); sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        ctbs = sum(ch.ctbs for ch in chains)
        return {"pipelines": pipelines, "turn_stage": turn_stage, "pictures_per_step": pictures, "ms_per_step": ms, "fraction_of_hbm_roofline": ctbs * BYTES_PER_CTB / (ms * 1e-3) / 8e12}
    finally:
        for ch in chains:
            ch.free()


def measure(lib, pictures=64, steps=3, cpu_seconds=6.0):
    lib.mi355_event_create.restype = C.c_void_p
    lib.mi355_event_elapsed_ms.restype = C.c_float
    ch = Chain(lib, pictures, fused=not os.environ.get("MI355_CHAIN_SPLIT"))      # MI355_CHAIN_SPLIT=1: prediction and residual as two launches (rounds 3-5)
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
        cpu = cpu_baseline(ch, cpu_seconds) if cpu_seconds else None
        return {"name": "config3_hevc_2160p10_chain", "pictures_per_step": pictures, "ms_per_step": ms, "pictures_per_s": pictures / ms * 1e3,
                "ctb_per_s": ch.ctbs / ms * 1e3, "macroblock_equivalents_per_s": 16 * ch.ctbs / ms * 1e3,
                "algorithmic_bytes_per_ctb": BYTES_PER_CTB, "fraction_of_hbm_roofline": ch.ctbs * BYTES_PER_CTB / (ms * 1e-3) / 8e12,
                "edge_emulated_windows_per_picture": ch.n_ee / pictures, "cpu_baseline": cpu,
                "verified_by": "tests/test_hevc_chain_gpu.py (this chain at 3840x2160 against the reference's own functions, every sample)",
                "note": "edge emulation of the windows that cross a picture border, fused MC+pred, idct32+add_residual, picture-level deblocking "
                        "(beta / tc derived on the device from bS arrays and qp_y_tab), SAO of every CTB (copy + pieces per CTB component): one "
                        "back-to-back sequence, each stage reading what the previous one wrote; launch gaps included"}
    finally:
        ch.free()


if __name__ == "__main__":
    import json
    import libav_amd
    if len(sys.argv) > 2:
        print(json.dumps(measure_pipelines(libav_amd.load(0), int(sys.argv[1]), int(sys.argv[2]), turn_stage=int(sys.argv[3]) if len(sys.argv) > 3 else -1)))
    else:
        print(json.dumps(measure(libav_amd.load(0), int(sys.argv[1]) if len(sys.argv) > 1 else 64, cpu_seconds=0)))


# ---- the same pictures through the reference's own functions (oracle/ref_hevc_chain.c in oracle/_ref/libhevcfilterref.so) --------
class RefChain(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("bd", C.c_int32), ("reserved", C.c_int32),
                ("ref", C.c_void_p * 3), ("cur", C.c_void_p * 3), ("out", C.c_void_p * 3), ("stride", C.c_int32 * 3), ("reserved1", C.c_int32),
                ("mv", C.c_void_p), ("coef", C.c_void_p), ("col_limit", C.c_void_p), ("vertical_bs", C.c_void_p), ("horizontal_bs", C.c_void_p),
                ("qp_y_tab", C.c_void_p), ("sao_type", C.c_void_p), ("sao_offset", C.c_void_p), ("sao_eo", C.c_void_p), ("sao_band", C.c_void_p)]


def ref_library():
    path = os.path.join(ROOT, "oracle", "_ref", "libhevcfilterref.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    if not hasattr(lib, "ref_hevc_chain_run"):
        return None
    lib.ref_hevc_chain_run.restype = C.c_int
    lib.ref_hevc_chain_run.argtypes = [C.c_void_p]
    lib.ref_hevc_chain_bench_threads.restype = C.c_long
    lib.ref_hevc_chain_bench_threads.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p]
    return lib


def ref_chain_desc(ch, p, cur, out, keep):
    """RefChain for picture p of Chain `ch` writing into the host surfaces cur / out ([y, c] arrays: (H, W) and (2, H/2, W/2));
    `keep` collects the arrays the descriptor points at"""
    h = ch.host
    g = p % h["G"]
    d = RefChain()
    d.W, d.H, d.bd = ch.W, ch.H, ch.BD
    ry, rc = h["ref_y"][g], h["ref_c"][g]
    for i, (r, cu, o) in enumerate(((ry, cur[0], out[0]), (rc[0], cur[1][0], out[1][0]), (rc[1], cur[1][1], out[1][1]))):
        d.ref[i], d.cur[i], d.out[i] = r.ctypes.data, cu.ctypes.data, o.ctypes.data
        d.stride[i] = r.strides[0]
        assert cu.strides[0] == r.strides[0] and o.strides[0] == r.strides[0]
    v, hh, qp = h["bs"][g]
    typ, offs, eo, band = h["sao"]
    arrs = [np.ascontiguousarray(a) for a in (h["mv"][p], h["coef"][g], h["col_limit"][g], v, hh, qp, typ[p], offs[p], eo[p], band[p])]
    keep.extend(arrs)
    (d.mv, d.coef, d.col_limit, d.vertical_bs, d.horizontal_bs, d.qp_y_tab, d.sao_type, d.sao_offset, d.sao_eo, d.sao_band) = [a.ctypes.data for a in arrs]
    return d


def surfaces(ch, fill=0):
    dt = np.uint16 if ch.BD > 8 else np.uint8
    return [np.full((ch.H, ch.W), fill, dt), np.full((2, ch.H // 2, ch.W // 2), fill, dt)]


def check_against_reference(lib, pictures=2, width=256, height=192, bd=10, seed=0x265, fused=True, filter_fused=None):
    """the measured chain at any size: device pictures (reconstruction after deblocking, SAO output) of every picture against the
    reference's own functions on the same parameters.  Surfaces start zeroed on both sides (the chain leaves the rows below the
    last whole 32 / 64 block unpredicted, as the workload is defined)."""
    ref = ref_library()
    assert ref is not None, "oracle/_ref/libhevcfilterref.so (with ref_hevc_chain.c) is missing"
    lib.mi355_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    ch = Chain(lib, pictures, distinct=min(2, pictures), seed=seed, width=width, height=height, bd=bd, fused=fused, filter_fused=filter_fused)
    try:
        zero = np.zeros(pictures * ch.ysz, np.uint8)
        for base, n in ((ch.rec_y, ch.ysz), (ch.out_y, ch.ysz), (ch.rec_c, 2 * ch.csz), (ch.out_c, 2 * ch.csz)):
            assert lib.mi355_memcpy_h2d(base, zero.ctypes.data, pictures * n) == 0
        ch.run()
        assert lib.mi355_sync(None) == 0
        for p in range(pictures):
            cur, out, keep = surfaces(ch), surfaces(ch), []
            d = ref_chain_desc(ch, p, cur, out, keep)
            assert ref.ref_hevc_chain_run(C.byref(d)) == 0
            got_cur, got_out = surfaces(ch, 0xAA), surfaces(ch, 0xAA)
            for arr, base, n in ((got_cur[0], ch.rec_y + p * ch.ysz, ch.ysz), (got_cur[1], ch.rec_c + 2 * p * ch.csz, 2 * ch.csz),
                                 (got_out[0], ch.out_y + p * ch.ysz, ch.ysz), (got_out[1], ch.out_c + 2 * p * ch.csz, 2 * ch.csz)):
                assert lib.mi355_memcpy_d2h(arr.ctypes.data, base, n) == 0
            # the fused filters never write the deblocked picture: its surfaces keep the unfiltered reconstruction, and the output is what is compared
            for name, a, b in ((("deblocked luma", got_cur[0], cur[0]), ("deblocked chroma", got_cur[1], cur[1])) if not ch.filter_fused else ()) + (
                               ("SAO luma", got_out[0], out[0]), ("SAO chroma", got_out[1], out[1])):
                assert np.array_equal(a, b), "picture %d: %s differs from the reference's functions (%d samples)" % (p, name, int((a != b).sum()))
        return ch.n_ee
    finally:
        ch.free()


def cpu_baseline(ch, seconds=8.0):
    """the reference's own functions (kind "reference") decoding picture 0's parameters on every logical CPU this process may run on,
    one pinned thread and one private pair of surfaces per CPU, for ~`seconds`"""
    ref = ref_library()
    if ref is None:
        return None
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    n = len(cpus)
    keep, descs = [], (RefChain * n)()
    for t in range(n):
        cur, out = surfaces(ch), surfaces(ch)
        keep.extend([cur, out])
        descs[t] = ref_chain_desc(ch, t % ch.P, cur, out, keep)
    one = ref.ref_hevc_chain_run(C.byref(descs[0]))                    # warms tables and pages of thread 0's surfaces
    assert one == 0
    cpu_arr = (C.c_int * n)(*cpus)
    wall = C.c_double(0.0)
    done = ref.ref_hevc_chain_bench_threads(C.cast(descs, C.c_void_p), n, C.cast(cpu_arr, C.c_void_p), float(seconds), C.byref(wall))
    ctbs = ((ch.W + 63) // 64) * ((ch.H + 63) // 64)
    return {"value": done / wall.value, "unit": "pictures/s", "ctb_per_s": done * ctbs / wall.value, "cores": n, "kind": "reference",
            "sample": "%d pictures of the same chain decoded by the reference's own functions (hevcdsp / videodsp tables, ff_hevc_hls_filters: "
                      "oracle/ref_hevc_chain.c over libavcodec compiled in place, C paths) in %.1f s on %d pinned threads, one picture and one "
                      "private pair of surfaces per thread" % (done, wall.value, n)}
