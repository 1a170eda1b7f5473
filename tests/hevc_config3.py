"""BASELINE config 3 as one picture-level chain through the batched HEVC entry points (include/mi355_hevc_batch.h),
checked against the oracle's table functions called in the same order: MC of every 32x32 PU (+ its two 16x16 chroma
blocks) -> put_unweighted_pred -> 32x32 idct + residual add -> luma and chroma edges (vertical pass, then horizontal)
-> SAO of the interior CTBs.  Used at full size (3840x2160, 10 bit) on the GPU and at a reduced size on the emulator."""
import ctypes as C

import numpy as np

import abi_ctypes as A
import hevc_batch as HB


def _ptr(a, off=0):
    return a.ctypes.data + off


def build(W, H, bd, seed):
    r = np.random.default_rng(seed)
    hi = (1 << bd) - 1
    dt = np.uint16 if bd > 8 else np.uint8
    m = dict(W=W, H=H, bd=bd, px=2 if bd > 8 else 1)
    # smooth-ish content so that the deblocking decisions go both ways, plus noise for SAO classes
    base = r.integers(0, hi + 1, (H // 8 + 1, W // 8 + 1), dtype=np.int32)
    up = np.repeat(np.repeat(base, 8, 0), 8, 1)[:H, :W]
    m["ref_y"] = np.clip(up + r.integers(-6, 7, (H, W), dtype=np.int32), 0, hi).astype(dt)
    m["ref_c"] = np.clip(up[::2, ::2][None] + r.integers(-6, 7, (2, H // 2, W // 2), dtype=np.int32), 0, hi).astype(dt)
    m["cur_y"] = r.integers(0, hi + 1, (H, W)).astype(dt)
    m["cur_c"] = r.integers(0, hi + 1, (2, H // 2, W // 2)).astype(dt)
    m["out_y"] = np.full((H, W), 0x155 & hi, dt)
    m["out_c"] = np.full((2, H // 2, W // 2), 0x155 & hi, dt)
    by, bx = H // 32, W // 32
    nb = by * bx
    m["blocks"] = [(y, x) for y in range(by) for x in range(bx)]
    m["mv"] = r.integers(-64, 64, (nb, 2))
    m["i16"] = np.zeros((nb, 32 * 32 + 2 * 16 * 16), np.int16)
    # coefficients: luma TU per block, chroma TU (32x32 of the half-size plane) per 64x64
    cby, cbx = H // 64, W // 64
    m["ctus"] = [(pl, y, x) for pl in range(2) for y in range(cby) for x in range(cbx)]
    ntu = nb + len(m["ctus"])
    coef = np.zeros((ntu, 32, 32), np.int16)
    sparse = r.random(ntu) < 0.75
    coef[:, :8, :8] = np.clip(np.rint(r.laplace(0, 64, (ntu, 8, 8))), -32767, 32767)
    dense = np.flatnonzero(~sparse)
    coef[dense] = np.clip(np.rint(r.laplace(0, 64, (len(dense), 32, 32))), -32767, 32767)
    m["coef"], m["sparse"] = coef.reshape(ntu, 1024), sparse
    # edges: (plane 0 luma / 1,2 chroma, horizontal, y, x of the first q-side sample, beta, tc0, tc1)
    edges = {}
    for horiz in (0, 1):
        parts = []
        gy, gx = np.mgrid[0:H // 8, 0:W // 8]
        keep = (gy if horiz else gx) != 0
        parts.append(np.stack([np.zeros(int(keep.sum()), np.int64), gy[keep] * 8, gx[keep] * 8], 1))
        for pl in (1, 2):
            gy, gx = np.mgrid[0:H // 16, 0:W // 16]
            keep = ((gy if horiz else gx) != 0) & (r.random(gy.shape) < 0.1)
            parts.append(np.stack([np.full(int(keep.sum()), pl, np.int64), gy[keep] * 8, gx[keep] * 8], 1))
        edges[horiz] = np.concatenate(parts).astype(np.int32)
    m["edges"] = edges
    m["edge_par"] = {h: np.stack([r.integers(20, 60, len(e)), r.integers(1, 12, len(e)), r.integers(1, 12, len(e))], 1).astype(np.int32)
                     for h, e in edges.items()}
    sao = []
    for c_idx in range(3):
        s = 64 if c_idx == 0 else 32
        for cy in range(1, H // 64 - 1):
            for cx in range(1, W // 64 - 1):
                sao.append((c_idx, cy * s, cx * s, s, int(r.random() < 0.67), int(r.integers(0, 4)), int(r.integers(0, 32)),
                            [0] + [int(v) for v in r.integers(-7 << (bd - 8), (7 << (bd - 8)) + 1, 4)]))
    m["sao"] = sao
    return m


def _plane(m, name, pl):
    """(array, byte offset of the plane, stride in bytes) for plane 0 (luma) / 1, 2 (chroma) of surface `name`"""
    if pl == 0:
        a = m[name + "_y"]
        return a, 0, a.strides[0]
    a = m[name + "_c"]
    return a, (pl - 1) * a.strides[0], a.strides[1]


def mc_geometry(m, k):
    """source position of block k's luma PU: the whole vector applied, clamped so that the taps stay inside the plane"""
    y, x = m["blocks"][k]
    mvx, mvy = int(m["mv"][k][0]), int(m["mv"][k][1])
    sx = min(max(x * 32 + (mvx >> 2), 8), m["W"] - 32 - 8) & ~1
    sy = min(max(y * 32 + (mvy >> 2), 8), m["H"] - 32 - 8) & ~1
    return sy, sx, mvx, mvy


def run_oracle(oracle, m):
    c = oracle.hevcdsp(m["bd"])
    px = m["px"]
    mcbuf = np.zeros((64 + 24) * 64, np.int16)
    i16 = m["i16"]
    for k, (y, x) in enumerate(m["blocks"]):
        sy, sx, mvx, mvy = mc_geometry(m, k)
        a, o, st = _plane(m, "ref", 0)
        c.put_hevc_qpel[int((mvy & 3) != 0)][int((mvx & 3) != 0)][5](HB._i16p(i16[k]), 64, HB._u8p(a, o + sy * st + sx * px), st, 32, mvx & 3, mvy & 3, HB._i16p(mcbuf))
        d, do, dst = _plane(m, "cur", 0)
        c.put_unweighted_pred[5](HB._u8p(d, do + y * 32 * dst + x * 32 * px), dst, HB._i16p(i16[k]), 64, 32)
        for pl in (1, 2):
            a, o, st = _plane(m, "ref", pl)
            off = 1024 + (pl - 1) * 256
            c.put_hevc_epel[int((mvy & 7) != 0)][int((mvx & 7) != 0)][5](HB._i16p(i16[k], off), 32, HB._u8p(a, o + (sy // 2) * st + (sx // 2) * px), st, 16,
                                                                       mvx & 7, mvy & 7, HB._i16p(mcbuf))
            d, do, dst = _plane(m, "cur", pl)
            c.put_unweighted_pred_chroma[5](HB._u8p(d, do + y * 16 * dst + x * 16 * px), dst, HB._i16p(i16[k], off), 32, 16)
    coef = m["coef"].copy()
    nb = len(m["blocks"])
    for k, (y, x) in enumerate(m["blocks"]):
        c.idct[3](HB._i16p(coef[k]), 12 if m["sparse"][k] else 32)
        d, do, dst = _plane(m, "cur", 0)
        c.add_residual[3](HB._u8p(d, do + y * 32 * dst + x * 32 * px), HB._i16p(coef[k]), dst)
    for t, (pl, y, x) in enumerate(m["ctus"]):
        k = nb + t
        c.idct[3](HB._i16p(coef[k]), 12 if m["sparse"][k] else 32)
        d, do, dst = _plane(m, "cur", 1 + pl)
        c.add_residual[3](HB._u8p(d, do + y * 32 * dst + x * 32 * px), HB._i16p(coef[k]), dst)
    zero = np.zeros(2, np.uint8)
    zp = HB._u8p(zero)
    for horiz in (0, 1):
        tcs = np.ascontiguousarray(m["edge_par"][horiz][:, 1:3])
        fl = c.hevc_h_loop_filter_luma if horiz else c.hevc_v_loop_filter_luma
        fc = c.hevc_h_loop_filter_chroma if horiz else c.hevc_v_loop_filter_chroma
        planes = [_plane(m, "cur", pl) for pl in range(3)]
        for i, (pl, y, x) in enumerate(m["edges"][horiz].tolist()):
            d, do, dst = planes[pl]
            pix = C.cast(d.ctypes.data + do + y * dst + x * px, A.u8p)
            tcp = C.cast(tcs.ctypes.data + 8 * i, A.intp)
            if pl == 0:
                fl(pix, dst, int(m["edge_par"][horiz][i][0]), tcp, zp, zp)
            else:
                fc(pix, dst, tcp, zp, zp)
    bo = np.zeros(4, np.int32)
    for c_idx, y, x, s, edge, eo, band, off in m["sao"]:
        sao = A.SAOParams()
        for i in range(5):
            sao.offset_val[c_idx][i] = off[i]
        sao.band_position[c_idx] = band
        sao.eo_class[c_idx] = eo
        a, o, st = _plane(m, "cur", c_idx)
        d, do, dst = _plane(m, "out", c_idx)
        if edge:
            c.sao_edge_filter[0](HB._u8p(d, do + y * dst + x * px), HB._u8p(a, o + y * st + x * px), st, C.byref(sao), C.cast(bo.ctypes.data, A.intp), s, s, c_idx, 0, 0, 0)
        else:
            c.sao_band_filter[0](HB._u8p(d, do + y * dst + x * px), HB._u8p(a, o + y * st + x * px), st, C.byref(sao), C.cast(bo.ctypes.data, A.intp), s, s, c_idx)
    return m["cur_y"], m["cur_c"], m["out_y"], m["out_c"]


def run_device(prov, m):
    lib, bd, px = prov.lib, m["bd"], m["px"]
    d = HB.Dev(lib)
    try:
        p = {n: d.up(m[n]) for n in ("ref_y", "ref_c", "cur_y", "cur_c", "out_y", "out_c", "i16", "coef")}

        def dev(name, pl):
            a, o, st = _plane(m, name, pl)
            return p[name + ("_y" if pl == 0 else "_c")] + o, st
        mcs, preds, tus = [], [], []
        nb = len(m["blocks"])
        for k, (y, x) in enumerate(m["blocks"]):
            sy, sx, mvx, mvy = mc_geometry(m, k)
            base = p["i16"] + k * m["i16"].strides[0]
            a, st = dev("ref", 0)
            mcs.append(HB.McJob(a + sy * st + sx * px, base, st, 64, 32, 32, mvx & 3, mvy & 3, 0))
            dd, dst = dev("cur", 0)
            preds.append(HB.PredJob(dd + y * 32 * dst + x * 32 * px, base, 0, dst, 64, 32, 32, 0, 0, 0, 0, 0, 0))
            tus.append(HB.TuJob(p["coef"] + k * 2048, dd + y * 32 * dst + x * 32 * px, dst, 5, 12 if m["sparse"][k] else 32, 0, 0))
            for pl in (1, 2):
                a, st = dev("ref", pl)
                off = 2 * (1024 + (pl - 1) * 256)
                mcs.append(HB.McJob(a + (sy // 2) * st + (sx // 2) * px, base + off, st, 32, 16, 16, mvx & 7, mvy & 7, 1))
                dd, dst = dev("cur", pl)
                preds.append(HB.PredJob(dd + y * 16 * dst + x * 16 * px, base + off, 0, dst, 32, 16, 16, 0, 0, 0, 0, 0, 0))
        for t, (pl, y, x) in enumerate(m["ctus"]):
            k = nb + t
            dd, dst = dev("cur", 1 + pl)
            tus.append(HB.TuJob(p["coef"] + k * 2048, dd + y * 32 * dst + x * 32 * px, dst, 5, 12 if m["sparse"][k] else 32, 0, 0))
        assert lib.mi355_hevc_mc_batch_dev(C.c_void_p(d.up_jobs(mcs)), len(mcs), bd, None) == 0
        assert lib.mi355_hevc_pred_batch_dev(C.c_void_p(d.up_jobs(preds)), len(preds), bd, None) == 0
        assert lib.mi355_hevc_residual_batch_dev(C.c_void_p(d.up_jobs(tus)), len(tus), bd, None) == 0
        for horiz in (0, 1):
            arr = (HB.LfJob * len(m["edges"][horiz]))()
            planes = [dev("cur", pl) for pl in range(3)]
            for i, ((pl, y, x), (beta, t0, t1)) in enumerate(zip(m["edges"][horiz].tolist(), m["edge_par"][horiz].tolist())):
                dd, dst = planes[pl]
                j = arr[i]
                j.pix, j.stride, j.beta = dd + y * dst + x * px, dst, beta
                j.tc[0], j.tc[1] = t0, t1
                j.horizontal_edge, j.chroma = horiz, int(pl != 0)
            assert lib.mi355_hevc_deblock_batch_dev(C.c_void_p(d.up_struct(arr)), len(arr), bd, None) == 0
        sj = []
        for c_idx, y, x, s, edge, eo, band, off in m["sao"]:
            a, st = dev("cur", c_idx)
            dd, dst = dev("out", c_idx)
            j = HB.SaoJob(dd + y * dst + x * px, a + y * st + x * px, st, s, s)
            for i in range(5):
                j.offset_val[i] = off[i]
            j.cls, j.edge, j.c_idx, j.eo_class, j.band_position = 0, edge, c_idx, eo, band
            sj.append(j)
        assert lib.mi355_hevc_sao_batch_dev(C.c_void_p(d.up_jobs(sj)), len(sj), bd, None) == 0
        return (d.down(p["cur_y"], m["cur_y"]), d.down(p["cur_c"], m["cur_c"]), d.down(p["out_y"], m["out_y"]), d.down(p["out_c"], m["out_c"]))
    finally:
        d.free()


def check(prov, oracle, W, H, bd, seed):
    m = build(W, H, bd, seed)
    got = run_device(prov, m)          # reads the initial surfaces; the oracle then works on them in place
    want = run_oracle(oracle, m)
    for name, g, w in zip(("deblocked luma", "deblocked chroma", "SAO luma", "SAO chroma"), got, want):
        assert np.array_equal(g, w), "%s differs: %d samples" % (name, int((g != w).sum()))
    changed = int((want[2] != (0x155 & ((1 << bd) - 1))).sum())
    return changed
