#!/usr/bin/env python3
"""Throughput of the HEVC intra_pred wrapper kernel (mi355_hevc_intra_pred_blocks_dev) on 10-bit 2160p pictures: one launch
= every block of one of the four (even / odd column, even / odd row) classes of a uniform block grid over `--pictures`
pictures — members of a class do not read what another member writes.  HIP events; algorithmic bytes per block =
(4 * size + 1) neighbour samples read + size * size samples written.  Not the headline metric; numbers go to DESIGN.md."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hevc_intra_cases as IC  # noqa: E402
import providers  # noqa: E402

W, H, BD, PX = 3840, 2160, 10, 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pictures", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    lib = providers.mi355().lib
    lib.mi355_malloc.restype = C.c_void_p
    lib.mi355_malloc.argtypes = [C.c_size_t]
    lib.mi355_event_create.restype = C.c_void_p
    lib.mi355_event_elapsed_ms.restype = C.c_float
    fn = lib.mi355_hevc_intra_pred_blocks_dev
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rng = np.random.default_rng(0x18)

    def up(arr):
        arr = np.ascontiguousarray(arr)
        p = lib.mi355_malloc(max(arr.nbytes, 16))
        assert p and lib.mi355_memcpy_h2d(C.c_void_p(p), C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes)) == 0
        return p
    P = a.pictures
    zs, tbw = IC.zscan_table(W, H, 6, 2)
    d_zs = up(zs)
    mvf = np.zeros((H // 4) * (W // 4), IC.MVF_DT)
    mvf["is_intra"] = (rng.random(mvf.shape[0]) < 0.7).astype(np.uint8)
    d_mvf = up(mvf)
    luma = rng.integers(0, 1 << BD, (H, W), dtype=np.uint16)
    chroma = rng.integers(0, 1 << BD, (H // 2, W // 2), dtype=np.uint16)
    out = []
    for cip in (0, 1):
        descs = (IC.IntraPicture * P)()
        for i in range(P):
            d = descs[i]
            d.data[0], d.data[1], d.data[2] = up(luma), up(chroma), up(chroma)
            d.linesize[0], d.linesize[1], d.linesize[2] = W * PX, W // 2 * PX, W // 2 * PX
            d.width, d.height, d.hshift, d.vshift = W, H, 1, 1
            d.log2_min_pu_size, d.log2_min_tb_size = 2, 2
            d.min_pu_width, d.min_pu_height, d.min_tb_width = W // 4, H // 4, tbw
            d.constrained_intra_pred, d.strong_intra_smoothing = cip, 1
            d.tab_mvf, d.min_tb_addr_zs = d_mvf, d_zs
        d_desc = up(np.frombuffer(bytes(descs), np.uint8))
        for l2 in ((5, 4, 3, 2) if not cip else (4, 2)):
            n = 1 << l2
            blk = np.zeros(P * (W // n // 2) * (H // n // 2), np.dtype([("pic", "<i4"), ("x0", "<u2"), ("y0", "<u2"), ("l2", "u1"), ("c", "u1"), ("mode", "u1"), ("cand", "u1")]))
            k = 0
            for p in range(P):
                ys, xs = np.mgrid[0:H // n // 2, 0:W // n // 2]
                m = ys.size
                v = blk[k:k + m]
                v["pic"], v["x0"], v["y0"], v["l2"], v["c"] = p, (xs.reshape(-1) * 2 + 1) * n, (ys.reshape(-1) * 2 + 1) * n, l2, 0
                v["mode"] = rng.integers(0, 35, m)
                v["cand"] = 31
                k += m
            blk = blk[(blk["x0"] + 2 * n <= W) & (blk["y0"] + 2 * n <= H)]
            d_blk = up(blk)
            cnt = len(blk)
            assert fn(d_desc, d_blk, cnt, BD, None) == 0 and lib.mi355_sync(None) == 0
            e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
            lib.mi355_event_record(C.c_void_p(e0), None)
            for _ in range(a.steps):
                fn(d_desc, d_blk, cnt, BD, None)
            lib.mi355_event_record(C.c_void_p(e1), None)
            lib.mi355_sync(None)
            ms = lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1)) / a.steps
            byts = (n * n + 4 * n + 1) * PX
            out.append({"stage": "intra_pred wrapper %dx%d luma%s" % (n, n, ", constrained intra" if cip else ""), "blocks_per_launch": cnt,
                        "ms_per_launch": ms, "blocks_per_s": cnt / ms * 1e3, "samples_per_s": cnt * n * n / ms * 1e3,
                        "algorithmic_GBps": cnt * byts / ms / 1e6, "frac_of_8TBps": cnt * byts / ms / 1e6 / 8000})
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
