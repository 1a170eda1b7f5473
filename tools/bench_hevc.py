#!/usr/bin/env python3
"""Throughput of the batched HEVC kernels on a BASELINE.json config 3 shaped workload (10-bit 2160p,
CTB 64, 32x32 transform units, 32x32 uni-predicted PUs, all 8-sample edges of the 8x8 grid, SAO on every
CTB) — not the headline metric; numbers go to DESIGN.md.  Each stage is one launch over `--pictures`
pictures resident in HBM, timed with HIP events; algorithmic bytes per stage as in SURVEY.md §8d
(every input byte read once, every output byte written once).  The CPU oracle is timed on a sample."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hevc_batch as HB  # noqa: E402
import providers  # noqa: E402

W, H, BD, PX = 3840, 2160, 10, 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pictures", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    prov = providers.mi355()
    lib = prov.lib
    lib.mi355_event_create.restype = C.c_void_p
    lib.mi355_event_elapsed_ms.restype = C.c_float
    P = a.pictures
    d = HB.Dev(lib)
    rng = np.random.default_rng(0x265)
    stride = W * PX
    ysz, csz = stride * H, (stride // 2) * (H // 2)
    pic = rng.integers(0, 1 << BD, (P, H, W), dtype=np.uint16)
    p_a, p_b = d.up(pic), d.up(pic)                  # two luma surfaces (ref / recon, deblocked / SAO out)
    cpic = rng.integers(0, 1 << BD, (P, 2, H // 2, W // 2), dtype=np.uint16)
    p_ca, p_cb = d.up(cpic), d.up(cpic)
    del pic, cpic
    results = []

    def timed(name, fn, units, bytes_per_unit):
        fn()
        lib.mi355_sync(None)
        e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
        lib.mi355_event_record(C.c_void_p(e0), None)
        for _ in range(a.steps):
            fn()
        lib.mi355_event_record(C.c_void_p(e1), None)
        lib.mi355_sync(None)
        ms = lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1)) / a.steps
        results.append({"stage": name, "jobs_per_launch": units, "ms_per_launch": ms, "jobs_per_s": units / ms * 1e3,
                        "algorithmic_GBps": units * bytes_per_unit / ms / 1e6, "frac_of_8TBps": units * bytes_per_unit / ms / 1e6 / 8000,
                        "pictures_per_s": P / ms * 1e3})

    # ---- transform units: 32x32 luma + 32x32 per chroma plane, 75 % with col_limit <= 12 -----------------
    tus = []
    nl = (H // 32) * (W // 32)
    ncp = (H // 64) * (W // 64)
    n_tu = P * (nl + 2 * ncp)
    coef = np.zeros((n_tu, 1024), np.int16)
    sparse = rng.random(n_tu) < 0.75
    lap = np.clip(np.rint(rng.laplace(0, 64, (n_tu, 8, 8))), -32767, 32767).astype(np.int16)
    coef.reshape(n_tu, 32, 32)[:, :8, :8] = lap
    dense = np.flatnonzero(~sparse)
    coef[dense] = np.clip(np.rint(rng.laplace(0, 64, (len(dense), 1024))), -32767, 32767).astype(np.int16)
    p_coef = d.up(coef)
    k = 0
    for p in range(P):
        for by in range(H // 32):
            for bx in range(W // 32):
                tus.append(HB.TuJob(p_coef + k * 2048, p_a + p * ysz + by * 32 * stride + bx * 32 * PX, stride, 5, 12 if sparse[k] else 32, 0, 0))
                k += 1
        for pl in range(2):
            for by in range(H // 64):
                for bx in range(W // 64):
                    tus.append(HB.TuJob(p_coef + k * 2048, p_ca + (p * 2 + pl) * csz + by * 32 * (stride // 2) + bx * 32 * PX, stride // 2, 5,
                                        12 if sparse[k] else 32, 0, 0))
                    k += 1
    assert k == n_tu
    # two transform units share a wave: a bridge bins its job list by (size, col_limit class) so that both halves take the
    # same (pruned or full) path; the jobs are independent, their order is the caller's choice
    tus.sort(key=lambda j: j.col_limit)
    p_tus = d.up_jobs(tus)
    timed("idct32 + add_residual", lambda: lib.mi355_hevc_residual_batch_dev(C.c_void_p(p_tus), n_tu, BD, None), n_tu, 2048 + 2048 + 2048)

    # ---- MC: one 32x32 uni-predicted PU per 32x32 luma block + its two 16x16 chroma blocks --------------------
    mcs = []
    n_mc = P * nl * 3
    p_i16 = d.up(np.zeros((P * nl, 32 * 32 + 2 * 16 * 16), np.int16))
    k = 0
    for p in range(P):
        for by in range(H // 32):
            for bx in range(W // 32):
                mvx, mvy = int(rng.integers(-64, 64)), int(rng.integers(-64, 64))
                x = min(max(bx * 32 + (mvx >> 2), 8), W - 32 - 8)
                y = min(max(by * 32 + (mvy >> 2), 8), H - 32 - 8)
                base = p_i16 + k * 3072
                mcs.append(HB.McJob(p_b + p * ysz + y * stride + x * PX, base, stride, 64, 32, 32, mvx & 3, mvy & 3, 0))
                for pl in range(2):
                    mcs.append(HB.McJob(p_cb + (p * 2 + pl) * csz + (y // 2) * (stride // 2) + (x // 2) * PX, base + 2048 + pl * 512,
                                        stride // 2, 32, 16, 16, mvx & 7, mvy & 7, 1))
                k += 1
    p_mcs = d.up_jobs(mcs)
    timed("qpel/epel MC to 14 bit", lambda: lib.mi355_hevc_mc_batch_dev(C.c_void_p(p_mcs), n_mc, BD, None), n_mc, (2048 + 2 * 512) / 3 * 2)

    # ---- put_unweighted_pred: the 14-bit intermediates of the PUs -> samples of the reconstruction surface -----------
    preds = []
    k = 0
    for p in range(P):
        for by in range(H // 32):
            for bx in range(W // 32):
                base = p_i16 + k * 3072
                preds.append(HB.PredJob(p_a + p * ysz + by * 32 * stride + bx * 32 * PX, base, 0, stride, 64, 32, 32, 0, 0, 0, 0, 0, 0))
                for pl in range(2):
                    preds.append(HB.PredJob(p_ca + (p * 2 + pl) * csz + by * 16 * (stride // 2) + bx * 16 * PX, base + 2048 + pl * 512, 0,
                                            stride // 2, 32, 16, 16, 0, 0, 0, 0, 0, 0))
                k += 1
    p_preds = d.up_jobs(preds)
    timed("put_unweighted_pred", lambda: lib.mi355_hevc_pred_batch_dev(C.c_void_p(p_preds), len(preds), BD, None), len(preds), (2048 + 2 * 512) / 3 * 2)

    # ---- the same PUs through the fused entry point: MC and prediction in one launch, the 14-bit intermediate stays in LDS
    fused = []
    for k in range(0, len(mcs), 3):
        for q in range(3):
            m, pr = mcs[k + q], preds[k + q]
            fused.append(HB.McPredJob(m.src, 0, pr.dst, m.src_stride, 0, pr.dst_stride, m.width, m.height, m.chroma, 0, m.mx, m.my, 0, 0, 0))
    p_fused = d.up_jobs(fused)
    timed("MC + put_unweighted_pred fused", lambda: lib.mi355_hevc_mcpred_batch_dev(C.c_void_p(p_fused), len(fused), BD, None), len(fused), (2048 + 2 * 512) / 3 * 2)

    # ---- deblocking: every 8-sample luma edge segment of the 8x8 grid, vertical pass then horizontal pass ------
    def edges(horizontal):
        out = []
        for p in range(P):
            for gy in range(H // 8):
                for gx in range(W // 8):
                    if (gy if horizontal else gx) == 0:
                        continue
                    j = HB.LfJob(p_a + p * ysz + gy * 8 * stride + gx * 8 * PX, stride, int(rng.integers(20, 60)))
                    j.tc[0], j.tc[1] = int(rng.integers(1, 12)), int(rng.integers(1, 12))
                    j.horizontal_edge = horizontal
                    out.append(j)
        return out
    for horizontal, name in ((0, "deblock luma, vertical edges"), (1, "deblock luma, horizontal edges")):
        js = edges(horizontal)
        p_js = d.up_jobs(js)
        timed(name, lambda p_js=p_js, n=len(js): lib.mi355_hevc_deblock_batch_dev(C.c_void_p(p_js), n, BD, None), len(js), 8 * 8 * PX * 2)

    # chroma edges lie on the 8x8 chroma grid and are filtered only for bS 2 (10 % of them)
    def chroma_edges(horizontal):
        out = []
        for p in range(P):
            for pl in range(2):
                for gy in range(H // 16):
                    for gx in range(W // 16):
                        if (gy if horizontal else gx) == 0:
                            continue
                        for half in range(1):
                            if rng.random() >= 0.1:
                                continue
                            j = HB.LfJob(p_ca + (p * 2 + pl) * csz + gy * 8 * (stride // 2) + gx * 8 * PX, stride // 2, 0)
                            j.tc[0], j.tc[1] = int(rng.integers(1, 12)), int(rng.integers(1, 12))
                            j.horizontal_edge = horizontal
                            j.chroma = 1
                            out.append(j)
        return out
    for horizontal, name in ((0, "deblock chroma, vertical edges (bS 2)"), (1, "deblock chroma, horizontal edges (bS 2)")):
        js = chroma_edges(horizontal)
        p_js = d.up_jobs(js)
        timed(name, lambda p_js=p_js, n=len(js): lib.mi355_hevc_deblock_batch_dev(C.c_void_p(p_js), n, BD, None), len(js), 8 * 4 * PX * 2)

    # ---- SAO on every luma CTB (edge class, no picture-border special cases) + both chroma CTBs -------------------
    sao = []
    for p in range(P):
        for cy in range(1, H // 64 - 1):
            for cx in range(1, W // 64 - 1):
                j = HB.SaoJob(p_b + p * ysz + cy * 64 * stride + cx * 64 * PX, p_a + p * ysz + cy * 64 * stride + cx * 64 * PX, stride, 64, 64)
                for i in range(1, 5):
                    j.offset_val[i] = int(rng.integers(-28, 28))
                j.cls, j.edge, j.c_idx, j.eo_class = 0, int(rng.random() < 0.67), 0, int(rng.integers(0, 4))
                j.band_position = int(rng.integers(0, 32))
                sao.append(j)
    p_sao = d.up_jobs(sao)
    timed("SAO luma CTB (class 0 region)", lambda: lib.mi355_hevc_sao_batch_dev(C.c_void_p(p_sao), len(sao), BD, None), len(sao), 54 * 58 * PX * 2)

    csao = []
    cs = stride // 2
    for p in range(P):
        for pl in range(2):
            for cy in range(1, H // 64 - 1):
                for cx in range(1, W // 64 - 1):
                    o = (p * 2 + pl) * csz + cy * 32 * cs + cx * 32 * PX
                    j = HB.SaoJob(p_cb + o, p_ca + o, cs, 32, 32)
                    for i in range(1, 5):
                        j.offset_val[i] = int(rng.integers(-28, 28))
                    j.cls, j.edge, j.c_idx, j.eo_class = 0, int(rng.random() < 0.67), 1 + pl, int(rng.integers(0, 4))
                    j.band_position = int(rng.integers(0, 32))
                    csao.append(j)
    p_csao = d.up_jobs(csao)
    timed("SAO chroma CTBs (class 0 region)", lambda: lib.mi355_hevc_sao_batch_dev(C.c_void_p(p_csao), len(csao), BD, None), len(csao), 26 * 28 * PX * 2)

    # ---- the chain: one launch per stage for the whole batch, summed --------------------------------------------
    separate = ("qpel/epel MC to 14 bit", "put_unweighted_pred")
    chain_ms = sum(r["ms_per_launch"] for r in results if r["stage"] not in separate)
    chain_separate_ms = sum(r["ms_per_launch"] for r in results if r["stage"] != "MC + put_unweighted_pred fused")
    ctbs = P * (W // 64) * ((H + 63) // 64)
    chain = {"chain": "config 3: residual + fused MC/pred + deblock (luma V/H, chroma V/H) + SAO (luma, chroma)", "pictures": P,
             "ms_per_batch_with_separate_mc_and_pred": chain_separate_ms,
             "ms_per_batch": chain_ms, "pictures_per_s": P / chain_ms * 1e3, "ctb_per_s": ctbs / chain_ms * 1e3,
             "mb_equivalents_per_s": 16 * ctbs / chain_ms * 1e3, "algorithmic_bytes_per_ctb": 73984,
             "frac_of_8TBps": ctbs * 73984 / chain_ms / 1e6 / 8000}

    # ---- CPU oracle on a sample of the transform stage ---------------------------------------------------------
    orc = providers.oracle().hevcdsp(BD)
    blk = coef[:256].copy()
    t = time.time()
    for i in range(256):
        orc.idct[3](C.cast(blk[i].ctypes.data, C.POINTER(C.c_int16)), 12 if sparse[i] else 32)
    cpu = 256 / (time.time() - t)
    for r in results:
        print(json.dumps(r))
    print(json.dumps(chain))
    print(json.dumps({"cpu_oracle_idct32_per_s_1core": cpu}))
    d.free()


if __name__ == "__main__":
    main()
