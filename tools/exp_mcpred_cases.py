#!/usr/bin/env python3
"""Developer experiment: k_hevc_mcpred_batch on ONE kind of block per launch (10-bit, uni-predicted), so that a rocprofv3 --pmc run of
this script gives instructions per block-wave by case: luma 32x32 / chroma 16x16, copy / horizontal / vertical / 2-D.
  rocprofv3 --kernel-trace --stats ... -- python tools/exp_mcpred_cases.py     (one launch per case, in the order printed)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hevc_batch as HB  # noqa: E402
import libav_amd  # noqa: E402

lib = libav_amd.load()
lib.mi355_event_create.restype = C.c_void_p
lib.mi355_event_elapsed_ms.restype = C.c_float
W, H, BD, PX, P = 3840, 2160, 10, 2, 8
d = HB.Dev(lib)
rng = np.random.default_rng(1)
ref = d.up(rng.integers(0, 1 << BD, (P, H + 16, W + 16), dtype=np.uint16))
dst = d.up(np.zeros((P, H, W), np.uint16))
rs, ds = (W + 16) * PX, W * PX
cases = [("empty", 0, 0, 1, 1), ("luma32 copy", 32, 0, 0, 0), ("luma32 h", 32, 0, 2, 0), ("luma32 v", 32, 0, 0, 2), ("luma32 hv", 32, 0, 1, 3),
         ("chroma16 copy", 16, 1, 0, 0), ("chroma16 h", 16, 1, 3, 0), ("chroma16 v", 16, 1, 0, 5), ("chroma16 hv", 16, 1, 3, 5)]
for name, size, chroma, mx, my in cases:
    jobs = []
    for p in range(P):
        for y in range(0, H - max(size, 32) + 1, max(size, 32)):
            for x in range(0, W - max(size, 32) + 1, max(size, 32)):
                j = HB.McPredJob()
                j.src0 = ref + (p * (H + 16) + y + 8) * rs + (x + 8) * PX
                j.dst = dst + (p * H + y) * ds + x * PX
                j.src0_stride, j.dst_stride, j.width, j.height, j.chroma, j.kind, j.mx0, j.my0 = rs, ds, size, size, chroma, 0, mx, my
                jobs.append(j)
        if len(jobs) > 40000:
            break
    n = len(jobs)
    dj = d.up_jobs(jobs)
    e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
    lib.mi355_hevc_mcpred_batch_dev(C.c_void_p(dj), n, BD, None)
    lib.mi355_sync(None)
    lib.mi355_event_record(C.c_void_p(e0), None)
    lib.mi355_hevc_mcpred_batch_dev(C.c_void_p(dj), n, BD, None)
    lib.mi355_event_record(C.c_void_p(e1), None)
    lib.mi355_sync(None)
    ms = lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1))
    print("%-14s %6d blocks  %.3f ms  %.2f ns per block" % (name, n, ms, ms * 1e6 / n), flush=True)
