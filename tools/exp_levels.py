#!/usr/bin/env python3
"""The hop from level to level on the device: a chain of L levels of n transform units each (8x8 units adding to the same few blocks: the chain is real) as ONE launch
(mi355_hevc_recon_levels_dev) and as a launch per level (mi355_hevc_recon_level_dev), timed with events.  tools/exp_levels.py [L] [n]   (GPU box)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import providers
import hevc_batch as HB

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prov = providers.mi355()
lib = prov.lib
d = HB.Dev(lib)
lib.mi355_event_create.restype = C.c_void_p
lib.mi355_event_elapsed_ms.restype = C.c_float
lib.mi355_hevc_recon_levels_dev.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
lib.mi355_hevc_recon_level_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
pic = np.zeros((64, 64 * n), np.uint8)
p_pic = d.up(pic)
coef = np.zeros((n, 64), np.int16)
coef[:, 0] = 64
p_coef = d.up(coef)
tus = [HB.TuJob(p_coef + 128 * (i % n), p_pic + 8 * (i % n), 64 * n, 3, 4, 0, 0) for i in range(L * n)]
p_tu = d.up_jobs(tus)
levels, wg = [], 0
for l in range(L):
    levels.append(HB.Level(wg, 0, 0, l * n, n, 0, 0, 0))
    wg += (n + 1) // 2
p_lv = d.up_jobs(levels)
e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        lib.mi355_sync(None)
        lib.mi355_event_record(C.c_void_p(e0), None)
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        lib.mi355_event_record(C.c_void_p(e1), None)
        assert lib.mi355_sync(None) == 0
        best = min(best, lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1)))
    return best, (t1 - t0) * 1e3


def one():
    assert lib.mi355_hevc_recon_levels_dev(p_lv, L, wg, None, p_tu, None, None, None, 8, None) == 0


def per_level():
    for l in range(L):
        lib.mi355_hevc_recon_level_dev(None, 0, p_tu + l * n * C.sizeof(HB.TuJob), n, None, None, None, 0, 8, None)


for name, fn in (("one launch", one), ("a launch per level", per_level)):
    ms, host = timed(fn)
    print("%-20s L=%d n=%d: %.3f ms on the device = %.2f us per level (host side of the calls %.1f ms)" % (name, L, n, ms, ms * 1e3 / L, host))
d.free()
