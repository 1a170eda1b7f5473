#!/usr/bin/env python3
"""Developer experiment (GPU box, round 5): the High 10 workload of bench.py's extra point (second kernel set, mi355_h264_decode_frames_wide_dev) as P pipelines on streams, the
reconstruction passes taking turns or not.  usage: exp_wide_pipelines.py [F]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libav_amd
import h264_frames as HF
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = libav_amd.load(0)


class P_:
    pass


prov = P_()
prov.lib = lib
mbw, mbh = 120, 68
base = HF.synth_frames_fast(4, mbw, mbh, seed=0x264, lib=lib)
dev = HF.DeviceFrames(prov, base, replicate=F, bit_depth=10)
lw = bench.level_widths(base)
fn = lib.mi355_h264_decode_frames_wide_dev
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.mi355_stream_create.restype = C.c_void_p
lib.mi355_event_create.restype = C.c_void_p
lib.mi355_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
lib.mi355_event_record.argtypes = [C.c_void_p, C.c_void_p]
fb = C.sizeof(dev.host_desc) // F


def measure(P, phased):
    streams = [C.c_void_p(lib.mi355_stream_create()) for _ in range(P)] if P > 1 else [None]
    turn = [C.c_void_p(lib.mi355_event_create()) for _ in range(P)]
    counts = [F // P + (1 if i < F % P else 0) for i in range(P)]
    started = [False]

    def step():
        for p, st in enumerate(streams):
            d = C.c_void_p(dev.d_desc + sum(counts[:p]) * fb)
            if phased and P > 1 and (p > 0 or started[0]):
                lib.mi355_stream_wait_event(st, turn[(p - 1) % P])
            assert fn(d, counts[p], mbw, mbh, base.max_intra_level, lw, 10, 1, 1, st) == 0
            if phased and P > 1:
                lib.mi355_event_record(turn[p], st)
                started[0] = True
            assert fn(d, counts[p], mbw, mbh, base.max_intra_level, lw, 10, 1, 6, st) == 0

    def sync():
        for st in streams:
            lib.mi355_sync(st)
    step(); sync()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    sync()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    v = F * mbw * mbh / (ms * 1e-3)
    print("F %d pipelines %d %s  %.2f ms  %.1f M MB/s  frac %.4f" % (F, P, "turns" if phased else "free", ms, v / 1e6, v * bench.B_FUSED_HIGH10 / bench.HBM_PEAK), flush=True)


for P, ph in ((1, False), (2, False), (2, True), (3, True), (3, False), (1, False)):
    measure(P, ph)
dev.free()
