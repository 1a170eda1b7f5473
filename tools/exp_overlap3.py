#!/usr/bin/env python3
"""Developer experiment (GPU box, round 5: the single-layout kernels on tiled surfaces): the headline workload as P independent pipelines of F / P pictures, each on its own HIP
stream (reconstruction of one pipeline overlapping the loop filter of another).  usage: exp_overlap2.py [F] [steps]"""
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

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = libav_amd.load(0)


class P:
    pass


prov = P()
prov.lib = lib
mbw, mbh = 120, 68
fs = HF.synth_frames_fast(4, mbw, mbh, seed=0x264, lib=lib)
dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=True)
for name, res, at in (("mi355_h264_recon_inter_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                      ("mi355_h264_recon_intra_levels_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                      ("mi355_h264_deblock_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                      ("mi355_stream_create", C.c_void_p, []), ("mi355_sync", C.c_int, [C.c_void_p])):
    getattr(lib, name).restype = res
    getattr(lib, name).argtypes = at
lw = bench.level_widths(fs)
FR = C.sizeof(HF.Frame)


def run(npipe, order):
    streams = [lib.mi355_stream_create() for _ in range(npipe)]
    per = F // npipe

    def step():
        if order == "pass":       # pass by pass over the pipelines
            for s in range(npipe):
                assert lib.mi355_h264_recon_inter_layouts_dev(dev.d_desc + s * per * FR, per, mbw, mbh, 2, streams[s]) == 0
            for s in range(npipe):
                assert lib.mi355_h264_recon_intra_levels_dev(dev.d_desc + s * per * FR, per, fs.max_intra_level, lw, streams[s]) == 0
            for s in range(npipe):
                assert lib.mi355_h264_deblock_layouts_dev(dev.d_desc + s * per * FR, per, mbw, mbh, 2, streams[s]) == 0
        else:                     # pipeline by pipeline
            for s in range(npipe):
                d = dev.d_desc + s * per * FR
                assert lib.mi355_h264_recon_inter_layouts_dev(d, per, mbw, mbh, 2, streams[s]) == 0
                assert lib.mi355_h264_recon_intra_levels_dev(d, per, fs.max_intra_level, lw, streams[s]) == 0
                assert lib.mi355_h264_deblock_layouts_dev(d, per, mbw, mbh, 2, streams[s]) == 0
    for _ in range(2):
        step()
    for s in streams:
        lib.mi355_sync(s)
    t0 = time.perf_counter()
    for _ in range(STEPS):
        step()
    for s in streams:
        lib.mi355_sync(s)
    dt = (time.perf_counter() - t0) / STEPS
    print("pipelines %d order %-8s F=%d: %.2f ms per step, %.0f M MB/s" % (npipe, order, F, dt * 1e3, F * mbw * mbh / dt / 1e6), flush=True)


for npipe, order in ((1, "pipe"), (2, "pipe"), (2, "pass"), (4, "pipe"), (4, "pass"), (8, "pipe"), (3, "pipe")):
    if F % npipe == 0 or npipe == 3:
        run(npipe, order)
