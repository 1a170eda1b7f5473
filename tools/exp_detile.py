#!/usr/bin/env python3
"""mi355_h264_surface_convert_dev alone: F tiled 1080p pictures -> planes with line strides, device time by events.  tools/exp_detile.py [F]   (GPU box)"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import libav_amd
import h264_frames as HF
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = libav_amd.load(0)
class P: pass
prov = P(); prov.lib = lib
lib.mi355_event_create.restype = C.c_void_p; lib.mi355_event_elapsed_ms.restype = C.c_float
lib.mi355_event_record.argtypes = [C.c_void_p, C.c_void_p]; lib.mi355_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
fs = HF.synth_frames_fast(2, 120, 68, seed=0x264, lib=lib)
dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=True)
conv = bench.detile_jobs(lib, dev, fs, F)
best = 1e9
for _ in range(5):
    e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
    lib.mi355_event_record(e0, None)
    assert lib.mi355_h264_surface_convert_dev(conv, F, 120, 68, None) == 0
    lib.mi355_event_record(e1, None); lib.mi355_sync(None)
    best = min(best, lib.mi355_event_elapsed_ms(e0, e1))
gb = F * 120 * 68 * 384 * 2 / 1e9
print("surface_convert F=%d: %.3f ms, %.2f GB moved = %.2f TB/s" % (F, best, gb, gb / best))
dev.free()
