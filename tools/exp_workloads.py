#!/usr/bin/env python3
"""Developer experiment (GPU box, round 5): pass times of the three passes on a few workloads of the config-2 family with the library that is in
libav_amd/ right now (the session script swaps build/variants/*.so in).  usage: exp_workloads.py <label> [workload ...]   workloads: base mixed f512 f64 intra512 intra64"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libav_amd
import h264_frames as HF
import bench

label = sys.argv[1] if len(sys.argv) > 1 else "built"
which = sys.argv[2:] or ["base", "mixed"]
lib = libav_amd.load(0)


class P:
    pass


prov = P()
prov.lib = lib
mbw, mbh = 120, 68
for name, res, at in (("mi355_h264_recon_inter_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                      ("mi355_h264_recon_intra_all_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                      ("mi355_h264_deblock_layouts_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                      ("mi355_event_create", C.c_void_p, []), ("mi355_event_record", C.c_int, [C.c_void_p, C.c_void_p]),
                      ("mi355_event_elapsed_ms", C.c_float, [C.c_void_p, C.c_void_p]), ("mi355_sync", C.c_int, [C.c_void_p])):
    getattr(lib, name).restype = res
    getattr(lib, name).argtypes = at


def measure(name, fs, F):
    dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=True)
    lw = bench.level_widths(fs)
    try:
        def passes(ev=None):
            if ev: lib.mi355_event_record(ev[0], None)
            assert lib.mi355_h264_recon_inter_layouts_dev(dev.d_desc, F, mbw, mbh, 2, None) == 0
            if ev: lib.mi355_event_record(ev[1], None)
            assert lib.mi355_h264_recon_intra_all_dev(dev.d_desc, F, mbw, mbh, fs.max_intra_level, lw, None) == 0
            if ev: lib.mi355_event_record(ev[2], None)
            assert lib.mi355_h264_deblock_layouts_dev(dev.d_desc, F, mbw, mbh, 2, None) == 0
            if ev: lib.mi355_event_record(ev[3], None)
        passes()
        passes()
        best = None
        for _ in range(4):
            ev = [lib.mi355_event_create() for _ in range(4)]
            passes(ev)
            lib.mi355_sync(None)
            t = [lib.mi355_event_elapsed_ms(ev[i], ev[i + 1]) for i in range(3)]
            if best is None or sum(t) < sum(best):
                best = t
        v = F * mbw * mbh / (sum(best) * 1e-3)
        print("%-14s %-8s inter %.3f intra %.3f deblock %.3f  step %.3f ms  frac %.4f" % (label, name, best[0], best[1], best[2], sum(best), v * bench.B_FUSED / bench.HBM_PEAK), flush=True)
    finally:
        dev.free()


base = HF.synth_frames_fast(4, mbw, mbh, seed=0x264, lib=lib)
for w in which:
    if w == "base":
        measure("base", base, 2048)
    elif w == "mixed":
        measure("mixed", HF.synth_frames_fast(4, mbw, mbh, seed=0x2640, lib=lib, partitions="mixed"), 2048)
    elif w == "f512":
        measure("f512", base, 512)
    elif w == "f64":
        measure("f64", base, 64)
    elif w == "intra512":
        measure("intra512", HF.synth_frames_fast(2, mbw, mbh, seed=0x1264, lib=lib, intra_frac=1.0), 512)
    elif w == "intra64":
        measure("intra64", HF.synth_frames_fast(2, mbw, mbh, seed=0x1264, lib=lib, intra_frac=1.0), 64)
