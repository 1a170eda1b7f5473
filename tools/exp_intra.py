#!/usr/bin/env python3
"""Developer experiment (GPU box): the intra pass as one launch per level against the one-launch form (MI355_INTRA_PERSISTENT=0 / 1):
all-intra pictures and the headline P pictures.  usage: exp_intra.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libav_amd
import h264_frames as HF
import bench

lib = libav_amd.load(0)


class P:
    pass


prov = P()
prov.lib = lib
for name, res, at in (("mi355_h264_recon_inter_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                      ("mi355_h264_recon_intra_levels_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
                      ("mi355_h264_deblock_dev", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
                      ("mi355_event_create", C.c_void_p, []), ("mi355_event_record", C.c_int, [C.c_void_p, C.c_void_p]),
                      ("mi355_event_elapsed_ms", C.c_float, [C.c_void_p, C.c_void_p])):
    getattr(lib, name).restype = res
    getattr(lib, name).argtypes = at
mbw, mbh = 120, 68
for what, fs, F in (("all intra", HF.synth_frames_fast(2, mbw, mbh, seed=0x1264, lib=lib, intra_frac=1.0), 512),
                    ("all intra", HF.synth_frames_fast(2, mbw, mbh, seed=0x1264, lib=lib, intra_frac=1.0), 64),
                    ("P, 5 % intra", HF.synth_frames_fast(4, mbw, mbh, seed=0x264, lib=lib), 2048),
                    ("P, 30 % intra", HF.synth_frames_fast(4, mbw, mbh, seed=0x264, lib=lib, intra_frac=0.3), 1024)):
    dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=True)
    lw = bench.level_widths(fs)
    for mode in ("0", "1"):
        os.environ["MI355_INTRA_PERSISTENT"] = mode
        ev = [lib.mi355_event_create() for _ in range(2)]
        for rep in range(3):
            if rep == 1:
                lib.mi355_event_record(ev[0], None)
            assert lib.mi355_h264_recon_intra_levels_dev(dev.d_desc, F, fs.max_intra_level, lw, None) == 0
        lib.mi355_event_record(ev[1], None)
        ms = lib.mi355_event_elapsed_ms(ev[0], ev[1]) / 2
        print("%-14s F=%4d levels %3d: intra pass %-22s %.3f ms" % (what, F, fs.max_intra_level, "one launch per level" if mode == "0" else "one launch (workgroups)", ms), flush=True)
    dev.free()
