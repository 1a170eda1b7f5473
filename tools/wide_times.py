"""Developer timing of the second H.264 kernel set (mi355_h264_decode_frames_wide_dev): pass times of config 2 as a High 10 / 8-bit 4:2:0 batch.
Usage (GPU box, repo root): python tools/wide_times.py [frames] [bit_depth]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import libav_amd
import h264_frames as HF

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bd = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = libav_amd.load(0)
class P: pass
prov = P(); prov.lib = lib
lib.mi355_event_create.restype = C.c_void_p; lib.mi355_event_elapsed_ms.restype = C.c_float
lib.mi355_event_record.argtypes = [C.c_void_p, C.c_void_p]; lib.mi355_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
fn = lib.mi355_h264_decode_frames_wide_dev
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
for content, kw in (("noise", {}), ("mixed", dict(partitions="mixed"))):
    fs = HF.synth_frames_fast(4, 120, 68, seed=0x264, lib=lib, **kw)
    lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
    dev = HF.DeviceFrames(prov, fs, replicate=F, bit_depth=bd)
    fn(dev.d_desc, F, 120, 68, fs.max_intra_level, lw, bd, 1, 7, None)
    out = []
    for p in (1, 2, 4):
        best = 1e9
        for _ in range(2):
            e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
            lib.mi355_event_record(e0, None)
            assert fn(dev.d_desc, F, 120, 68, fs.max_intra_level, lw, bd, 1, p, None) == 0
            lib.mi355_event_record(e1, None); lib.mi355_sync(None)
            best = min(best, lib.mi355_event_elapsed_ms(e0, e1))
        out.append(best)
    print("wide %d-bit %-6s F=%d inter %.2f intra %.2f deblock %.2f ms -> %.1f M MB/s" % (bd, content, F, out[0], out[1], out[2], F * 8160 / sum(out) / 1e3), flush=True)
    dev.free()
