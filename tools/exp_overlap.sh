#!/bin/bash
# Developer experiment: two half-batches on two streams, reconstruction of one overlapping the
# deblocking of the other, against the sequential schedule.  Usage: bash tools/exp_overlap.sh F
set -e
F=${1:-1792}
F=$F python3 - <<'PY'
import os, sys, ctypes as C, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import libav_amd, h264_frames as HF
F = int(os.environ["F"])
lib = libav_amd.load(0)
class P: pass
prov = P(); prov.lib = lib
fs = HF.synth_frames_fast(4, 120, 68, seed=0x264, lib=lib)
dev = HF.DeviceFrames(prov, fs, replicate=F)
for n in ("mi355_event_create", "mi355_stream_create"): getattr(lib, n).restype = C.c_void_p
lib.mi355_event_elapsed_ms.restype = C.c_float
import ctypes
fsz = C.sizeof(HF.Frame)
def run(d, n, st):
    lib.mi355_h264_recon_inter_dev(C.c_void_p(d), n, 120, 68, st)
    lib.mi355_h264_recon_intra_dev(C.c_void_p(d), n, fs.max_intra_level, fs.max_level_width, st)
    lib.mi355_h264_deblock_dev(C.c_void_p(d), n, 120, 68, st)
def wall(fn, steps=6):
    fn(); lib.mi355_sync(None)
    t = time.perf_counter()
    for _ in range(steps): fn()
    lib.mi355_sync(None)
    return (time.perf_counter() - t) / steps * 1e3
seq = wall(lambda: run(dev.d_desc, F, None))
sa, sb = C.c_void_p(lib.mi355_stream_create()), C.c_void_p(lib.mi355_stream_create())
h = F // 2
def piped():
    # stream A: half 0, stream B: half 1, B's reconstruction starts when A's has finished
    ev = C.c_void_p(lib.mi355_event_create())
    lib.mi355_h264_recon_inter_dev(C.c_void_p(dev.d_desc), h, 120, 68, sa)
    lib.mi355_h264_recon_intra_dev(C.c_void_p(dev.d_desc), h, fs.max_intra_level, fs.max_level_width, sa)
    lib.mi355_event_record(ev, sa)
    lib.mi355_h264_deblock_dev(C.c_void_p(dev.d_desc), h, 120, 68, sa)
    lib.mi355_stream_wait_event(sb, ev)
    run(dev.d_desc + h * fsz, F - h, sb)
def piped_sync():
    piped(); lib.mi355_sync(sa); lib.mi355_sync(sb)
def wall2(steps=6):
    piped_sync()
    t = time.perf_counter()
    for _ in range(steps): piped()
    lib.mi355_sync(sa); lib.mi355_sync(sb)
    return (time.perf_counter() - t) / steps * 1e3
ov = wall2()
print("F=%d sequential %.2f ms/step (%.1f M MB/s)   two-stream pipeline %.2f ms/step (%.1f M MB/s)" % (F, seq, F*8160/seq/1e3, ov, F*8160/ov/1e3))
PY
