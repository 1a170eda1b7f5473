#!/bin/bash
# Developer experiment: time the three passes with a compile-time variant (-D...) in a scratch copy.
# Usage: bash tools/exp_deblock.sh "<hipcc flags>" F1 [F2 ...]
set -e
FLAG=$1; shift
rm -rf /tmp/exp && mkdir -p /tmp/exp && cp -r libav_amd include tests oracle /tmp/exp/
cd /tmp/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $FLAG -I include -o libav_amd/libmi355dsp.so libav_amd/csrc/*.hip
for F in "$@"; do
F=$F python3 - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import libav_amd, h264_frames as HF
F = int(os.environ["F"])
lib = libav_amd.load(0)
class P: pass
prov = P(); prov.lib = lib
fs = HF.synth_frames_fast(4, 120, 68, seed=0x264, lib=lib)
dev = HF.DeviceFrames(prov, fs, replicate=F)
lib.mi355_event_create.restype = C.c_void_p; lib.mi355_event_elapsed_ms.restype = C.c_float
d = C.c_void_p(dev.d_desc)
def timed(fn):
    e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
    lib.mi355_event_record(C.c_void_p(e0), None); fn(); lib.mi355_event_record(C.c_void_p(e1), None); lib.mi355_sync(None)
    return lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1))
for rep in range(3):
    ti = timed(lambda: lib.mi355_h264_recon_inter_dev(d, F, 120, 68, None))
    tx = timed(lambda: lib.mi355_h264_recon_intra_dev(d, F, fs.max_intra_level, fs.max_level_width, None))
    td = timed(lambda: lib.mi355_h264_deblock_dev(d, F, 120, 68, None))
nmb = F * 8160
print("F=%d inter %.3f intra %.3f deblock %.3f ms -> %.1f M MB/s (deblock alone %.1f M MB/s)" % (F, ti, tx, td, nmb / (ti + tx + td) / 1e3, nmb / td / 1e3))
PY
done
