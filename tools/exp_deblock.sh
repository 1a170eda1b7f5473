#!/bin/bash
# Developer experiment: time k_deblock with a compile-time variant (-D<flag>) in a scratch copy.
set -e
FLAG=$1; F=${2:-1024}
rm -rf /tmp/exp && mkdir -p /tmp/exp && cp -r libav_amd include tests oracle /tmp/exp/
cd /tmp/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $FLAG -I include -o libav_amd/libmi355dsp.so libav_amd/csrc/*.hip
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
lib.mi355_event_create.restype = C.c_void_p; lib.mi355_event_elapsed_ms.restype = C.c_float
lib.mi355_h264_recon_inter_dev(C.c_void_p(dev.d_desc), F, 120, 68, None)
for rep in range(3):
    e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
    lib.mi355_event_record(C.c_void_p(e0), None)
    lib.mi355_h264_deblock_dev(C.c_void_p(dev.d_desc), F, 120, 68, None)
    lib.mi355_event_record(C.c_void_p(e1), None)
    lib.mi355_sync(None)
    print("F=%d deblock %.3f ms" % (F, lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1))))
PY
