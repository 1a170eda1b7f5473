#!/bin/bash
# Developer experiment: how much of k_recon_inter's time is the scattered reference fetch?  Same kernels, same
# number of pictures, workloads that differ only in the spread of vectors / number of references.
set -e
F=${1:-1024}
F=$F python3 - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import libav_amd, h264_frames as HF
F = int(os.environ["F"])
lib = libav_amd.load(0)
class P: pass
prov = P(); prov.lib = lib
lib.mi355_event_create.restype = C.c_void_p; lib.mi355_event_elapsed_ms.restype = C.c_float
def timed(fn):
    e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
    lib.mi355_event_record(C.c_void_p(e0), None); fn(); lib.mi355_event_record(C.c_void_p(e1), None); lib.mi355_sync(None)
    return lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1))
for name, kw in (("config 2: 4 refs, +-16 samples", dict(nrefs=4, mv_range=64)), ("1 ref, +-16 samples", dict(nrefs=1, mv_range=64)),
                 ("4 refs, +-0.25 sample", dict(nrefs=4, mv_range=1)), ("1 ref, +-0.25 sample", dict(nrefs=1, mv_range=1))):
    fs = HF.synth_frames_fast(4, 120, 68, seed=0x264, lib=lib, **kw)
    dev = HF.DeviceFrames(prov, fs, replicate=F)
    d = C.c_void_p(dev.d_desc)
    for rep in range(3):
        ti = timed(lambda: lib.mi355_h264_recon_inter_dev(d, F, 120, 68, None))
    print("%-32s k_recon_inter %.3f ms for %d pictures" % (name, ti, F), flush=True)
    dev.free()
PY
