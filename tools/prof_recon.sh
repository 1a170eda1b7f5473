#!/bin/bash
# Developer tool: per-phase shader-clock profile of k_recon_inter (first 65536 waves), library built with -DMI355_PROF
# (tools/exp_variants.sh build prof "-DMI355_PROF").  GPU box: bash tools/prof_recon.sh [frames]
F=${1:-2048}
cd $(dirname $0)/..
F=$F python3 - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import h264_frames as HF
F = int(os.environ["F"])
lib = C.CDLL(os.path.abspath("build/variants/prof.so"))
lib.mi355_init.restype = C.c_int
assert lib.mi355_init(0) == 0
class P: pass
prov = P(); prov.lib = lib
fs = HF.synth_frames_fast(4, 120, 68, seed=0x264, lib=lib)
dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=os.environ.get("LAYOUT", "tiled") == "tiled")
lib.mi355_debug_rprof.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
d = C.c_void_p(dev.d_desc)
names = ["start: descriptor, indices", "record + vectors + coefficients (issue, wait, -> LDS)", "windows: addresses + issue", "windows: wait + -> LDS",
         "luma filter", "chroma filter", "residual", "store"]
for rep in range(2):
    lib.mi355_debug_rprof(out, 1)
    lib.mi355_h264_recon_inter_dev(d, F, 120, 68, None)
    lib.mi355_debug_rprof(out, 0)
    n = 65536 * 0.95
    tot = sum(out[i] for i in range(8))
    print("F=%d rep %d: %.0f clk per macroblock-wave" % (F, rep, tot / n))
    for i in range(8):
        print("  %-56s %8.0f  %5.1f%%" % (names[i], out[i] / n, 100.0 * out[i] / tot))
PY
