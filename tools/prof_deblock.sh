#!/bin/bash
# Developer tool: per-phase shader-clock profile of k_deblock (first 64 blocks), from a library built with -DMI355_PROF
# (build/variants/prof.so, built here: tools/exp_variants.sh build prof "-DMI355_PROF").  GPU box: bash tools/prof_deblock.sh [frames]
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
dev = HF.DeviceFrames(prov, fs, replicate=F)
lib.mi355_debug_prof.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
d = C.c_void_p(dev.d_desc)
lib.mi355_h264_recon_inter_dev(d, F, 120, 68, None)
lib.mi355_h264_recon_intra_dev(d, F, fs.max_intra_level, fs.max_level_width, None)
lib.mi355_debug_prof(out, 1)
names = ["loop top: consume prefetch (vmcnt)", "prefetch issue (records, vectors)", "sync + rows from the group above", "bS + gathers",
         "params: combos -> LDS -> perms", "D0 + issue_chunk", "D1 + commit_chunk", "flush_chunk (+final)"]
for rep in range(2):
    lib.mi355_h264_deblock_dev(d, F, 120, 68, None)
    lib.mi355_debug_prof(out, 1)
    steps = 17 * 126 * min(F, 64)
    tot = sum(out[i] for i in range(8))
    print("F=%d rep %d: %.0f clk per 4-MB step (s_memtime ticks, 100 MHz constant clock?)" % (F, rep, tot / steps))
    for i in range(8):
        print("  %-40s %9.1f /step  %5.1f%%" % (names[i], out[i] / steps, 100.0 * out[i] / tot))
PY
