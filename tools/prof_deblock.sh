#!/bin/bash
# Developer tool: per-phase shader-clock profile of k_deblock (block 0 only), built with -DMI355_PROF into
# a scratch copy of the library.  Run on the GPU box from the repo root: bash tools/prof_deblock.sh [frames]
set -e
F=${1:-512}
rm -rf /tmp/prof && mkdir -p /tmp/prof && cp -r libav_amd include tests oracle /tmp/prof/
cd /tmp/prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMI355_PROF -I include -o libav_amd/libmi355dsp.so libav_amd/csrc/*.hip
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
lib.mi355_debug_prof.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
lib.mi355_debug_prof(out, 1)
for rep in range(2):
    lib.mi355_h264_recon_inter_dev(C.c_void_p(dev.d_desc), F, 120, 68, None)
    lib.mi355_debug_prof(out, 1)
    tot = sum(out[i] for i in range(8, 13))
    print("k_recon_inter, first 64 blocks: %.0f clk per MB" % (tot / 64))
    for i, n in zip(range(8, 13), ["load_mb", "hl_motion (MC)", "residual_luma", "residual_chroma", "store_mb"]):
        print("  %-20s %8.0f  %5.1f%%" % (n, out[i] / 64, 100.0 * out[i] / tot))
lib.mi355_h264_recon_intra_dev(C.c_void_p(dev.d_desc), F, fs.max_intra_level, fs.max_level_width, None)
for rep in range(2):
    lib.mi355_h264_deblock_dev(C.c_void_p(dev.d_desc), F, 120, 68, None)
    lib.mi355_debug_prof(out, 1)
    steps = 17 * 126 * min(F, 64)
    names = ["B1: rows from the group above", "C: bS", "D0: vertical edges + rows->LDS", "top rows->LDS", "D1: horizontal edges", "E: stores+carry", "-", "loop top (wait prefetch)"]
    tot = sum(out[i] for i in range(8)) + sum(out[i] for i in (13, 14, 15))
    for i, n in ((13, "A0: chunk flush / commit / issue"), (14, "A1: prefetch issue"), (15, "B0: records -> LDS, wave sync")):
        print("  %-34s %8.0f /step" % (n, out[i] / steps))
    print("F=%d rep %d: total %.0f clk/step (100 MHz ticks? see below)" % (F, rep, tot / steps))
    for i in (7, 0, 1, 2, 3, 4, 5):
        print("  %-34s %8.0f /step  %5.1f%%" % (names[i], out[i] / steps, 100.0 * out[i] / tot))
PY
