#!/bin/bash
# Round 4, session a: the single-launch loop filter on hardware — parity (frame / field / session suites + the hand-down stress
# test), then pass times of every library in build/variants for 2048 / 512 / 64 pictures, and the old forms beside them.
# Usage (repo root, via gpurun): bash tools/gpu_r04a.sh <tag>     (R04_QUICK=1: pass times of the built library and the variants only)
set -u
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "${R04_QUICK:-}" ] || timeout 900 python -m pytest tests/test_frame_gpu.py tests/test_field_gpu.py tests/test_session_gpu.py tests/test_bridge_gpu.py tests/test_synth_streams_gpu.py -m gpu -x -q > $OUT/pytest_frames.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_frames.txt
tail -6 $OUT/pytest_frames.txt
cp libav_amd/libmi355dsp.so /tmp/orig.so
times() {   # $1 = label
python3 - "$1" <<'PY' 2>&1 | tee -a $OUT/passes.txt
import os, sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import libav_amd, h264_frames as HF
lib = libav_amd.load(0)
class P: pass
prov = P(); prov.lib = lib
lib.mi355_event_create.restype = C.c_void_p; lib.mi355_event_elapsed_ms.restype = C.c_float
for content, kw in (("noise", {}), ("smooth", dict(refs="smooth", coef_b=4)), ("mixed", dict(partitions="mixed"))):
    fs = HF.synth_frames_fast(4, 120, 68, seed={"noise": 0x264, "smooth": 0x2264, "mixed": 0x2640}[content], lib=lib, **kw)
    lw = (C.c_int32 * max(1, fs.max_intra_level))(*fs.level_widths[:fs.max_intra_level])
    for F in ((2048, 512, 64) if content == "noise" else (2048,)):
        dev = HF.DeviceFrames(prov, fs, replicate=F, tiled=True)
        d = C.c_void_p(dev.d_desc)
        def timed(fn, reps=3):
            best = 1e9
            for _ in range(reps):
                e0, e1 = lib.mi355_event_create(), lib.mi355_event_create()
                lib.mi355_event_record(C.c_void_p(e0), None); fn(); lib.mi355_event_record(C.c_void_p(e1), None); lib.mi355_sync(None)
                best = min(best, lib.mi355_event_elapsed_ms(C.c_void_p(e0), C.c_void_p(e1)))
            return best
        ti = timed(lambda: lib.mi355_h264_recon_inter_layouts_dev(d, F, 120, 68, 2, None) if hasattr(lib, 'mi355_h264_recon_inter_layouts_dev') else lib.mi355_h264_recon_inter_dev(d, F, 120, 68, None))
        tx = timed(lambda: lib.mi355_h264_recon_intra_levels_dev(d, F, fs.max_intra_level, lw, None))
        td = timed(lambda: lib.mi355_h264_deblock_layouts_dev(d, F, 120, 68, 2, None) if hasattr(lib, 'mi355_h264_deblock_layouts_dev') else lib.mi355_h264_deblock_dev(d, F, 120, 68, None), 5)
        nmb = F * 8160
        print("%-22s %-6s F=%-5d inter %.3f intra %.3f deblock %.3f ms -> %.1f M MB/s" % (sys.argv[1], content, F, ti, tx, td, nmb / (ti + tx + td) / 1e3), flush=True)
        dev.free()
PY
}
[ -n "${R04_QUICK:-}" ] || MI355_DEBLOCK_FORM=-1 times old_forms
times default
for so in build/variants/*.so; do
  [ -e "$so" ] || continue
  cp $so libav_amd/libmi355dsp.so
  times $(basename $so .so)
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
