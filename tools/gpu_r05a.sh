#!/bin/bash
# Round 5 session A: run-kernel parity on the device, then pass times (bench.py --no-extra) of the built library and of every library in
# build/variants on THIS box.  Stops at the first failure (a faulting kernel costs minutes of core dumps).  Usage (gpurun): bash tools/gpu_r05a.sh <tag> [pytest -k expr]
set -u
TAG=${1:-r05a}; KEXPR=${2:-run_kernel or layout_entry_points}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HSA_ENABLE_COREDUMP=0 2>/dev/null
ulimit -c 0
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_frame_gpu.py -m gpu -q -x -k "$KEXPR" > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/pytest_gpu.txt | cut -c1-400
[ $rc -eq 0 ] || exit 1
run_bench() {   # name, extra args
  timeout 200 python bench.py --no-cpu-baseline --no-extra --pipelines 1 --steps 10 --warmup 2 $2 > /tmp/b.json 2> /tmp/b.err || { echo "$1: bench failed"; tail -3 /tmp/b.err; return 1; }
  python3 - "$1" <<'PY' | tee -a $OUT/pass_ms.txt
import json, sys
b = json.load(open("/tmp/b.json"))
print("%-24s %s  ms/step %.3f  %.1f M MB/s  frac %.4f" % (sys.argv[1], " ".join("%s %.3f" % kv for kv in b["pass_ms"].items()), b["ms_per_step"], b["value"] / 1e6, b["config"]["fused_fraction_of_hbm_roofline"]))
PY
}
cp libav_amd/libmi355dsp.so /tmp/orig.so
run_bench built_f64 "--frames 64" || exit 1
run_bench built_f512 "--frames 512" || exit 1
run_bench built "" || exit 1
for so in build/variants/*.so; do
  [ -f "$so" ] || continue
  cp $so libav_amd/libmi355dsp.so
  run_bench $(basename $so .so) ""
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
run_bench built_again ""
