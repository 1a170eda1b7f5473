#!/bin/bash
# Round 5 session X: the built library and build/variants/*.so, bench step as 2 and 1 pipelines, alternating; then mixed partitions
set -u
TAG=${1:-r05x}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python -m pytest tests/test_frame_gpu.py -m gpu -q -x -k "two_partition or mixed or run_kernel" 2>&1 | tail -2
cp libav_amd/libmi355dsp.so /tmp/orig.so
one() {  # label pipelines
  timeout 200 python bench.py --no-cpu-baseline --no-extra --pipelines $2 --steps 10 --warmup 2 > /tmp/b.json 2> /tmp/b.err || { echo "bench failed"; tail -3 /tmp/b.err; exit 1; }
  python3 - $1 $2 <<'PY' | tee -a $OUT/pipelines.txt
import json, sys
b = json.load(open("/tmp/b.json"))
print("%-8s pipelines %s  ms/step %.3f  %.1f M MB/s  frac %.4f  %s" % (sys.argv[1], sys.argv[2], b["ms_per_step"], b["value"] / 1e6, b["config"]["fused_fraction_of_hbm_roofline"], " ".join("%s %.3f" % kv for kv in b["pass_ms"].items())))
PY
}
for round in 1 2 3; do
  for p in 2 1; do
    one built $p
    for so in build/variants/*.so; do
      cp $so libav_amd/libmi355dsp.so; one $(basename $so .so) $p; cp /tmp/orig.so libav_amd/libmi355dsp.so
    done
  done
done
bash tools/gpu_r05p.sh $TAG 2 mixed
