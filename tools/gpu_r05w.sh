#!/bin/bash
# Round 5 session W: the bench's step as 1 / 2 / 4 pipelines on streams, alternating, on the box of the moment
set -u
TAG=${1:-r05w}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for round in 1 2 3; do
  for p in 1 4 2; do
    timeout 200 python bench.py --no-cpu-baseline --no-extra --pipelines $p --steps 10 --warmup 2 > /tmp/b.json 2> /tmp/b.err || { echo "bench failed"; tail -3 /tmp/b.err; exit 1; }
    python3 - $p <<'PY' | tee -a $OUT/pipelines.txt
import json, sys
b = json.load(open("/tmp/b.json"))
print("pipelines %s  ms/step %.3f  %.1f M MB/s  frac %.4f  %s" % (sys.argv[1], b["ms_per_step"], b["value"] / 1e6, b["config"]["fused_fraction_of_hbm_roofline"], " ".join("%s %.3f" % kv for kv in b["pass_ms"].items())))
PY
  done
done
