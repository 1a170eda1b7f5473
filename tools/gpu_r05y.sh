#!/bin/bash
# Round 5 session Y: bench step as one pipeline, two, and two whose reconstruction launches take turns (--phased), alternating
set -u
TAG=${1:-r05y}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
one() {
  timeout 200 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 "$@" > /tmp/b.json 2> /tmp/b.err || { echo "bench failed"; tail -3 /tmp/b.err; exit 1; }
  python3 - "$*" <<'PY' | tee -a $OUT/pipelines.txt
import json, sys
b = json.load(open("/tmp/b.json"))
print("%-28s ms/step %.3f  %.1f M MB/s  frac %.4f  %s" % (sys.argv[1], b["ms_per_step"], b["value"] / 1e6, b["config"]["fused_fraction_of_hbm_roofline"], " ".join("%s %.3f" % kv for kv in b["pass_ms"].items())))
PY
}
# (the sessions r05y / r05y2 ran this script when --phased was still an option; since then turns are bench.py's default and --no-phased the option)
if [ "${2:-}" = sweep ]; then
  for round in 1 2; do
    for p in 2 3 4 6 8 16; do one --pipelines $p; done
  done
  exit 0
fi
for round in 1 2 3; do
  one --pipelines 1
  one --pipelines 2 --no-phased
  one --pipelines 2
  one --pipelines 3
done
