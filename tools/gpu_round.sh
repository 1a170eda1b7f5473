#!/bin/bash
# One GPU-box session: parity suite, default bench line, rocprofv3 kernel stats of the same command.
# Usage (from the repo root, via gpurun): bash tools/gpu_round.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py "$@" > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.txt 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec rm {} \;   # tens of MB; the stats are what is kept
head -8 $OUT/kernel_stats.csv
