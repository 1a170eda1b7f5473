#!/bin/bash
# Round 4 checkpoint session: what the driver runs (smoke, pytest -m gpu, bench at N=1).  Usage: bash tools/gpu_r04_check.sh <tag> [bench args]
set -u
TAG=${1:-r04}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt
timeout 1200 python bench.py "$@" > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-3000 $OUT/bench.txt; wc -c $OUT/bench.txt
