#!/bin/bash
# Round 5 session L: the whole frame test file on the device, pass times (built + build/variants), counters of the intra kernel
set -u
TAG=${1:-r05l}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_gpu.py -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.txt | cut -c1-300
[ $rc -eq 0 ] || exit 1
bash tools/gpu_r05c.sh $TAG 0
bash tools/pmc_kernel.sh k_recon_intra python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --frames 512 --steps 2 --warmup 1 > $OUT/pmc_intra.txt 2>&1; tail -22 $OUT/pmc_intra.txt
