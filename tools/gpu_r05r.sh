#!/bin/bash
# Round 5 session R: the intra levels in one launch (k_recon_intra_all) against a launch per level (MI355_INTRA_SINGLE=0), same library, same box:
# frame / session / stream tests on the device with the single launch, then pass times on P and I batches, both forms alternating
set -u
TAG=${1:-r05r}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_frame_gpu.py tests/test_stream_parity.py tests/test_bridge_gpu.py -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_gpu.txt | cut -c1-300; [ $rc -eq 0 ] || exit 1
for round in 1 2 3; do
  for form in 1 0; do
    MI355_INTRA_SINGLE=$form timeout 300 python tools/exp_workloads.py single=$form base f512 f64 intra512 intra64 2>&1 | grep -v "^$" | tee -a $OUT/pass_ms.txt; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
  done
done
