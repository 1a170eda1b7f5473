#!/bin/bash
# Round 5 session Q: frame tests on the device, pass times built / variants (tools/gpu_r05p.sh), HBM traffic of the built library's kernels
set -u
TAG=${1:-r05q}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_frame_gpu.py -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_gpu.txt | cut -c1-300; [ $rc -eq 0 ] || exit 1
bash tools/gpu_r05p.sh $TAG 3 base mixed
bash tools/gpu_traffic.sh ${TAG}_traffic --no-extra --steps 1 --warmup 0 2>&1 | tail -3
cat gpurun_out/${TAG}_traffic/traffic.json
