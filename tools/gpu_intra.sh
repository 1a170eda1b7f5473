#!/bin/bash
# GPU session for the HEVC intra_pred wrapper: parity tests, throughput, kernel stats.
set -u
TAG=${1:-r02i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hevc_intra_gpu.py tests/test_hevc_batch_gpu.py -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
tail -3 $OUT/pytest_gpu.txt
timeout 300 python tools/bench_hevc_intra.py > $OUT/hevc_intra_bench.jsonl 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/hevc_intra_bench.jsonl | cut -c1-260
tail -3 $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/tools/bench_hevc_intra.py --steps 2 > /dev/null 2> $OUT/stats.err; echo "stats rc=$?"
cat $(find $OUT/stats -name "*kernel_stats.csv" | head -1) | cut -c1-200
