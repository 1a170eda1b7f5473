#!/bin/bash
# Round 5 session N: the inter pass as two launches (runs + the macroblocks they leave) against the commit before (build/variants/prev.so) on one box:
# frame tests on the device, then pass times on the headline and the mixed-partition workloads, each library twice.
set -u
TAG=${1:-r05n}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_frame_gpu.py -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_gpu.txt | cut -c1-300; [ $rc -eq 0 ] || exit 1

cp libav_amd/libmi355dsp.so /tmp/orig.so
for round in 1 2; do
  timeout 300 python tools/exp_workloads.py built base mixed f512 f64 2>&1 | grep -v "^$" | tee -a $OUT/pass_ms.txt; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
  for so in build/variants/*.so; do
    [ -f "$so" ] || continue
    cp $so libav_amd/libmi355dsp.so
    timeout 300 python tools/exp_workloads.py $(basename $so .so) base mixed f512 f64 2>&1 | grep -v "^$" | tee -a $OUT/pass_ms.txt; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
  done
  cp /tmp/orig.so libav_amd/libmi355dsp.so
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/tools/exp_workloads.py built base mixed > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err )
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/prof
head -8 $OUT/kernel_stats.csv | cut -c1-200
