#!/bin/bash
# Round 5 session P: pass times of the built library and of every library in build/variants, alternating, N rounds, on the workloads named.
# usage (gpurun): bash tools/gpu_r05p.sh <tag> <rounds> <workload ...>
set -u
TAG=${1:-r05p}; N=${2:-3}; shift 2
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
cp libav_amd/libmi355dsp.so /tmp/orig.so
for round in $(seq $N); do
  timeout 300 python tools/exp_workloads.py built "$@" 2>&1 | grep -v "^$" | tee -a $OUT/pass_ms.txt; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
  for so in build/variants/*.so; do
    [ -f "$so" ] || continue
    cp $so libav_amd/libmi355dsp.so
    timeout 300 python tools/exp_workloads.py $(basename $so .so) "$@" 2>&1 | grep -v "^$" | tee -a $OUT/pass_ms.txt; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
  done
  cp /tmp/orig.so libav_amd/libmi355dsp.so
done
