#!/bin/bash
# Round 6: counters of the swscale kernels on config 5's points: tools/gpu_pmc_sws.sh <tag> [configs]
TAG=${1:-r06sws}; CFG=${2:-hd_generic,uhd_to_hd}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for c in $(echo $CFG | tr ',' ' '); do
  echo "=== $c"
  python tools/bench_sws.py --configs $c 2>&1 | tail -1 | cut -c1-300
  bash tools/pmc_kernel.sh k_sws python $GRAFT_REPO_ROOT/tools/bench_sws.py --configs $c --steps 5 2>&1 | grep -v "^pass"
done 2>&1 | tee gpurun_out/$TAG/pmc_sws.txt
