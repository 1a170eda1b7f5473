#!/bin/bash
# Round 5 session D: run-kernel parity, pass times (built + build/variants), SQ counters of the inter kernel.  Usage (gpurun): bash tools/gpu_r05d.sh <tag>
set -u
TAG=${1:-r05d}
cd $GRAFT_REPO_ROOT
bash tools/gpu_r05a.sh $TAG "run_kernel or (layout_entry_points and True)" || exit 1
export TMPDIR=/tmp
bash tools/pmc_kernel.sh k_recon_inter_tiled python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --frames 512 --steps 2 --warmup 1 > gpurun_out/$TAG/pmc_inter.txt 2>&1; tail -22 gpurun_out/$TAG/pmc_inter.txt
