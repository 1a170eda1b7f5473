#!/bin/bash
# Round 6: instruction / wait counters of named kernels of the config-3 chain: tools/gpu_pmc_hevc.sh <tag> <kernel[,kernel]>
TAG=${1:-r06pmc}; K=${2:-k_hevc_sao_ctbs}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
bash tools/pmc_kernel.sh $K python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 2>&1 | grep -v "^pass" | tee gpurun_out/$TAG/pmc_$(echo $K | tr ',' '_').txt
