#!/bin/bash
# Developer A/B of the second H.264 kernel set on ONE box: the library of the working tree against libav_amd/libmi355dsp_base.so (a build of the
# commit to compare with, made by hand), then instruction counters of the new library's kernels.
# Usage (via gpurun): bash tools/wide_ab.sh [frames]
F=${1:-512}
cd $GRAFT_REPO_ROOT
echo "== new"; python tools/wide_times.py $F 10
if [ -f libav_amd/libmi355dsp_base.so ]; then
  cp libav_amd/libmi355dsp.so /tmp/new.so; cp libav_amd/libmi355dsp_base.so libav_amd/libmi355dsp.so
  echo "== base"; python tools/wide_times.py $F 10
  cp /tmp/new.so libav_amd/libmi355dsp.so
fi
echo "== new again"; python tools/wide_times.py $F 10
bash tools/pmc_kernel.sh k_wide_inter,k_wide_deblock,k_wide_intra python $GRAFT_REPO_ROOT/tools/wide_times.py 128 10
