#!/bin/bash
# round 3, last GPU call: the fused intra prediction + residual launch — parity on the GPU, and the HEVC bridge with and without it (same box)
mkdir -p gpurun_out
( timeout 40 python -m pytest tests/test_hevc_intra_gpu.py tests/test_hevc_bridge_gpu.py -q -x 2>&1 | tail -4 ) > gpurun_out/r03y_pytest_gpu_hevc_fused.txt 2>&1
{
for s in pb_1080p_few_intra pb_1080p_ctb64 i_ctb64; do
  for mode in fused split; do
    if [ $mode = split ]; then export MI355_HEVC_BRIDGE_SPLIT_INTRA=1; else unset MI355_HEVC_BRIDGE_SPLIT_INTRA; fi
    echo -n "$s $mode: "; timeout 20 oracle/_ref/hevc_bridge_gpu tests/golden/hevc_synth_$s.samples - 10 2>&1 | tail -1
  done
done
} > gpurun_out/r03y_hevc_bridge_fused_vs_split.txt 2>&1
tail -3 gpurun_out/r03y_pytest_gpu_hevc_fused.txt; cat gpurun_out/r03y_hevc_bridge_fused_vs_split.txt | cut -c1-330
