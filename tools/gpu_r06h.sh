#!/bin/bash
# where the one-launch form's time goes: the form with its fences / its back-off switched off (MI355_LEVELS_DBG: wrong pictures, timing only); every command under its own timeout
tag=${1:-r06h}; out=$PWD/gpurun_out/$tag; mkdir -p $out
exe=$PWD/oracle/_ref/hevc_bridge_gpu
for name in i_ctb64 pb_1080p_few_intra; do
  src=tests/golden/hevc_synth_$name.samples
  echo "$name level launches: $(MI355_HEVC_BRIDGE_MIN_PIXELS=0 MI355_HEVC_BRIDGE_LEVEL_LAUNCHES=1 timeout 120 $exe $src - 10 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["outputs_identical"], d["pictures_per_s"])')"
  for dbg in 0 1 2 3 4 7; do
    echo "$name dbg=$dbg: $(MI355_HEVC_BRIDGE_MIN_PIXELS=0 MI355_LEVELS_DBG=$dbg timeout 120 $exe $src - 10 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["outputs_identical"], d["pictures_per_s"])')"
  done
done 2>&1 | tee $out/dbg.txt
