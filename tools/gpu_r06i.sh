#!/bin/bash
tag=${1:-r06i}; out=gpurun_out/$tag; mkdir -p $out
for args in "1000 4" "1000 64" "200 2000" "3000 8"; do
  for dbg in 0 3 4; do echo "dbg=$dbg"; MI355_LEVELS_DBG=$dbg timeout 120 python tools/exp_levels.py $args; done
done 2>&1 | tee $out/levels.txt
