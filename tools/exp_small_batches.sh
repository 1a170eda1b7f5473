#!/bin/bash
# Round 6 experiment (GPU box): 64 / 512 pictures per step as 1 / 2 / 3 / 4 shares through the pipelines object
cd $GRAFT_REPO_ROOT
for F in 64 512; do for P in 1 2 3 4; do
  python bench.py --frames $F --pipelines $P --no-extra --no-cpu-baseline --no-alone --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('F', $F, 'P', $P, 'ms', round(d['ms_per_step'],3), 'frac', round(d['config']['fused_fraction_of_hbm_roofline'],4), {k: round(v,3) for k,v in d['pass_ms'].items()})"
done; done
