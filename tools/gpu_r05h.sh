#!/bin/bash
# Round 5 session H: is the step time stable?  bench.py with 40 steps twice (built library), the shader clock sampled every 50 ms meanwhile; then run lengths.
set -u
TAG=${1:-r05h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/gpu_r05a.sh $TAG "run_kernel or (layout_entry_points and True)" || exit 1
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.05; done > $OUT/clocks.txt ) &
SMI=$!
for k in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-extra --steps 40 --warmup 2 > /tmp/b.json 2>/dev/null
  python3 -c "
import json; b=json.load(open('/tmp/b.json')); print('steps40 run $k', b['pass_ms'], 'ms/step %.3f' % b['ms_per_step'])" | tee -a $OUT/pass_ms.txt
done
kill $SMI 2>/dev/null
sort $OUT/clocks.txt | uniq -c | sort -rn | head -12
for r in 8 12 15 20 30; do
  MI355_RECON_RUN=$r timeout 200 python bench.py --no-cpu-baseline --no-extra --steps 10 --warmup 2 > /tmp/b.json 2>/dev/null
  python3 -c "
import json; b=json.load(open('/tmp/b.json')); print('run $r', b['pass_ms'], 'ms/step %.3f' % b['ms_per_step'])" | tee -a $OUT/pass_ms.txt
done
