#!/bin/bash
# Developer experiment: loop-filter chunk size (MI355_DCH_LOG) variants from build/variants: parity of the frame tests, then
# pass times at F = 2048 / 512 / 64.
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
cp libav_amd/libmi355dsp.so /tmp/orig.so
for so in build/variants/*.so; do
  n=$(basename $so .so)
  cp $so libav_amd/libmi355dsp.so
  echo "$n tests: $(timeout 300 python -m pytest tests/test_frame_gpu.py -x -q -m gpu 2>&1 | tail -1)"
  for F in 2048 512 64; do
    timeout 200 python bench.py --no-cpu-baseline --no-extra --frames $F --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n F=$F', round(d['value']/1e6,1), {k: round(v,2) for k,v in d['pass_ms'].items()})"
  done
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
