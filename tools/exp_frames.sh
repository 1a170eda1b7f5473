#!/bin/bash
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
cp libav_amd/libmi355dsp.so /tmp/orig.so
for so in build/variants/*.so; do
  n=$(basename $so .so)
  cp $so libav_amd/libmi355dsp.so
  for F in 3072 4096; do
    timeout 300 python bench.py --no-cpu-baseline --no-extra --frames $F --steps 6 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n F=$F', round(d['value']/1e6,1), {k: round(v,2) for k,v in d['pass_ms'].items()})"
  done
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
