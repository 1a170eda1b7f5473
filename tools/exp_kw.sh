#!/bin/bash
# Developer experiment: the small-batch deblocking form with 3 (the built library), 4, 6, 8 waves per workgroup
# (build/variants/kwN.so from tools/exp_variants.sh build): pass times at F = 64 / 256 / 512.
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
cp libav_amd/libmi355dsp.so /tmp/orig.so
for so in /tmp/orig.so build/variants/kw*.so; do
  cp $so libav_amd/libmi355dsp.so
  for F in 64 256 512; do
    timeout 200 python bench.py --no-cpu-baseline --no-extra --frames $F --steps 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(basename $so) F=$F', round(d['value']/1e6,1), {k: round(v,2) for k,v in d['pass_ms'].items()})"
  done
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
