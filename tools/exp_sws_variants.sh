#!/bin/bash
# Developer experiment: k_sws_generic with parts compiled out (build/variants/sws_*.so from tools/exp_variants.sh build)
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
cp libav_amd/libmi355dsp.so /tmp/orig.so
for so in /tmp/orig.so build/variants/sws_*.so; do
  cp $so libav_amd/libmi355dsp.so
  echo "== $(basename $so)"
  timeout 200 python tools/bench_sws.py --steps 10 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ', d.get('config'), 'ms %.4f' % d['ms_per_launch'])"
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
