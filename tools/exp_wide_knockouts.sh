#!/bin/bash
# the second kernel set's inter kernel with parts compiled out (build/variants/wide_*.so: -DMI355_WIDE_EXP_NOLUMA / _NOCHROMA / _NORES): pass times of the High 10 workload on one box
F=${1:-512}
cp libav_amd/libmi355dsp.so /tmp/built.so
for rep in 1 2; do
  echo "== built"; timeout 300 python tools/wide_times.py $F 10 2>&1 | grep wide
  for v in build/variants/wide_*.so; do
    cp $v libav_amd/libmi355dsp.so; echo "== $(basename $v .so)"; timeout 300 python tools/wide_times.py $F 10 2>&1 | grep wide
    cp /tmp/built.so libav_amd/libmi355dsp.so
  done
done
