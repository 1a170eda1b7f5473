cp libav_amd/libmi355dsp.so /tmp/built.so
for rep in 1 2; do
  for F in 512 2048; do
    echo "== built F=$F"; timeout 600 python tools/wide_times.py $F 10 2>&1 | grep noise
    for v in build/variants/wide_base.so; do
      cp $v libav_amd/libmi355dsp.so; echo "== $(basename $v .so) F=$F"; timeout 600 python tools/wide_times.py $F 10 2>&1 | grep noise
      cp /tmp/built.so libav_amd/libmi355dsp.so
    done
  done
done
