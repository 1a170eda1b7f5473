cp libav_amd/libmi355dsp.so /tmp/built.so
for rep in 1 2; do
  echo "== built"; timeout 300 python tools/exp_detile.py 1024 2>&1 | tail -1
  for v in build/variants/conv_*.so; do cp $v libav_amd/libmi355dsp.so; echo "== $(basename $v .so)"; timeout 300 python tools/exp_detile.py 1024 2>&1 | tail -1; cp /tmp/built.so libav_amd/libmi355dsp.so; done
done
