#!/bin/bash
# the second kernel set's loop filter on unit-wide tiles: device tests of every format it serves, then pass times against build/variants/wide_base.so (the commit before) on one box
tag=${1:-r06l}; out=gpurun_out/$tag; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests/test_frame_gpu.py tests/test_synth_streams_gpu.py -x -q -m gpu -k "second_kernel or high10 or wide or 422 or synth or bridge or stream" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.txt
  for u in 2 3; do MI355_WIDE_UNIT=$u timeout 900 python -m pytest tests/test_frame_gpu.py -x -q -m gpu -k "second_kernel or 422" > $out/pytest_unit$u.txt 2>&1; echo "unit $u pytest rc=$?"; tail -1 $out/pytest_unit$u.txt; done
fi
cp libav_amd/libmi355dsp.so /tmp/built.so
for rep in 1 2; do
  for F in 512 2048; do
    echo "== built F=$F"; timeout 600 python tools/wide_times.py $F 10 2>&1 | grep wide
    for v in build/variants/wide_base.so; do
      [ -f $v ] || continue
      cp $v libav_amd/libmi355dsp.so; echo "== $(basename $v .so) F=$F"; timeout 600 python tools/wide_times.py $F 10 2>&1 | grep wide
      cp /tmp/built.so libav_amd/libmi355dsp.so
    done
  done
done 2>&1 | tee $out/wide_times.txt
