#!/bin/bash
# k_sws_ident1 (a generic context that does not scale, straight from the source bytes): device tests, then config 5's unscaled generic point with and without it
tag=${1:-r06p}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_sws_gpu.py tests/test_sws_binding_gpu.py -x -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.txt
for rep in 1 2; do
  echo "== ident1"; timeout 300 python tools/bench_sws.py --configs hd_generic,hd_special 2>&1 | grep -o '"name": "[a-z_]*"\|"frames_per_s": [0-9.]*\|"fraction_of_hbm_roofline": [0-9.]*' | paste - - -
  echo "== through the tile"; MI355_SWS_NO_IDENT1=1 timeout 300 python tools/bench_sws.py --configs hd_generic 2>&1 | grep -o '"name": "[a-z_]*"\|"frames_per_s": [0-9.]*\|"fraction_of_hbm_roofline": [0-9.]*' | paste - - -
done 2>&1 | tee $out/sws_ident1.txt
