#!/bin/bash
# all dependency levels of a launch set in ONE launch (mi355_hevc_recon_levels_dev) against a launch per level: device tests, then the HEVC bridge's rates both ways
#   gpurun -- 'bash tools/gpu_r06g.sh <tag>'
tag=${1:-r06g}; out=gpurun_out/$tag; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests/test_hevc_batch_gpu.py tests/test_hevc_bridge_gpu.py -x -q -m gpu > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.txt
fi
exe=oracle/_ref/hevc_bridge_gpu
for rep in 1 2; do
for name in i_ctb64 pb_ctb64_depth0 pb_480p_ctb64 pb_1080p_few_intra pb_1080p_ctb64; do
  src=tests/golden/hevc_synth_$name.samples
  for form in one_launch level_launches; do
    e="MI355_HEVC_BRIDGE_MIN_PIXELS=0"; [ $form = one_launch ] && e="$e MI355_HEVC_BRIDGE_ONE_LAUNCH=1"
    echo "$name $form x1: $(env $e timeout 600 $exe $src - 20 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["outputs_identical"], d["pictures_output"], d["reconstruction_launches"], d["dependency_levels"], d["pictures_per_s"])')"
  done
done
done | tee $out/bridge_forms.txt
for name in pb_1080p_few_intra; do
  src=tests/golden/hevc_synth_$name.samples
  for nthr in 4 16; do
  for form in one_launch level_launches c_decoder; do
    e="X=1"; [ $form = one_launch ] && e="MI355_HEVC_BRIDGE_ONE_LAUNCH=1"; [ $form = c_decoder ] && e="MI355_HEVC_RECON_PLAIN=1 MI355_HEVC_LF_PLAIN=1"
    echo "$name $form x$nthr: $(env $e timeout 900 $exe $src - 4 $nthr | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["outputs_identical"], d["pictures_output"], d["reconstruction_launches"], d["pictures_per_launch_set"], d["pictures_per_s"])')"
  done
  done
done | tee -a $out/bridge_forms.txt
