#!/bin/bash
# Round 3: new GPU tests, the bench line with extras, kernel stats of the headline launch alone
set -u
TAG=${1:-r03e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hevc_chain_gpu.py tests/test_hevc_batch_gpu.py tests/test_sws_binding_gpu.py tests/test_frame_gpu.py tests/test_tier1_hevc_decoder_gpu.py -m gpu -q -x > $OUT/pytest_new.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_new.txt
timeout 1200 python bench.py > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench.txt"))
print(round(d["value"] / 1e6, 1), "M MB/s", d["pass_ms"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
for e in d.get("extra", []):
    print("  ", e.get("name"), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k in ("macroblocks_per_s", "fused_fraction_of_hbm_roofline", "fraction_of_hbm_roofline", "pictures_per_s", "frames_per_s", "ms_per_step", "error", "edge_emulated_windows_per_picture")}, (e.get("cpu_baseline") or {}).get("value"))
PY
