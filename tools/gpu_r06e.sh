#!/bin/bash
# Round 6: HEVC parity on the device + the config-3 chain's time and kernel stats
set -u
TAG=${1:-r06e}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
[ -n "${SKIP_TESTS:-}" ] || timeout 900 python -m pytest tests/test_hevc_batch_gpu.py tests/test_hevc_chain_gpu.py tests/test_hevc_gpu.py tests/test_hevc_filter_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_hevc.txt
short() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], 'ms', round(d['ms_per_step'],3), 'frac', round(d['fraction_of_hbm_roofline'],4))" "$1"; }
for round in 1 2; do
  timeout 300 python tools/hevc_chain.py 64 | tee -a $OUT/chain_fused.json | short fused8
done
rm -rf /tmp/prof_f; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > /tmp/prof_f.log 2>&1 )
cp $(find /tmp/prof_f -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_hevc_chain.csv
python3 - $OUT/kernel_stats_hevc_chain.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_hevc" in r["Name"] or "k_edge" in r["Name"]:
        print(r["Name"].replace("(anonymous namespace)::", "").split("(")[0][:60], round(float(r["AverageNs"]) / 1e6, 3), "ms")
PY
