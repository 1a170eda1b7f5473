#!/bin/bash
# Round 6 session A: the fused coding-tree-block kernel on hardware: parity, the chain's time both ways, kernel stats
set -u
TAG=${1:-r06a}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_hevc_batch_gpu.py tests/test_hevc_chain_gpu.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_hevc.txt
short() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], 'ms', round(d['ms_per_step'],3), 'frac', round(d['fraction_of_hbm_roofline'],4))" "$1"; }
for round in 1 2; do
  MI355_CHAIN_SPLIT=1 timeout 300 python tools/hevc_chain.py 64 | tee -a $OUT/chain_split.json | short split
  timeout 300 python tools/hevc_chain.py 64 | tee -a $OUT/chain_fused.json | short fused8
  MI355_CTB_WAVES=4 timeout 300 python tools/hevc_chain.py 64 | tee -a $OUT/chain_fused_w4.json | short fused4
  MI355_CHAIN_NO_PROMISE=1 timeout 300 python tools/hevc_chain.py 64 | tee -a $OUT/chain_fused_nopromise.json | short fused8_two_launches
  MI355_CTB_GENERAL_ONLY=1 timeout 300 python tools/hevc_chain.py 64 | tee -a $OUT/chain_general.json | short general_only
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > /tmp/prof_f.log 2>&1 )
cp $(find /tmp/prof_f -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_hevc_chain_fused.csv
( cd /tmp && MI355_CHAIN_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > /tmp/prof_s.log 2>&1 )
cp $(find /tmp/prof_s -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_hevc_chain_split.csv
head -12 $OUT/kernel_stats_hevc_chain_fused.csv
head -12 $OUT/kernel_stats_hevc_chain_split.csv
