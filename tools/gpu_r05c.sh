#!/bin/bash
# Round 5 session C: pass times of every library in build/variants (knock-outs) and of the built one on THIS box, then instruction / wait /
# memory-path counters of the inter kernel.  Usage (gpurun): bash tools/gpu_r05c.sh <tag> [pmc: 0|1]
set -u
TAG=${1:-r05c}; PMC=${2:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
cd $GRAFT_REPO_ROOT
run_bench() {   # name, extra args
  timeout 200 python bench.py --no-cpu-baseline --no-extra --pipelines 1 --steps 6 --warmup 2 $2 > /tmp/b.json 2> /tmp/b.err || { echo "$1: bench failed"; tail -3 /tmp/b.err; return 1; }
  python3 - "$1" <<'PY' | tee -a $OUT/pass_ms.txt
import json, sys
b = json.load(open("/tmp/b.json"))
print("%-24s %s  ms/step %.3f  %.1f M MB/s  frac %.4f" % (sys.argv[1], " ".join("%s %.3f" % kv for kv in b["pass_ms"].items()), b["ms_per_step"], b["value"] / 1e6, b["config"]["fused_fraction_of_hbm_roofline"]))
PY
}
cp libav_amd/libmi355dsp.so /tmp/orig.so
run_bench built "" || exit 1
for so in build/variants/*.so; do
  [ -f "$so" ] || continue
  cp $so libav_amd/libmi355dsp.so
  run_bench $(basename $so .so) ""
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
run_bench built_again ""
[ "$PMC" = 1 ] || exit 0
bash tools/pmc_kernel.sh k_recon_inter_tiled python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --frames 512 --steps 2 --warmup 1 > $OUT/pmc_inter.txt 2>&1; tail -24 $OUT/pmc_inter.txt
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "GRBM_GUI_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pm$i -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --frames 512 --steps 2 --warmup 1 > /tmp/pm$i.log 2>&1
  echo "mem pass $i rc=$?"
done
python3 - <<'PY' | tee $OUT/pmc_mem.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pm*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, a in agg.items():
    if k.startswith("k_"):
        print(k)
        for c, v in sorted(a.items()):
            print("   %-40s %.5g" % (c, v))
PY
