#!/bin/bash
# Developer experiment: per-wave instruction mix and wait counters of one kernel.
# Usage (via gpurun): bash tools/pmc_kernel.sh <kernel substring[,substring...]> <command...>
set -u
K=$1; shift
export TMPDIR=/tmp
OUT=/tmp/pmck; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1 ); echo "pass $i rc=$?"
done
python3 - "$K" <<'PY'
import csv, glob, collections, sys
for K in sys.argv[1].split(","):                      # several kernels: comma-separated substrings
    agg = collections.defaultdict(float)
    for f in glob.glob("/tmp/pmck/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if K in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
    w = agg.get("SQ_WAVES", 1) / 3 or 1
    print("==", K, "waves (per pass)", w)
    for k, v in sorted(agg.items()):
        print("%-28s total %.4g  per wave %.1f" % (k, v, v / (w * (3 if k == "SQ_WAVES" else 1))))
PY
