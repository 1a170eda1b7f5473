#!/bin/bash
# PMC passes over the bench (small batch): per-kernel instruction mix and wait breakdown.
# Usage: bash tools/gpu_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k in agg:
            print(k, {c: (round(v), n[(k, c)]) for c, v in agg[k].items()})
PY
