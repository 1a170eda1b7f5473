#!/bin/bash
# Round 6 session C: counters of the fused coding-tree-block kernel and of SAO on the config-3 chain
set -u
TAG=${1:-r06c}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
bash tools/pmc_kernel.sh k_hevc_recon_ctbs,k_hevc_sao_ctbs,k_hevc_deblock_pictures python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 2>&1 | tee $OUT/pmc_hevc_chain.txt
SQ_MF="SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_BUSY_CYCLES"
( cd /tmp && timeout 600 rocprofv3 --pmc $SQ_MF --output-format csv -d /tmp/pmcm -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > /tmp/pmcm.log 2>&1 ); echo "mfma pass rc=$?"
tail -3 /tmp/pmcm.log
python3 - <<'PY' | tee -a $OUT/pmc_hevc_chain.txt
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("/tmp/pmcm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_hevc_recon_ctbs" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()): print(k, v, v / max(agg.get("SQ_WAVES", 1), 1))
PY
