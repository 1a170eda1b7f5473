#!/bin/bash
# Developer experiment (GPU box): instructions per block-wave of k_hevc_mcpred_batch by case (tools/exp_mcpred_cases.py: two launches per case)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/mcp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/mcp -- python $GRAFT_REPO_ROOT/tools/exp_mcpred_cases.py > /tmp/mcp.log 2>&1 )
python3 - <<'PY'
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob("/tmp/mcp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_hevc_mcpred_batch" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["empty", "luma32 copy", "luma32 h", "luma32 v", "luma32 hv", "chroma16 copy", "chroma16 h", "chroma16 v", "chroma16 hv"]
for i, (k, v) in enumerate(sorted(rows.items())):
    if i % 2: continue
    w = v["SQ_WAVES"]
    print("%-14s per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f" % (names[i // 2], v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w, v["SQ_INSTS_VMEM_RD"] / w))
PY
