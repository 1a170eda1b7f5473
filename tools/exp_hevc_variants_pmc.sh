#!/bin/bash
# Developer experiment (GPU box): per-wave instruction counts and the kernel's time for one kernel of the HEVC chain, for every library in build/variants
K=${1:-k_hevc_recon_ctbs}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp libav_amd/libmi355dsp.so /tmp/orig.so
for so in build/variants/*.so; do
  cp $so libav_amd/libmi355dsp.so
  rm -rf /tmp/hv; ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/hv -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > /tmp/hv.log 2>&1 )
  rm -rf /tmp/hs; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hs -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > /tmp/hs.log 2>&1 )
  python3 - "$K" "$(basename $so .so)" <<'PY'
import csv, glob, collections, sys
K, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(float)
for f in glob.glob("/tmp/hv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
w = agg.get("SQ_WAVES", 1) or 1
ms = 0
for f in glob.glob("/tmp/hs/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Name"]: ms = float(r["AverageNs"]) / 1e6
print("%-24s %.3f ms | per wave: " % (name, ms) + "  ".join("%s %.0f" % (k.replace("SQ_", ""), v / w) for k, v in sorted(agg.items()) if k != "SQ_WAVES"))
PY
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
