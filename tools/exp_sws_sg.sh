#!/bin/bash
# Developer experiment (GPU box): swscale timings and per-wave instruction counts of every library in build/variants
# (tools/exp_variants.sh build ...; tools/bench_sws.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CFG=${1:-hd_generic,uhd_to_hd}
cp libav_amd/libmi355dsp.so /tmp/orig.so
for so in build/variants/*.so; do
  cp $so libav_amd/libmi355dsp.so
  echo "== $(basename $so .so)"
  timeout 300 python tools/bench_sws.py --frames 32 --steps 10 --configs $CFG 2>&1 | grep -o '"name": "[a-z_0-9]*"\|"ms_per_launch": [0-9.]*' | paste - - | head -5
  ( cd /tmp && rm -rf /tmp/pmcs && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pmcs -- python $GRAFT_REPO_ROOT/tools/bench_sws.py --frames 32 --steps 2 --configs uhd_to_hd > /tmp/pmcs.log 2>&1 )
  python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("/tmp/pmcs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sws_generic" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
w = agg.get("SQ_WAVES", 0) or 1
print("   uhd_to_hd per wave:", " ".join("%s %.0f" % (k[9:], v / w) for k, v in sorted(agg.items()) if k != "SQ_WAVES"))
PY
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
