#!/bin/bash
# Developer experiment: per-wave SQ_INSTS_* of the frame kernels for every library in build/variants (one rocprofv3 --pmc
# pass each at F = 128, no timing).  GPU box: tools/exp_valu_only.sh
ROOT=$GRAFT_REPO_ROOT; cd $ROOT
export TMPDIR=/tmp
cp libav_amd/libmi355dsp.so /tmp/orig.so
for so in build/variants/*.so; do
  name=$(basename $so .so)
  cp $so libav_amd/libmi355dsp.so
  ( cd /tmp && rm -rf /tmp/pmcv && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmcv -- python $ROOT/bench.py --no-cpu-baseline --no-extra --frames 128 --steps 1 --warmup 0 > /tmp/pmcv.log 2>&1 )
  python3 - "$name" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmcv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
s = "%-12s" % sys.argv[1]
for k in ("k_recon_inter", "k_recon_intra"):
    a = agg.get(k)
    if a:
        w = a["SQ_WAVES"] or 1
        s += " | %s waves %.0f VALU %.1f SALU %.1f LDS %.1f" % (k[2:], w, a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w)
print(s, flush=True)
PY
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
