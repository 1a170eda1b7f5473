#!/bin/bash
# Developer experiment: per-wave instruction counts (one rocprofv3 --pmc pass) and pass times of prebuilt library
# variants.  Build the variants HERE (no GPU needed), run them on the GPU box:
#   tools/exp_variants.sh build NAME "<-D flags>" [NAME "<flags>" ...]    -> build/variants/NAME.so
#   tools/exp_variants.sh run <tag> [frames]                                -> gpurun_out/<tag>/variants.txt
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
if [ "$1" = build ]; then
  shift
  mkdir -p $ROOT/build/variants
  while [ $# -gt 0 ]; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $2 -I $ROOT/include -o $ROOT/build/variants/$1.so $ROOT/libav_amd/csrc/*.hip &
    shift 2
  done
  wait
  ls -la $ROOT/build/variants
  exit 0
fi
TAG=$2; F=${3:-256}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cp $ROOT/libav_amd/libmi355dsp.so /tmp/orig.so
for so in $ROOT/build/variants/*.so; do
  name=$(basename $so .so)
  cp $so $ROOT/libav_amd/libmi355dsp.so
  ( cd /tmp && rm -rf /tmp/pmcv && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmcv -- python $ROOT/bench.py --no-cpu-baseline --frames $F --steps 1 --warmup 0 > /tmp/pmcv.log 2>&1 )
  timeout 200 python $ROOT/bench.py --no-cpu-baseline --frames 2048 --steps 5 --warmup 1 > /tmp/b.json 2>/dev/null || true
  python3 - "$name" <<'PY' | tee -a $OUT/variants.txt
import csv, glob, collections, sys, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmcv/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
try:
    b = json.load(open("/tmp/b.json")); t = " ".join("%s %.2f" % (k, v) for k, v in b["pass_ms"].items()) + " ms  %.0f M MB/s" % (b["value"] / 1e6)
except Exception as e:
    t = "bench failed"
s = "%-28s %s" % (sys.argv[1], t)
for k in ("k_recon_inter", "k_deblock", "k_recon_intra"):
    a = agg.get(k)
    if a:
        w = a["SQ_WAVES"] or 1
        s += " | %s VALU %.0f SALU %.0f LDS %.0f" % (k[2:], a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w)
print(s, flush=True)
PY
done
cp /tmp/orig.so $ROOT/libav_amd/libmi355dsp.so
