#!/bin/bash
# Developer experiment: memory-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the second H.264 kernel set's kernels over one
# run of tools/wide_times.py.  Usage (via gpurun): bash tools/wide_traffic.sh [frames]
F=${1:-512}
export TMPDIR=/tmp
OUT=/tmp/wtraf; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python $GRAFT_REPO_ROOT/tools/wide_times.py $F 10 > $OUT/$c.log 2>&1
  echo "$c rc=$?"; tail -2 $OUT/$c.log
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.Counter(); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    for k in sorted(tot):
        if "wide" in k: print(c, k, "launches", n[k], "total raw %.4g" % tot[k])
PY
