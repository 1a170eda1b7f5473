#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of known byte counts (tools/ubench/copy_calib.hip, prebuilt as build/copy_calib)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-calib}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- $GRAFT_REPO_ROOT/build/copy_calib > $OUT/$c.log 2>&1; echo "$c rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        tot = collections.Counter(); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
        for k in tot:
            res[k][c + "_per_launch_raw"] = tot[k] / n[k]
json.dump(res, open("$OUT/calib.json", "w"), indent=1)
print(open("$OUT/FETCH_SIZE.log").read()[-400:])
for k, v in res.items(): print(k, v)
PY
