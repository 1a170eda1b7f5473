#!/bin/bash
# HBM-side traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md, "HBM": TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2).
# Usage: bash tools/gpu_traffic.sh <tag> [bench args...]   -> gpurun_out/<tag>/traffic.json
set -u
TAG=${1:-traffic}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/$c.log 2>&1
  echo "$c rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        tot = collections.Counter(); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
        for k in tot:
            if not k.startswith("__"):
                res[k][c + "_per_launch_raw"] = tot[k] / n[k]; res[k]["launches"] = n[k]
json.dump(res, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
