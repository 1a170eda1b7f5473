#!/bin/bash
# Round 3, session A: tiled-surface parity on the GPU, linear vs tiled A/B of the bench, instruction / L2-request counters.
# Usage (GPU box, repo root): bash tools/gpu_r03a.sh <tag>
set -u
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_frame_gpu.py tests/test_field_gpu.py -m gpu -q -x > $OUT/pytest_frame.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_frame.txt
for lay in linear tiled; do
  timeout 600 python bench.py --no-extra --no-cpu-baseline --layout $lay > $OUT/bench_$lay.txt 2> $OUT/bench_$lay.err; echo "bench $lay rc=$?"
  python - <<PY
import json
d = json.load(open("$OUT/bench_$lay.txt"))
print("$lay", round(d["value"] / 1e6, 1), "M MB/s", d["pass_ms"], d["roofline"]["frac"])
PY
done
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 > $OUT/p$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k in agg:
            if k.startswith("k_"):
                for c, v in agg[k].items():
                    res[k].setdefault(c, v)
json.dump(res, open("$OUT/pmc.json", "w"), indent=1)
for k, a in res.items():
    print(k)
    for c, v in sorted(a.items()):
        print("   %-40s %.5g" % (c, v))
PY
find $OUT -name '*.csv' -size +1M -delete
