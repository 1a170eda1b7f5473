#!/bin/bash
# Round-2 GPU session: parity tests, bench, kernel stats and the instruction-mix counters of the H.264 kernels.
# Usage (on the GPU box, from the repo root): bash tools/gpu_r02.sh <tag> [quick]
set -u
TAG=${1:-r02a}; MODE=${2:-full}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$MODE" = quick ]; then
  timeout 900 python -m pytest tests/test_frame_gpu.py tests/test_stream_parity.py tests/test_tier1_gpu.py -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
else
  timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"
fi
tail -3 $OUT/pytest_gpu.txt
timeout 300 python bench.py --no-cpu-baseline --no-extra > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); print('MB/s %.1fM' % (d['value']/1e6), d['pass_ms'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $OUT/bench_rocprof.json 2> $OUT/stats.err; echo "stats rc=$?"
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  n=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$n -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --frames 256 --steps 1 --warmup 0 > $OUT/pmc_$n.log 2>&1; echo "pmc $n rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
per_file = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if k.startswith("k_"):
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in tot:
        for c, v in tot[k].items():
            per_file[k][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in a.items()} for k, a in per_file.items()}
for k, a in out.items():
    w = a.get("SQ_WAVES") or 1
    a["per_wave"] = {c: round(v / w, 1) for c, v in a.items() if c.startswith("SQ_INSTS")}
json.dump(out, open("$OUT/pmc.json", "w"), indent=1)
for k, a in out.items():
    print(k, "waves", round(a.get("SQ_WAVES", 0)), a["per_wave"], {c: round(v) for c, v in a.items() if not c.startswith("SQ_INSTS") and c not in ("per_wave", "SQ_WAVES")})
PY
