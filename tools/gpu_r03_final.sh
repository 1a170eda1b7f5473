#!/bin/bash
# Round-3 GPU session for the committed evidence: what the driver runs (smoke, pytest -m gpu, bench at N=1 directly and under
# torch.distributed.run) plus the rocprof summaries that go to profiles/: kernel stats of the headline launch alone
# (--no-extra: the csv row IS the F=2048 launch), calibrated HBM traffic, instruction / L2-request counters.
set -u
TAG=${1:-r03final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 1200 python bench.py > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_torchrun.txt 2> $OUT/bench_torchrun.err; echo "torchrun bench rc=$?"; cut -c1-200 $OUT/bench_torchrun.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.txt 2> $OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec rm {} \;
head -6 $OUT/kernel_stats.csv
bash tools/gpu_traffic.sh ${TAG}_traffic --no-extra --steps 1 --warmup 0 2>&1 | tail -2
cd /tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
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
json.dump({"command": "rocprofv3 --pmc <set> -- python bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 (2048 pictures, one step; totals over all launches of the kernel in that step)", "kernels": res}, open("$OUT/pmc.json", "w"), indent=1)
for k, a in res.items():
    print(k, {c: "%.4g" % v for c, v in sorted(a.items())})
PY
find $OUT -name '*.csv' -size +1M -delete
# config-3 / config-5 kernels: times, traffic and instruction counters (own rocprofv3 runs) -> <tag>_k/{hevc_chain,sws}_kernels.json
cd $GRAFT_REPO_ROOT && bash tools/gpu_r03f.sh ${TAG}_k 2>&1 | tail -30
