#!/bin/bash
# Round-6 GPU session for the committed evidence: what the driver runs (smoke, pytest -m gpu, bench at N=1 directly and under torch.distributed.run) plus the
# rocprof summaries that go to profiles/: kernel stats of the headline launches alone (--no-extra), of the config-3 chain; HBM traffic (FETCH_SIZE / WRITE_SIZE in
# separate passes) of both, with every launch of the profiled bench command share-sized (--no-alone); instruction counters of the hot kernels; the
# FETCH_SIZE calibration including the LDS-DMA shapes (build/copy_calib, built on the host by hipcc from tools/ubench/copy_calib.hip).
# Usage (repo root, via gpurun): bash tools/gpu_r06_final.sh <tag>
set -u
TAG=${1:-r06fin}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 1200 python bench.py > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-2600 $OUT/bench.txt; wc -c $OUT/bench.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_torchrun.txt 2> $OUT/bench_torchrun.err; echo "torchrun bench rc=$?"; cut -c1-200 $OUT/bench_torchrun.txt
rm -rf /tmp/prof_b; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-alone > $OUT/bench_under_rocprof.txt 2> $OUT/rocprof.err ); echo "rocprof bench rc=$?"
cp $(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_bench_f2048.csv; head -6 $OUT/kernel_stats_bench_f2048.csv | cut -c1-200
rm -rf /tmp/prof_h; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > $OUT/hevc_chain_under_rocprof.txt 2>&1 ); echo "rocprof chain rc=$?"
cp $(find /tmp/prof_h -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_hevc_chain.csv; head -8 $OUT/kernel_stats_hevc_chain.csv | cut -c1-200
python tools/hevc_chain.py 64 > $OUT/hevc_chain_one.json; python tools/hevc_chain.py 64 2 > $OUT/hevc_chain_two.json; cut -c1-300 $OUT/hevc_chain_one.json; cut -c1-200 $OUT/hevc_chain_two.json
# HBM traffic: the bench's share-sized launches, and the chain's kernels
bash tools/gpu_traffic.sh ${TAG}_traffic --no-extra --no-alone --steps 1 --warmup 0 2>&1 | tail -2
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 500 rocprofv3 --pmc $c --output-format csv -d $OUT/hevc_$c -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > $OUT/hevc_$c.log 2>&1 ); echo "hevc $c rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/hevc_%s/**/*counter_collection.csv" % c, recursive=True):
        tot = collections.Counter(); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
        for k in tot:
            if not k.startswith("__"):
                res[k][c + "_per_launch_raw_KiB"] = tot[k] / n[k]; res[k]["launches"] = n[k]
json.dump(res, open("$OUT/hevc_chain_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/hevc_FETCH_SIZE $OUT/hevc_WRITE_SIZE
for k in k_recon_inter_tiled k_deblock_tiled; do
  bash tools/pmc_kernel.sh $k python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --no-alone --frames 2048 --steps 1 --warmup 0 > $OUT/pmc_$k.txt 2>&1
  tail -21 $OUT/pmc_$k.txt | head -3
done
bash tools/pmc_kernel.sh k_hevc_recon_ctbs,k_hevc_sao_ctbs,k_hevc_deblock_pictures python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > $OUT/pmc_hevc_chain.txt 2>&1
[ -x build/copy_calib ] && bash tools/gpu_calib.sh ${TAG}_calib 2>&1 | tail -12
find $OUT -name '*.csv' -size +1M -delete
