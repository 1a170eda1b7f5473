#!/bin/bash
# Round-5 GPU session for the committed evidence: what the driver runs (smoke, pytest -m gpu, bench at N=1 directly and under
# torch.distributed.run, and `bench.py --gpus 1` spawning nothing) plus the rocprof summaries that go to profiles/: kernel stats of the
# headline launches alone (--no-extra), HBM traffic (FETCH_SIZE / WRITE_SIZE passes), instruction counters of the two hot kernels.
# Usage (repo root, via gpurun): bash tools/gpu_r05_final.sh <tag>
set -u
TAG=${1:-r05final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 1200 python bench.py > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.txt; wc -c $OUT/bench.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_torchrun.txt 2> $OUT/bench_torchrun.err; echo "torchrun bench rc=$?"; cut -c1-200 $OUT/bench_torchrun.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $OUT/bench_under_rocprof.txt 2> $OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec rm {} \;
head -8 $OUT/kernel_stats.csv
bash tools/gpu_traffic.sh ${TAG}_traffic --no-extra --steps 1 --warmup 0 2>&1 | tail -2
for k in k_recon_inter_tiled k_deblock_tiled k_recon_intra; do
  bash tools/pmc_kernel.sh $k python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --frames 2048 --steps 1 --warmup 0 > $OUT/pmc_$k.txt 2>&1
  tail -22 $OUT/pmc_$k.txt | head -30
done
find $OUT -name '*.csv' -size +1M -delete
