#!/bin/bash
# End-of-round GPU session: what the driver will run (smoke, pytest -m gpu, bench at N=1 directly and under
# torch.distributed.run), plus the rocprof summaries committed under profiles/.
set -u
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.txt 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.txt
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_torchrun.txt 2> $OUT/bench_torchrun.err; echo "torchrun bench rc=$?"; cut -c1-200 $OUT/bench_torchrun.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- \
    python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.txt 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name '*kernel_trace.csv' -exec rm {} \;
head -6 $OUT/kernel_stats.csv
bash tools/gpu_traffic.sh ${TAG}_traffic --no-extra --steps 1 --warmup 0 2>&1 | tail -2
