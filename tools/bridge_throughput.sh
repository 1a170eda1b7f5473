#!/bin/bash
# End-to-end throughput of the reference decoder + Tier-2 bridge on realshort.mp4 (GPU box, repo root):
# N decoder threads = N streams, each decoding the clip `loops` times.  -> gpurun_out/<tag>/bridge.jsonl
TAG=${1:-bridge}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, os, struct
sys.path.insert(0, "tests/golden")
import mp4_samples
avcc, samples = mp4_samples.extract("/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4")
with open("/tmp/realshort.samples", "wb") as f:
    f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", len(samples)))
    for s in samples:
        f.write(struct.pack("<I", len(s)) + s)
PY
: > $OUT/bridge.jsonl
for mode in "" "MI355_BRIDGE_LAZY=1"; do
  for t in 1 8 32 128; do
    env $mode oracle/_ref/h264_bridge_gpu /tmp/realshort.samples - $t 20 2>/dev/null | sed "s/^{/{\"mode\": \"${mode:-sync}\", /" | tee -a $OUT/bridge.jsonl
  done
done
