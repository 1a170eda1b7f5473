#!/bin/bash
# End-to-end throughput of the reference decoder + Tier-2 bridge on realshort.mp4 (GPU box, repo root):
# N decoder threads = N streams, each decoding the clip `loops` times.  -> gpurun_out/<tag>/bridge.jsonl
# modes: plain = the reference's C path alone (CPU), batched = pictures of all streams through the dispatcher,
# direct = one HIP stream and one launch set per decoder thread; "lazy" = wait only for the picture about to be output.
TAG=${1:-bridge}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
CLIP=${2:-realshort}     # realshort (320x240 4:2:0, 36 pictures) or cockatoo (1280x720 4:4:4, 280 pictures, P + B)
cd $GRAFT_REPO_ROOT
CLIP=$CLIP python3 - <<'PY'
import sys, os, struct
sys.path.insert(0, "tests/golden")
import mp4_samples
import os
clip = os.environ.get("CLIP", "realshort")
avcc, samples = mp4_samples.extract("/opt/conda/lib/python3.9/site-packages/imageio/resources/images/%s.mp4" % clip)
with open("/tmp/%s.samples" % clip, "wb") as f:
    f.write(struct.pack("<I", len(avcc)) + avcc + struct.pack("<I", len(samples)))
    for s in samples:
        f.write(struct.pack("<I", len(s)) + s)
PY
OUTF=$OUT/bridge_$CLIP.jsonl
: > $OUTF
LOOPS=20; [ "$CLIP" = cockatoo ] && LOOPS=1
run() {  # name, env assignments, threads, loops
  env $2 timeout 200 oracle/_ref/h264_bridge_gpu /tmp/$CLIP.samples - $3 $4 2>/dev/null | sed "s/^{/{\"clip\": \"$CLIP\", \"mode\": \"$1\", /" | tee -a $OUTF
}
if [ "$CLIP" = cockatoo ]; then
  for t in 1 16 64 128; do run plain "MI355_BRIDGE_PLAIN=1" $t $LOOPS; done
  for t in 1 16 64 128; do run batched "X=1" $t $LOOPS; done
  for t in 1 64; do run batched_lazy "MI355_BRIDGE_LAZY=1" $t $LOOPS; done
  for t in 1 16; do run direct "MI355_BRIDGE_DIRECT=1" $t $LOOPS; done
else
  for t in 1 32 128 256; do run plain "MI355_BRIDGE_PLAIN=1" $t $LOOPS; done
  for t in 1 8 32 128 256; do run batched "X=1" $t $LOOPS; done
  for t in 32 128; do run batched_linear "MI355_BRIDGE_LINEAR=1" $t $LOOPS; done
  for t in 1 32; do run sessions "MI355_BRIDGE_SESSION=1" $t $LOOPS; done
  for t in 1 32 128 256; do run batched_lazy "MI355_BRIDGE_LAZY=1" $t $LOOPS; done
  for t in 1 8 32; do run direct "MI355_BRIDGE_DIRECT=1" $t $LOOPS; done
fi
