#!/bin/bash
# Developer measurement: N decoders in ONE process (a thread each) through the HEVC Tier-2 bridge — shared launches (1, 4, 8 launch sets side by side),
# their own launches (MI355_HEVC_BRIDGE_SOLO=1; on 8 streams and on the default stream) — and the reference's C decoder with as many threads.
# Usage (via gpurun): bash tools/hevc_bridge_threads.sh [stream [loops ["thread counts"]]]
S=${1:-pb_1080p_few_intra}; L=${2:-4}; NS=${3:-1 4 16 32}
cd $GRAFT_REPO_ROOT
EXE=oracle/_ref/hevc_bridge_gpu; SRC=tests/golden/hevc_synth_$S.samples
show() { python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sets %4d  pictures/set %5.2f  launches %7d  pictures/s %6.1f  identical %s' % (d['launch_sets'], d['pictures_per_launch_set'], d['reconstruction_launches'], d['pictures_per_s'], d['outputs_identical']))"; }
for N in $NS; do
  echo "== $N decoder(s), $L passes each over $S"
  echo -n "shared, 4 sets side by side (default): "; $EXE $SRC - $L $N | show
  echo -n "shared, 1 set at a time:               "; MI355_HEVC_BRIDGE_SETS_IN_FLIGHT=1 $EXE $SRC - $L $N | show
  echo -n "shared, 8 sets side by side:           "; MI355_HEVC_BRIDGE_SETS_IN_FLIGHT=8 $EXE $SRC - $L $N | show
  echo -n "own launches, 8 side by side:          "; MI355_HEVC_BRIDGE_SOLO=1 MI355_HEVC_BRIDGE_SETS_IN_FLIGHT=8 $EXE $SRC - $L $N | show
  echo -n "own launches, default stream (before): "; MI355_HEVC_BRIDGE_SOLO=1 MI355_HEVC_BRIDGE_DEFAULT_STREAM=1 $EXE $SRC - $L $N | show
  echo -n "C decoder:                             "; MI355_HEVC_RECON_PLAIN=1 MI355_HEVC_LF_PLAIN=1 $EXE $SRC - $L $N | show
done
