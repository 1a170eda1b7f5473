#!/bin/bash
# Developer measurement: N decoders in ONE process (a thread each) through the HEVC Tier-2 bridge — shared launches, their own launches
# (MI355_HEVC_BRIDGE_SOLO=1) — and the reference's C decoder with as many threads.  Usage (via gpurun): bash tools/hevc_bridge_threads.sh [stream [loops]]
S=${1:-pb_1080p_few_intra}; L=${2:-4}
cd $GRAFT_REPO_ROOT
EXE=oracle/_ref/hevc_bridge_gpu; SRC=tests/golden/hevc_synth_$S.samples
for N in 1 4 16 32; do
  echo "== $N decoder(s), $L passes each over $S"
  echo -n "shared launches: "; $EXE $SRC - $L $N | tail -1
  echo -n "own launches:    "; MI355_HEVC_BRIDGE_SOLO=1 $EXE $SRC - $L $N | tail -1
  echo -n "C decoder:       "; MI355_HEVC_RECON_PLAIN=1 MI355_HEVC_LF_PLAIN=1 $EXE $SRC - $L $N | tail -1
done
