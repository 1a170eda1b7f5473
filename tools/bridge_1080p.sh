#!/bin/bash
# End-to-end throughput of the reference decoder + Tier-2 bridge on a GENERATED 1080p stream (tests/golden/h264_synth_1080p.samples:
# tools/make_1080p_stream.py: mb_w=120, mb_h=68, 10 pictures I/P/B, 4 slices, 8x8 transform, three references, sparse
# residuals — 109 KB per picture).  GPU box, repo root: tools/bridge_1080p.sh <tag>   -> gpurun_out/<tag>/bridge_1080p.jsonl
TAG=${1:-bridge1080}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
S=tests/golden/h264_synth_1080p.samples
MI355_BRIDGE_PLAIN=1 oracle/_ref/h264_bridge_gpu $S /tmp/plain.yuv 1 1 > /dev/null 2>&1
oracle/_ref/h264_bridge_gpu $S /tmp/bridge.yuv 1 1 2>&1 | tail -1 | cut -c1-200
cmp /tmp/plain.yuv /tmp/bridge.yuv && echo "bridge output identical to the reference decoder's ($(md5sum < /tmp/plain.yuv | cut -c1-32))" | tee $OUT/identical.txt
OUTF=$OUT/bridge_1080p.jsonl; : > $OUTF
run() { env $2 timeout 120 oracle/_ref/h264_bridge_gpu $S - $3 $4 2>/dev/null | sed "s/^{/{\"clip\": \"synth_1080p\", \"mode\": \"$1\", /" | tee -a $OUTF | cut -c1-230; }
for t in 1 32 128 256; do run plain "MI355_BRIDGE_PLAIN=1" $t 3; done
for t in 1 8 32 128 256; do run batched "X=1" $t 3; done
for t in 1 128; do run batched_lazy "MI355_BRIDGE_LAZY=1" $t 3; done
