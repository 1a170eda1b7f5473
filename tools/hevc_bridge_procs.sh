#!/bin/bash
# GPU box: N PROCESSES of the reference's HEVC decoder on one stream, with the Tier-2 bridges and plain (the bridges keep their state per process: one decoder each),
# aggregate pictures per second.  usage: bash tools/hevc_bridge_procs.sh <stream name> <processes> [loops]
S=tests/golden/hevc_synth_${1:-pb_1080p_few_intra}.samples; N=${2:-16}; L=${3:-10}
for mode in bridge plain; do
  if [ $mode = plain ]; then export MI355_HEVC_RECON_PLAIN=1 MI355_HEVC_LF_PLAIN=1; else unset MI355_HEVC_RECON_PLAIN MI355_HEVC_LF_PLAIN; fi
  t0=$(date +%s.%N)
  for i in $(seq $N); do oracle/_ref/hevc_bridge_gpu $S - $L > /tmp/hb_$i.json 2>/dev/null & done
  wait
  t1=$(date +%s.%N)
  python3 - "$mode" "$N" "$t0" "$t1" <<'PY'
import sys, json, glob
mode, n, t0, t1 = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
pics = sum(json.loads(open("/tmp/hb_%d.json" % i).read().strip().splitlines()[-1])["pictures_output"] for i in range(1, n + 1))
print("%s: %d processes, %d pictures in %.2f s wall (process start and device set-up included) = %.1f pictures/s" % (mode, n, pics, t1 - t0, pics / (t1 - t0)))
PY
done
