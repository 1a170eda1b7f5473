for f in h264_synth_1080p_high10 h264_synth_1080p; do
for mode in bridge plain; do
  if [ $mode = plain ]; then export MI355_BRIDGE_PLAIN=1; else unset MI355_BRIDGE_PLAIN; fi
  echo -n "$f $mode: "; oracle/_ref/h264_bridge_gpu tests/golden/$f.samples - 64 3 2>/dev/null | tail -1 | cut -c1-260
done; done
unset MI355_BRIDGE_PLAIN
echo -n "high10 lazy: "; MI355_BRIDGE_LAZY=1 oracle/_ref/h264_bridge_gpu tests/golden/h264_synth_1080p_high10.samples - 64 3 2>/dev/null | tail -1 | cut -c1-260
echo -n "high10 16 threads: "; oracle/_ref/h264_bridge_gpu tests/golden/h264_synth_1080p_high10.samples - 16 3 2>/dev/null | tail -1 | cut -c1-260
