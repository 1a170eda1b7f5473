#!/bin/bash
# GPU box: every build/streams/sweep_*.samples (tools/gen_sweep_streams.py) through _ref/h264_bridge_gpu with the bridge (2 decoder threads) and plain;
# md5 of the outputs compared, pictures on the device counted.  usage: bash tools/run_sweep_streams.sh <tag>
TAG=${1:-sweep}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ok=0; bad=0; skip=0
for s in build/streams/sweep_*.samples; do
  n=$(basename $s .samples)
  MI355_BRIDGE_PLAIN=1 timeout 120 oracle/_ref/h264_bridge_gpu $s /tmp/p.yuv 1 1 > /tmp/p.json 2> /tmp/p.err
  if [ -s /tmp/p.err ]; then skip=$((skip+1)); continue; fi
  timeout 120 oracle/_ref/h264_bridge_gpu $s /tmp/b.yuv 2 1 > /tmp/b.json 2> /tmp/b.err
  a=$(md5sum < /tmp/p.yuv); b=$(md5sum < /tmp/b.yuv)
  dev=$(tail -1 /tmp/b.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['pictures_on_device'], j['pictures_output'])")
  if [ "$a" = "$b" ]; then ok=$((ok+1)); echo "$n OK on device $dev" >> $OUT/sweep.txt; else bad=$((bad+1)); echo "$n MISMATCH on device $dev $(head -c 200 /tmp/b.err)" | tee -a $OUT/sweep.txt; fi
done
echo "identical $ok, different $bad, rejected by the reference decoder $skip" | tee -a $OUT/sweep.txt
