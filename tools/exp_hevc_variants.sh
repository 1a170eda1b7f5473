#!/bin/bash
# Developer experiment (GPU box): per-kernel times of the HEVC chain (tools/hevc_chain.py 64) for every library in build/variants
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp libav_amd/libmi355dsp.so /tmp/orig.so
for round in 1 2; do
for so in build/variants/*.so; do
  cp $so libav_amd/libmi355dsp.so
  rm -rf /tmp/hs; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hs -- python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64 > /tmp/hs.log 2>&1 )
  echo "== $(basename $so .so): $(find /tmp/hs -name '*kernel_stats.csv' -exec cat {} \; | grep 'k_hevc\|k_edge' | sed 's/(anonymous namespace):://; s/(.*)"//' | awk -F, '{printf "%s %.3f ms  ", $1, $4/1e6}')"
done
done
cp /tmp/orig.so libav_amd/libmi355dsp.so
