#!/bin/bash
# host side of the HEVC bridge without a device: the reference's decoder + the two bridges linked against tools/null_device (kernels
# return at once, copies are memcpy) beside the plain decoder, N passes over a stream.  CPU only; needs /root/reference (oracle/_ref objects).
# usage: tools/host_side_time.sh [stream name] [passes]        -> seconds per picture, with -pg: gprof's flat profile of the host functions
set -e
cd "$(dirname "$0")/../oracle"
S=${1:-pb_1080p_few_intra}; N=${2:-5}
make -s _ref/hevc_bridge_emu
CMD=$(make -n -B _ref/hevc_bridge_emu 2>/dev/null | grep -- "-o _ref/hevc_bridge_emu" | tail -1)
CMD=${CMD//-o _ref\/hevc_bridge_emu/-o _ref\/hevc_bridge_null}
CMD=${CMD//-L..\/tests\/_emu -lmi355dsp_emu/..\/tools\/null_device\/null_dev.c}
CMD=${CMD//cc -O3/cc -O3 $PROFILE_FLAGS}
eval "$CMD"
echo "== plain decoder"; MI355_HEVC_RECON_PLAIN=1 MI355_HEVC_LF_PLAIN=1 _ref/hevc_bridge_null ../tests/golden/hevc_synth_$S.samples - $N | tail -1
echo "== bridges over the null device (host side only)"; _ref/hevc_bridge_null ../tests/golden/hevc_synth_$S.samples - $N | tail -1
echo "== the same, two launches per intra block"; MI355_HEVC_BRIDGE_SPLIT_INTRA=1 _ref/hevc_bridge_null ../tests/golden/hevc_synth_$S.samples - $N | tail -1
