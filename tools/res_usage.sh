#!/bin/bash
# resource usage (registers, LDS, occupancy) of the kernels of one .hip file, compiled for gfx950: tools/res_usage.sh h264_deblock [kernel substring] [extra hipcc flags]
f=${1:-h264_deblock}; k=${2:-}; shift 2 2>/dev/null
cd "$(dirname "$0")/.." && mkdir -p build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -gline-tables-only --cuda-device-only -Rpass-analysis=kernel-resource-usage -I include "$@" \
    -o build/isa/$f.s libav_amd/csrc/$f.hip 2>&1 | grep -E "Function Name|SGPRs:|VGPRs:|Occupancy|Spill|LDS Size|ScratchSize" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | awk -v k="$k" '/Function Name/ {show = (k == "" || index($0, k) > 0)} show {printf "%s%s", $0, (/LDS Size/ ? "\n" : " | ")}'
