#!/bin/bash
# tools/build_ref_variant.sh <git ref> <name> ["<extra hipcc flags>"]: the library as of <ref> into build/variants/<name>.so (the same-box A/B partner of the working tree's build)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
REF=$1; NAME=$2; FLAGS=${3:-}
T=$ROOT/build/ref_src/$NAME
rm -rf $T && mkdir -p $T $ROOT/build/variants
git -C $ROOT archive $REF libav_amd/csrc include | tar -x -C $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $FLAGS -I $T/include -o $ROOT/build/variants/$NAME.so $T/libav_amd/csrc/*.hip
ls -la $ROOT/build/variants/$NAME.so
