#!/bin/bash
# Developer loop: the library from per-file objects (only changed sources are recompiled, in parallel).
#   tools/fastbuild.sh [-o out.so] [-f "<flags for the files named after it>" file.hip ...]
# __graft_entry__.build() compiles everything in one hipcc call; this produces the same library faster while iterating.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
OUT=$ROOT/libav_amd/libmi355dsp.so; FLAGS=""; SPECIAL=""
while [ $# -gt 0 ]; do
  case $1 in
    -o) OUT=$2; shift 2;;
    -f) FLAGS=$2; shift 2; SPECIAL="$@"; break;;
    *) shift;;
  esac
done
OBJ=$ROOT/build/obj; mkdir -p $OBJ
pids=""
objs=""
for src in $ROOT/libav_amd/csrc/*.hip; do
  b=$(basename $src .hip); o=$OBJ/$b.o; fl=""
  for s in $SPECIAL; do if [ "$s" = "$b.hip" ] || [ "$s" = "$b" ]; then fl=$FLAGS; o=$OBJ/$b.$(echo "$FLAGS" | md5sum | cut -c1-8).o; fi; done
  objs="$objs $o"
  new=0
  if [ ! -f $o ]; then new=1; else
    for d in $src $ROOT/libav_amd/csrc/*.h $ROOT/include/*.h; do if [ $d -nt $o ]; then new=1; break; fi; done
  fi
  if [ $new = 1 ]; then
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $fl -I $ROOT/include -c -o $o $src ) &
    pids="$pids $!"
    if [ $(jobs -r | wc -l) -ge 8 ]; then wait -n; fi
  fi
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $OUT $objs
ls -la $OUT
