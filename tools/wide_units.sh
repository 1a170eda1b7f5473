#!/bin/bash
# Developer experiment: the second kernel set's loop filter with 1 / 2 / 4 / 8 macroblocks per group and launch (MI355_WIDE_UNIT), several batch sizes.
cd $GRAFT_REPO_ROOT
for F in ${FS:-16 64 512 2048}; do
  for U in 1 2 4 8; do
    echo -n "F=$F unit=$U: "; MI355_WIDE_UNIT=$U python tools/wide_times.py $F 10 2>&1 | grep noise | sed 's/.*deblock //'
  done
done
