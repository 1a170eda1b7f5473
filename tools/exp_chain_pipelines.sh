#!/bin/bash
# Round 6 experiment (GPU box): the config-3 chain's 64 pictures as 1 / 2 / 3 / 4 chains on their own streams, with and without turns at a stage
cd $GRAFT_REPO_ROOT
short() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], 'ms', round(d['ms_per_step'],3), 'frac', round(d['fraction_of_hbm_roofline'],4))" "$1"; }
for r in 1 2; do
python tools/hevc_chain.py 64 | short one_chain
for p in 2 3 4; do
  python tools/hevc_chain.py 64 $p | short "chains_$p"
  python tools/hevc_chain.py 64 $p 1 | short "chains_${p}_turns_at_recon"
  python tools/hevc_chain.py 64 $p 4 | short "chains_${p}_turns_at_sao"
done
done
