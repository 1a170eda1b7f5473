#!/bin/bash
# Round 6 experiment (GPU box): the config-3 chain with its filters as picture-level deblocking + SAO launches / one fused launch / vertical edges in the picture + fused horizontal edges and SAO
cd $GRAFT_REPO_ROOT
short() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], 'ms', round(d['ms_per_step'],3), 'frac', round(d['fraction_of_hbm_roofline'],4))" "$1"; }
for r in 1 2; do
python tools/hevc_chain.py 64 | short split
MI355_CHAIN_FILTER_FUSED=1 python tools/hevc_chain.py 64 | short fused_vhs
MI355_CHAIN_FILTER_FUSED=1 MI355_FT_SKIP_V=1 MI355_DEBLOCK_DIRS=1 python tools/hevc_chain.py 64 | short v_then_fused_hs
done
