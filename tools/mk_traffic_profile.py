#!/usr/bin/env python3
"""gpurun_out/<tag>/traffic.json (tools/gpu_traffic.sh) -> profiles/<name>_hbm_traffic_f<F>.json, the file
bench.py reads for roofline.traffic.  Usage: tools/mk_traffic_profile.py <tag> <name> <frames PER LAUNCH> (bench.py's default since round 5:
two pipelines of 1024 pictures: 1024)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
tiled = (sys.argv[4] if len(sys.argv) > 4 else "tiled") == "tiled"
t = json.load(open(os.path.join(ROOT, "gpurun_out", tag, "traffic.json")))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 0 "
                  "(%d pictures per launch)" % frames,
       "frames_per_gpu": frames, "unit": "bytes per launch",
       "note": "raw counter x 1024 (counters are in KiB).  Calibration (profiles/r02_calibration.md): FETCH_SIZE counts reads that "
               "reach the L2 as 128-byte requests at half their size and everything else exactly; WRITE_SIZE is exact.  "
               "k_recon_inter / k_recon_inter_tiled: the only 128-byte-coalesced stream is the coefficients (768 B per inter macroblock), so "
               "traffic_bytes = fetch + 0.5 x coefficient bytes + write; k_recon_intra: as reported; k_deblock / k_deblock_tiled (round 4: one launch for all bands, tiles fetched as 16-byte pieces by LDS-DMA): as reported on surfaces with line "
               "strides — on macroblock-tiled surfaces (the default since round 3) its sample reads are whole 128-byte lines (a macroblock = three lines), "
               "so half of the 384 sample bytes per macroblock are added back",
       "surface_layout": "tiled" if tiled else "linear",
       "kernels": {}}
for k, v in t.items():
    f, w = v["FETCH_SIZE_per_launch_raw"] * 1024, v["WRITE_SIZE_per_launch_raw"] * 1024
    corr = 0.5 * 768 * frames * 8160 * 0.95 if k in ("k_recon_inter", "k_recon_inter_tiled") else 0.0      # coefficient stream, 95 % inter macroblocks
    if k in ("k_deblock", "k_deblock_tiled") and tiled:
        corr = 0.5 * 384 * frames * 8160 / (max(1, v["launches"]) if k == "k_deblock" else 1)     # per launch (k_deblock, round 3: a band per launch; k_deblock_tiled: all bands of `frames` pictures): the tiles' lines
    out["kernels"][k] = {"launches": v["launches"], "fetch_bytes_reported": f, "fetch_correction_bytes": corr, "write_bytes": w,
                         "traffic_bytes": f + corr + w}
path = os.path.join(ROOT, "profiles", "%s_hbm_traffic_f%d.json" % (name, frames))
json.dump(out, open(path, "w"), indent=1)
print(path)
