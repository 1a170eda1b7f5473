#!/usr/bin/env python3
"""gpurun_out/<tag>/traffic.json (tools/gpu_traffic.sh) -> profiles/<name>_hbm_traffic_f<F>.json, the file
bench.py reads for roofline.traffic.  Usage: tools/mk_traffic_profile.py <tag> <name> <frames>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
t = json.load(open(os.path.join(ROOT, "gpurun_out", tag, "traffic.json")))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --no-cpu-baseline "
                  "--frames %d --steps 1 --warmup 0" % frames,
       "frames_per_gpu": frames, "unit": "bytes per launch",
       "note": "raw counter x 1024 (counters are in KiB). MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide "
               "coalesced reads by up to 2x; fetch_bytes_x2 is the upper bound, WRITE_SIZE is uncalibrated but lands "
               "within a few % of the algorithmic store bytes here",
       "kernels": {}}
for k, v in t.items():
    f, w = v["FETCH_SIZE_per_launch_raw"] * 1024, v["WRITE_SIZE_per_launch_raw"] * 1024
    out["kernels"][k] = {"launches": v["launches"], "fetch_bytes": f, "fetch_bytes_x2": 2 * f, "write_bytes": w,
                         "traffic_bytes": f + w, "traffic_bytes_upper": 2 * f + w}
path = os.path.join(ROOT, "profiles", "%s_hbm_traffic_f%d.json" % (name, frames))
json.dump(out, open(path, "w"), indent=1)
print(path)
