#!/bin/bash
# Developer experiment: executed VALU / SALU / LDS instructions per wave of k_recon_inter with parts of the kernel
# compiled out (-DMI355_EXP_NO_*), via one rocprofv3 --pmc pass each on a small batch.
set -e
export TMPDIR=/tmp
for v in BASE MI355_EXP_NO_STAGE MI355_EXP_NO_LUMA MI355_EXP_NO_CHROMA MI355_EXP_NO_RESIDUAL "MI355_EXP_NO_LUMA -DMI355_EXP_NO_CHROMA -DMI355_EXP_NO_STAGE -DMI355_EXP_NO_RESIDUAL"; do
  rm -rf /tmp/expv && mkdir -p /tmp/expv && cp -r libav_amd include tests oracle bench.py profiles /tmp/expv/
  ( cd /tmp/expv && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -D$v -I include -o libav_amd/libmi355dsp.so libav_amd/csrc/*.hip )
  ( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/expv/out -- python /tmp/expv/bench.py --no-cpu-baseline --frames 64 --steps 1 --warmup 0 > /tmp/expv/log.txt 2>&1 )
  python3 - "$v" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/expv/out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_recon_inter" in r["Kernel_Name"]:
            agg["k"][r["Counter_Name"]] += float(r["Counter_Value"])
a = agg["k"]
w = a["SQ_WAVES"] or 1
print("%-90s VALU %.0f SALU %.0f LDS %.0f per wave" % (sys.argv[1], a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w), flush=True)
PY
done
