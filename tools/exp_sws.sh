#!/bin/bash
# Developer experiment: instruction counts and time of k_sws_generic with parts compiled out (-DMI355_SWS_NO_*).
set -e
export TMPDIR=/tmp
CFG=${1:-hd_generic}
for v in BASE MI355_SWS_NO_H MI355_SWS_NO_V MI355_SWS_NO_OUT "MI355_SWS_NO_H -DMI355_SWS_NO_V -DMI355_SWS_NO_OUT"; do
  rm -rf /tmp/exps && mkdir -p /tmp/exps && cp -r libav_amd include tests oracle tools /tmp/exps/
  ( cd /tmp/exps && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -D$v -I include -o libav_amd/libmi355dsp.so libav_amd/csrc/*.hip )
  ( cd /tmp/exps && python tools/bench_sws.py --configs $CFG --steps 10 2>&1 | python3 -c "import sys,json; [print('   ms %.4f' % json.loads(l)['ms_per_launch']) for l in sys.stdin if l.startswith('{')]" )
  ( cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/exps/out -- python /tmp/exps/tools/bench_sws.py --configs $CFG --steps 1 > /tmp/exps/log.txt 2>&1 )
  python3 - "$v" <<'PY'
import csv, glob, collections, sys
a = collections.defaultdict(float)
for f in glob.glob("/tmp/exps/out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sws_generic" in r["Kernel_Name"]:
            a[r["Counter_Name"]] += float(r["Counter_Value"])
w = a["SQ_WAVES"] or 1
print("%-60s VALU %.0f SALU %.0f LDS %.0f per wave" % (sys.argv[1], a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w), flush=True)
PY
done
