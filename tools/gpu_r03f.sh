#!/bin/bash
# Round 3: per-kernel times and counters of the config-3 (HEVC chain) and config-5 (swscale) kernels -> gpurun_out/<tag>/*.json
# (copied to profiles/ by hand).  Counters in their own rocprofv3 runs; FETCH_SIZE and WRITE_SIZE in separate passes.
set -u
TAG=${1:-r03f}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run_set() {   # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -- "$@" > $OUT/${name}_stats.log 2>&1; echo "$name stats rc=$?"
  local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
             "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/${name}_p$i -- "$@" > $OUT/${name}_p$i.log 2>&1; echo "$name pmc $i rc=$?"
  done
  python3 - "$OUT" "$name" "$*" <<'PY'
import csv, glob, collections, json, sys
out, name, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"command": cmd, "kernels": collections.defaultdict(dict)}
for f in glob.glob("%s/%s_stats/**/*kernel_stats.csv" % (out, name), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
        res["kernels"][k].update({"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6, "percent": float(r["Percentage"])})
for d in sorted(glob.glob("%s/%s_p*/" % (out, name))):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        tot = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
                n[k] += 1
        for k in tot:
            for c, v in tot[k].items():
                res["kernels"][k][c + "_per_launch"] = v / max(1, n[k])
res["note"] = "per-launch averages; FETCH_SIZE / WRITE_SIZE raw counter values (KiB: x 1024 = bytes; FETCH_SIZE counts 128-byte-coalesced reads at half size, profiles/r02_calibration.md)"
json.dump(res, open("%s/%s_kernels.json" % (out, name), "w"), indent=1)
for k, v in res["kernels"].items():
    if k.startswith("k_"):
        print(name, k, {a: round(b, 1) for a, b in v.items()})
PY
  find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.csv' -size +2M -delete
}
run_set hevc_chain python $GRAFT_REPO_ROOT/tools/hevc_chain.py 64

run_set sws python $GRAFT_REPO_ROOT/tools/bench_sws.py --frames 32 --steps 5
