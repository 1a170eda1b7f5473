cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, json
sys.path.insert(0, "tools")
import libav_amd, hevc_chain
lib = libav_amd.load(0)
for p in (1, 2, 3):
    if p == 1:
        r = hevc_chain.measure(lib, 64, steps=5, cpu_seconds=0)
    else:
        r = hevc_chain.measure_pipelines(lib, 64, p, steps=8)
    print(p, "chains", round(r["ms_per_step"], 3), "ms", round(r["fraction_of_hbm_roofline"], 4))
PY
