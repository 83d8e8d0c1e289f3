cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01_h
timeout 900 python bench.py > gpurun_out/r01_h/bench2.log 2>&1; echo "bench exit $?"
tail -1 gpurun_out/r01_h/bench2.log > gpurun_out/r01_h/bench2.json
python - <<PY
import json
d=json.load(open("gpurun_out/r01_h/bench2.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], r["traffic"], r.get("exclusive",{}).get("achieved"), d.get("alt_math",{}).get("value"), d.get("cpu_baseline",{}).get("value"))
PY
