cd $GRAFT_REPO_ROOT
TAG=r02_z2; mkdir -p gpurun_out/$TAG
SECONDS=0; timeout 900 python bench.py > gpurun_out/$TAG/bench.log 2> gpurun_out/$TAG/bench.err; echo "bench exit $?"
echo "bench wall ${SECONDS}s"; grep -E "bench \+" gpurun_out/$TAG/bench.err | tail -6
tail -1 gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench.json"))
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], "excl", r.get("exclusive",{}).get("kernel"), r.get("exclusive",{}).get("frac"),
      "alt", d.get("alt_math",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), "stock", d.get("stock_gpu_baseline",{}).get("value"), d.get("vs_stock_gpu"), "traffic", r.get("traffic"))
for o in d.get("other_configs", []): print("other", o["value"], o["ms_per_step"], o["roofline"]["kernel"], o["roofline"]["achieved"])
PY
