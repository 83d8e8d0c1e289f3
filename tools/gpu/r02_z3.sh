cd $GRAFT_REPO_ROOT
TAG=${1:-r02_z3}; mkdir -p gpurun_out/$TAG
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/$TAG/smoke.log
SECONDS=0; timeout 900 python bench.py > gpurun_out/$TAG/bench.log 2> gpurun_out/$TAG/bench.err; echo "bench exit $? wall ${SECONDS}s"
tail -1 gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench.json"))
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], "excl", r.get("exclusive",{}).get("frac"),
      "alt", d.get("alt_math",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), "stock", d.get("stock_gpu_baseline",{}).get("value"), d.get("vs_stock_gpu"), "traffic", r.get("traffic"))
for o in d.get("other_configs", []): print("other", o["value"], o["ms_per_step"], o["roofline"]["kernel"], o["roofline"]["achieved"])
for w in r.get("wasp_dilated", []): print("wasp", w["dilation"], w["ms"], w["effective_mfma_frac"])
PY
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/$TAG/pytest_gpu.log | tail -1
UNIPOSE_CONV_MATH=bf16x3 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 > gpurun_out/$TAG/pytest_gpu_bf16x3.log 2>&1; echo "pytest(bf16x3 default) exit $?"
grep -E "passed|failed" gpurun_out/$TAG/pytest_gpu_bf16x3.log | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_lstm -o bench -- python $GRAFT_REPO_ROOT/bench.py --model lstm --num-classes 13 --batch 8 --frames 5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_lstm.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_lstm -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_lstm.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
head -6 gpurun_out/$TAG/kernel_stats_lstm.txt
