# Round-end measurement set: default bench line (with cpu_baseline + alt_math), rocprofv3 kernel stats of the
# default command and of the exclusive (single-stream) variant, smoke, full gpu test suite.
cd $GRAFT_REPO_ROOT
TAG=${1:-r01_f}
mkdir -p gpurun_out/$TAG
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke exit $?"
timeout 900 python bench.py > gpurun_out/$TAG/bench.log 2>&1; echo "bench exit $?"
tail -1 gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("exclusive",{}).get("achieved"), d.get("alt_math",{}).get("value"), d.get("cpu_baseline",{}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_sync -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_sync.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats.txt 2>&1
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_sync -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_exclusive.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
head -12 gpurun_out/$TAG/kernel_stats.txt
head -12 gpurun_out/$TAG/kernel_stats_exclusive.txt
[ -n "$SKIP_TESTS" ] && exit 0
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -2 gpurun_out/$TAG/pytest_gpu.log
UNIPOSE_CONV_MATH=bf16x3 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 > gpurun_out/$TAG/pytest_gpu_bf16x3.log 2>&1; echo "pytest(bf16x3 default) exit $?"
tail -2 gpurun_out/$TAG/pytest_gpu_bf16x3.log
