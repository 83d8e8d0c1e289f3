cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
UNIPOSE_CONV_MATH=bf16x3 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 > gpurun_out/pytest_gpu_bf16x3.log 2>&1; echo "pytest(bf16x3 default) exit $?"
grep -E "AssertionError|passed|failed|Error" gpurun_out/pytest_gpu_bf16x3.log | tail -12 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_alt.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_alt.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('alt_math'))"
