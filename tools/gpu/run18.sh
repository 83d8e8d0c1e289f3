cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -k "bf16" > gpurun_out/pytest_bf16.log 2>&1; echo "pytest exit $?"
grep -E "AssertionError|passed|failed|Error" gpurun_out/pytest_bf16.log | tail -5
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --math bf16x3 > gpurun_out/bench_bf16x3.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_bf16x3.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([ (r['kernel'], r['launches'], round(r['avg_ms'],4), round(r['tflops'],1)) for r in d['roofline']['by_kernel']][:6])"
