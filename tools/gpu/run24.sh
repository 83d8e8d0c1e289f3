cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/gpu/igemm_probe > gpurun_out/probe_split.log 2>&1; echo "probe exit $?"
grep -v "first blocks\|XCD finish\|timeline\|CU residency" gpurun_out/probe_split.log
UP_WGRAD_WORKGROUPS=512 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_split.log 2>&1
tail -1 gpurun_out/bench_split.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([ (r['kernel'], r['launches'], round(r['avg_ms'],4), round(r['tflops'],1)) for r in d['roofline']['by_kernel']][:6])"
UP_TAIL_SPLIT=0 UP_WGRAD_WORKGROUPS=512 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_nosplit.log 2>&1
tail -1 gpurun_out/bench_nosplit.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('nosplit', d['value'], d['ms_per_step'])"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x > gpurun_out/pytest_split.log 2>&1; echo "pytest exit $?"
tail -3 gpurun_out/pytest_split.log
