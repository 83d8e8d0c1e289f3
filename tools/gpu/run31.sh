cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_link.log 2>&1
tail -1 gpurun_out/bench_link.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([ (r['kernel'], r['launches'], round(r['avg_ms'],4), round(r['tflops'],1)) for r in d['roofline']['by_kernel']][:8])"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest exit $?"
tail -3 gpurun_out/pytest_gpu4.log
