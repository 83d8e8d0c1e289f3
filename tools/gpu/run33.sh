cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/gpu/igemm_probe wgrad > gpurun_out/probe_wgrad2.log 2>&1; echo "probe exit $?"
grep -v "first blocks\|XCD finish" gpurun_out/probe_wgrad2.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_wdb.log 2>&1
tail -1 gpurun_out/bench_wdb.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([ (r['kernel'], r['launches'], round(r['avg_ms'],4), round(r['tflops'],1)) for r in d['roofline']['by_kernel']][:8]); print(d['roofline'].get('exclusive'))"
