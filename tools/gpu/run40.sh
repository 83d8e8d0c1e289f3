cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/gpu/igemm_probe wgrad > gpurun_out/probe_wgrad5.log 2>&1; echo "probe exit $?"
grep "double-buffered\|^wgrad" gpurun_out/probe_wgrad5.log
for rep in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_xcd.log 2>&1
tail -1 gpurun_out/bench_xcd.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu7.log 2>&1; echo "pytest exit $?"
tail -1 gpurun_out/pytest_gpu7.log
