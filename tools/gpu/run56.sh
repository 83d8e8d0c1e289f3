cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 100 tools/gpu/igemm_probe wasp > gpurun_out/probe_wasp2.log 2>&1; echo "probe exit $?"
cut -c1-200 gpurun_out/probe_wasp2.log
UP_TAP_SKIP=0 timeout 100 tools/gpu/igemm_probe wasp 2>&1 | cut -c1-80
run() { env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_ab.log 2>&1; tail -1 gpurun_out/bench_ab.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
run UP_TAP_SKIP=1
run UP_TAP_SKIP=0
done
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu11.log 2>&1; echo "pytest exit $?"
tail -1 gpurun_out/pytest_gpu11.log
