cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" 1; do
UP_MAIN_PRIO=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_prio$v.log 2>&1
tail -1 gpurun_out/bench_prio$v.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('prio=$v', d['value'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest exit $?"
tail -2 gpurun_out/pytest_gpu3.log
