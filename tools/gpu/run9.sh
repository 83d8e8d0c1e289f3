cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
grep -E "AssertionError|passed|failed|Error" gpurun_out/pytest_gpu.log | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r9.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_r9.log | cut -c1-330
python - <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from unipose_amd import ops
ops.ASYNC_WGRAD = False
sys.argv = ['bench.py', '--steps', '6', '--warmup', '2', '--no-cpu-baseline']
os.environ['UP_PROFILE_CSV'] = 'gpurun_out/launches_r9.csv'
import runpy
runpy.run_path('bench.py', run_name='__main__')
PY
