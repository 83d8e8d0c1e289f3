cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_red.log 2>&1
tail -1 gpurun_out/bench_red.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu8.log 2>&1; echo "pytest exit $?"
tail -1 gpurun_out/pytest_gpu8.log
cd /tmp && export TMPDIR=/tmp
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sync4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math > $GRAFT_REPO_ROOT/gpurun_out/prof_sync4.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find gpurun_out/prof_sync4 -name "*.db" | head -1) 4 2>&1 | grep "bn_\|wgrad_reduce\|total kernel"
find gpurun_out/prof_sync4 -name "*.db" -delete
