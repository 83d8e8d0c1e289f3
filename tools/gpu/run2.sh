set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
lscpu | grep -E "Model name|^CPU\(s\)|Socket|NUMA node\(s\)" 
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_nocpu.log 2>&1; echo "bench exit $?"; tail -12 gpurun_out/bench_nocpu.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-profile --cpu-steps 2 > gpurun_out/bench_cpu.log 2>&1; echo "bench exit $?"; tail -8 gpurun_out/bench_cpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof exit $?"
tail -5 $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
ls -la $GRAFT_REPO_ROOT/gpurun_out/prof_r1/* | head
