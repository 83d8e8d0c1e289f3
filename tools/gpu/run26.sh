cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_pack.log 2>&1
tail -1 gpurun_out/bench_pack.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print([ (r['kernel'], r['launches'], round(r['avg_ms'],4), round(r['tflops'],1)) for r in d['roofline']['by_kernel']][:8])"
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest exit $?"
tail -3 gpurun_out/pytest_gpu2.log
cd /tmp && export TMPDIR=/tmp
UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/gpurun_out/launches_sync.csv timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sync2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math > $GRAFT_REPO_ROOT/gpurun_out/prof_sync2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find gpurun_out/prof_sync2 -name "*.db" | head -1) 4 > gpurun_out/sync2_kernel_stats.txt 2>&1
head -32 gpurun_out/sync2_kernel_stats.txt
find gpurun_out/prof_sync2 -name "*.db" -delete
