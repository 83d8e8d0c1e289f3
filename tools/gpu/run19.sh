cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in 1024 512 768 2048; do
  UP_WGRAD_WORKGROUPS=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_wg$t.log 2>&1
  tail -1 gpurun_out/bench_wg$t.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('target $t', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_sync -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math > $GRAFT_REPO_ROOT/gpurun_out/prof_sync.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/prof_sync.log
python tools/rocprof_summary.py $(find gpurun_out/prof_sync -name "*.db" | head -1) 4 > gpurun_out/sync_kernel_stats.txt 2>&1
head -30 gpurun_out/sync_kernel_stats.txt
find gpurun_out/prof_sync -name "*.db" -delete
