# 736^2 B=16 bf16-storage step: kernel stats with the two streams serialised (each kernel's own duration)
cd $GRAFT_REPO_ROOT
TAG=r02_p; mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --size 736 --batch 16 --math bf16s --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/$TAG/prof.log | cut -c1-300
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_736_bf16s_exclusive.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
head -32 gpurun_out/$TAG/kernel_stats_736_bf16s_exclusive.txt
UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=gpurun_out/$TAG/launches.csv timeout 300 python bench.py --size 736 --batch 16 --math bf16s --steps 2 --warmup 1 --no-cpu-baseline --no-alt-math --no-other-configs > gpurun_out/$TAG/bench.log 2>&1
tail -1 gpurun_out/$TAG/bench.log | cut -c1-200
