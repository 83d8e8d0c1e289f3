#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_g
mkdir -p $OUT
timeout 60 python tools/gpu/dbg_maxpool.py 2>&1 | tail -10
timeout 200 python tools/gpu/tune_ab.py --rounds 3 --steps 5 base tile_want=1000 > $OUT/tune_ab.log 2>&1; tail -1 $OUT/tune_ab.log
export TMPDIR=/tmp
cd /tmp && UNIPOSE_SYNC_WGRAD=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $OUT/prof -name "*.db" | head -1) 4 > $OUT/kernel_stats_exclusive.txt 2>&1
find $OUT -name "*.db" -delete
grep -E "finalize|total kernel" $OUT/kernel_stats_exclusive.txt | cut -c1-140
