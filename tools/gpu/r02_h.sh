#!/bin/bash
# per-shape exclusive timings (per-launch HIP events, streams serialised) with and without the swizzled LDS rows
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_h
mkdir -p $OUT
for v in 0 1; do
UP_LDS_SWZ=$v UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/$OUT/launches_swz$v.csv timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-math --no-other-configs > $OUT/bench_swz$v.log 2>&1
python tools/gpu/csv_loss.py $OUT/launches_swz$v.csv 157.3 14 > $OUT/lost_swz$v.txt; head -16 $OUT/lost_swz$v.txt
done
