# The second half of final.sh on its own (after a LIGHT=1 run): per-shape lost-time tables, PMC passes, SQ counters of the 736^2 step.
cd $GRAFT_REPO_ROOT
TAG=${1:-r03_z}
mkdir -p gpurun_out/$TAG
UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/gpurun_out/$TAG/launches.csv timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs > gpurun_out/$TAG/bench_csv.log 2>&1
python tools/gpu/csv_loss.py gpurun_out/$TAG/launches.csv.1 157.3 20 > gpurun_out/$TAG/lost_time_by_shape.txt 2>&1 || python tools/gpu/csv_loss.py gpurun_out/$TAG/launches.csv 157.3 20 > gpurun_out/$TAG/lost_time_by_shape.txt 2>&1
head -4 gpurun_out/$TAG/lost_time_by_shape.txt
bash tools/gpu/pmc.sh $TAG
UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/gpurun_out/$TAG/launches_736.csv timeout 300 python bench.py --size 736 --batch 16 --math bf16s --steps 4 --warmup 2 --no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs > gpurun_out/$TAG/bench_csv_736.log 2>&1
python tools/gpu/csv_loss.py $(ls gpurun_out/$TAG/launches_736.csv* | tail -1) 2500 24 > gpurun_out/$TAG/lost_time_by_shape_736.txt 2>&1; head -3 gpurun_out/$TAG/lost_time_by_shape_736.txt
[ -n "$NOSQ" ] || bash tools/gpu/pmc_sq.sh ${TAG}_736 --size 736 --batch 16 --math bf16s
