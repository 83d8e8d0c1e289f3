cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/gpurun_out/launches_final.csv timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-alt-math > gpurun_out/bench_csv.log 2>&1; echo "exit $?"
wc -l gpurun_out/launches_final.csv
