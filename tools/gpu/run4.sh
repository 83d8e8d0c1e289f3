cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
UP_PROFILE_CSV=gpurun_out/launches.csv timeout 300 python bench.py --steps 2 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r4.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_r4.log | cut -c1-400
