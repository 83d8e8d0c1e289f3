cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.log 2>&1; echo "bench exit $?"; tail -1 gpurun_out/bench_final.log | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1d -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/rocprof_d.log 2>&1; echo "rocprof exit $?"
