# Wall accounting of one workload from a rocprofv3 kernel trace:  bash tools/gpu/acct.sh <tag> <bench.py arguments>
# -> gpurun_out/<tag>/wall_accounting.txt (tools/wall_accounting.py) and kernel_stats_steady.txt (tools/rocprof_summary.py --steady)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/$TAG
ARGS="$* --steps 4 --warmup 2 --settle 0 --no-cpu-baseline --no-stock-baseline --no-profile --no-alt-math --no-other-configs"
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_acct -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_acct.log 2>&1 )
DB=$(find gpurun_out/$TAG/prof_acct -name "*.db" | head -1)
python tools/wall_accounting.py $DB > gpurun_out/$TAG/wall_accounting.txt 2>&1
python tools/rocprof_summary.py $DB --steady 1 > gpurun_out/$TAG/kernel_stats_steady.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
sed -n 1,25p gpurun_out/$TAG/wall_accounting.txt
