# Wall accounting of the headline step: rocprofv3 kernel trace of the default (two-stream) bench command, reduced per stream by
# tools/wall_accounting.py; the trace database is kept (gpurun_out/<tag>/acct.db) so the table can be re-derived.
# usage: bash tools/gpu/acct.sh <tag> [extra bench.py arguments]
cd $GRAFT_REPO_ROOT
TAG=${1:-r06_a}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
ARGS="--steps 4 --warmup 2 --settle 0 --no-cpu-baseline --no-stock-baseline --no-profile --no-alt-math --no-other-configs $@"
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_acct -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/prof_acct.log 2>&1 )
DB=$(find $OUT/prof_acct -name "*.db" | head -1)
python tools/wall_accounting.py $DB > $OUT/wall_accounting.txt 2>&1
python tools/rocprof_summary.py $DB --steady 1 > $OUT/kernel_stats_steady.txt 2>&1
cp $DB $OUT/acct.db; rm -rf $OUT/prof_acct
ls -la $OUT/acct.db
cat $OUT/wall_accounting.txt
