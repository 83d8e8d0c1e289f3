cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_ab.log 2>&1; tail -1 gpurun_out/bench_ab.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run UP_BNB_ROWS=128
run UP_BNB_ROWS=64
run UP_BNB_ROWS=256
run UP_GRID_CAP=2048
run UP_GRID_CAP=8192
run UP_GRID_CAP=16384
done
