cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_ab.log 2>&1; tail -1 gpurun_out/bench_ab.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run UP_SPLIT_MIN_SLICES=2
run UP_SPLIT_MIN_SLICES=4
run UP_SPLIT_MIN_SLICES=8
run UP_SPLIT_MAX_R_PCT=75
run UP_SPLIT_MAX_R_PCT=25
done
