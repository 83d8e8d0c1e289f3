cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_ab.log 2>&1; tail -1 gpurun_out/bench_ab.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run UP_TILE_WANT=1000
run UP_TILE_WANT=700
run UP_TILE_WANT=1500
run UP_WGRAD_PER_CU=3
run UP_WGRAD_PER_CU=1
done
