cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" 1 "" 1; do
UP_ADAM_FOREACH=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_adam$v.log 2>&1
tail -1 gpurun_out/bench_adam$v.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('foreach=$v', d['value'], d['ms_per_step'])"
done
