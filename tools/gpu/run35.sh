cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for v in db sb; do
UP_WGRAD_LOOP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_w$v.log 2>&1
tail -1 gpurun_out/bench_w$v.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
done
