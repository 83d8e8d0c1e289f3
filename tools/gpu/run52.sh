cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
for k in 256 1024; do
UP_DB_MIN_K=$k timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_dbk$k.log 2>&1
tail -1 gpurun_out/bench_dbk$k.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('min_k=$k', d['value'], d['ms_per_step'])"
done
done
