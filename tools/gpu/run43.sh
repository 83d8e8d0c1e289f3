cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --batch 2 --size 64 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_tiny.log 2>&1
tail -1 gpurun_out/bench_tiny.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('tiny (CPU-bound floor)', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --batch 8 --size 368 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_b8.log 2>&1
tail -1 gpurun_out/bench_b8.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('B=8', d['value'], d['ms_per_step'])"
