cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/gpu/igemm_probe > gpurun_out/probe_batch.log 2>&1; echo "probe exit $?"
grep "production\|^[0-9]" gpurun_out/probe_batch.log
for rep in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_pb.log 2>&1
tail -1 gpurun_out/bench_pb.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
