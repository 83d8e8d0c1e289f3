cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 tools/gpu/igemm_probe > gpurun_out/probe_timeline.log 2>&1; echo "probe exit $?"
cat gpurun_out/probe_timeline.log
for t in 256 384 512 640; do
  UP_WGRAD_WORKGROUPS=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_wg$t.log 2>&1
  tail -1 gpurun_out/bench_wg$t.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('target $t', d['value'], d['ms_per_step'])"
done
