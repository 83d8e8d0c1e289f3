# A/B of environment knobs inside one session:  bash tools/gpu/knob.sh "KNOB=a" "KNOB=b" ...   (three alternations)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3; do
for kv in "$@"; do
env $kv timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_ab.log 2>&1
tail -1 gpurun_out/bench_ab.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$kv', d['value'], d['ms_per_step'])"
done
done
