# A/B of two builds of libunipose_hip.so inside one session: tools/gpu/ab/lib_old.so vs lib_new.so
# (ABARGS="--size 736 --batch 16 --math bf16s" selects another workload)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { cp tools/gpu/ab/lib_$1.so unipose_amd/libunipose_hip.so; timeout 300 python bench.py ${ABARGS:-} --steps ${STEPS:-15} --warmup 3 --no-cpu-baseline --no-stock-baseline --no-other-configs --no-alt-math --no-profile > gpurun_out/bench_ab.log 2>&1; tail -1 gpurun_out/bench_ab.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
run old
run new
done
cp tools/gpu/ab/lib_new.so unipose_amd/libunipose_hip.so
