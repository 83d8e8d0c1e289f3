# Round-end measurement set: smoke, default bench line (cpu_baseline, stock_gpu_baseline, alt_math, other_configs), rocprofv3 kernel
# stats of the bench command in both stream modes, PMC passes; with TESTS=1 also the full -m gpu suite (default + bf16x3 arithmetic);
# LIGHT=1 stops after the kernel statistics.
cd $GRAFT_REPO_ROOT
TAG=${1:-r06_zz}
mkdir -p gpurun_out/$TAG
# the measured binary is the tree's: the stamp next to the shipped library names the sha256 of the sources it was built from
python -c "
from unipose_amd.build import library_is_current, source_hash
print('libunipose_hip.so built from this tree:', library_is_current(), 'sources sha256', source_hash())" | tee gpurun_out/$TAG/build_identity.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/$TAG/smoke.log
timeout 900 python bench.py > gpurun_out/$TAG/bench.log 2> gpurun_out/$TAG/bench.err; echo "bench exit $?"
tail -1 gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench.json"))
r=d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], r["kernel"], r["achieved"], r["frac"], "excl", r.get("exclusive",{}).get("frac"),
      "alt", d.get("alt_math",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), "stock", d.get("stock_gpu_baseline",{}).get("value"))
for o in d.get("other_configs", []): print("other", o["value"], o["ms_per_step"], o["roofline"]["kernel"], o["roofline"]["achieved"])
for w in r.get("wasp_dilated", []): print("wasp", w["dilation"], w["ms"], w["effective_mfma_frac"])
PY
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --settle 0 --no-cpu-baseline --no-stock-baseline --no-profile --no-alt-math --no-other-configs"
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_sync -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_sync.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_736 -o bench -- python $GRAFT_REPO_ROOT/bench.py --size 736 --batch 16 --math bf16s $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_736.log 2>&1
UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_736x -o bench -- python $GRAFT_REPO_ROOT/bench.py --size 736 --batch 16 --math bf16s $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_736x.log 2>&1
[ -n "$LIGHT" ] || timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_lstm -o bench -- python $GRAFT_REPO_ROOT/bench.py --model lstm $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_lstm.log 2>&1
[ -n "$LIGHT" ] || UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_lstmx -o bench -- python $GRAFT_REPO_ROOT/bench.py --model lstm $ARGS > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof_lstmx.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats.txt 2>&1
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_sync -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_exclusive.txt 2>&1
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_736 -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_736_bf16s.txt 2>&1
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_736x -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_736_bf16s_exclusive.txt 2>&1
[ -n "$LIGHT" ] || python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_lstm -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_lstm.txt 2>&1
[ -n "$LIGHT" ] || python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_lstmx -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_lstm_exclusive.txt 2>&1
# steady-state windows only (between two launches of the once-per-step loss kernel): what a step launches once tables, packed
# weights and the allocator have settled — the first step's ~260 fills / ~150 copies stay out
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_sync -name "*.db" | head -1) --steady 1 > gpurun_out/$TAG/kernel_stats_exclusive_steady.txt 2>&1
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof_736x -name "*.db" | head -1) --steady 1 > gpurun_out/$TAG/kernel_stats_736_bf16s_exclusive_steady.txt 2>&1
python tools/wall_accounting.py $(find gpurun_out/$TAG/prof -name "*.db" | head -1) > gpurun_out/$TAG/wall_accounting.txt 2>&1
python tools/wall_accounting.py $(find gpurun_out/$TAG/prof_736 -name "*.db" | head -1) > gpurun_out/$TAG/wall_accounting_736_bf16s.txt 2>&1
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof -name "*.db" | head -1) --steady 1 > gpurun_out/$TAG/kernel_stats_steady.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
python tools/gpu/skew.py --steps 10 > gpurun_out/$TAG/skew_fp32.txt 2>&1
# what the exact-fp32 MFMA sustains on this box, alone and next to the K loop's other instructions (round 5)
head -3 gpurun_out/$TAG/kernel_stats_exclusive_steady.txt
head -8 gpurun_out/$TAG/kernel_stats.txt
head -8 gpurun_out/$TAG/kernel_stats_exclusive.txt
head -8 gpurun_out/$TAG/kernel_stats_736_bf16s.txt
# LIGHT=1: bench line + kernel statistics only (the per-shape tables and the counter passes of the round stay those of the last full run)
[ -n "$LIGHT" ] && exit 0
UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/gpurun_out/$TAG/launches.csv timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs > gpurun_out/$TAG/bench_csv.log 2>&1
python tools/gpu/csv_loss.py gpurun_out/$TAG/launches.csv.1 157.3 20 > gpurun_out/$TAG/lost_time_by_shape.txt 2>&1 || python tools/gpu/csv_loss.py gpurun_out/$TAG/launches.csv 157.3 20 > gpurun_out/$TAG/lost_time_by_shape.txt 2>&1
head -4 gpurun_out/$TAG/lost_time_by_shape.txt
bash tools/gpu/pmc.sh $TAG
# whole-step A/B of the switches that carry the round's design (fp32 headline step, alternating, two rounds): BatchNorm finalize folded
# into the producers (0 = the stand-alone arrive kernels), fused BatchNorm-backward reduction, hybrid operand path, async re-pack
VARIANTS="A=default;UP_BN_FOLD=0;UNIPOSE_BN_FUSE_REDUCE=0;UP_BREG=1;UNIPOSE_ASYNC_REPACK=0" REPS=2 bash tools/gpu/run.sh $TAG abenv368 > gpurun_out/$TAG/knob_ab.txt 2>&1; cat gpurun_out/$TAG/knob_ab.txt
UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/gpurun_out/$TAG/launches_736.csv timeout 300 python bench.py --size 736 --batch 16 --math bf16s --steps 4 --warmup 2 --no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs > gpurun_out/$TAG/bench_csv_736.log 2>&1
python tools/gpu/csv_loss.py $(ls gpurun_out/$TAG/launches_736.csv* | tail -1) 2500 24 > gpurun_out/$TAG/lost_time_by_shape_736.txt 2>&1; head -3 gpurun_out/$TAG/lost_time_by_shape_736.txt
bash tools/gpu/pmc_sq.sh ${TAG}_736 --size 736 --batch 16 --math bf16s
# the same A/B on the 736^2 bf16-storage step: third-generation tiles (bf16s_big.h) off / on / with two LDS stages / forward only
VARIANTS="UP_GLDS_BIG=0;A=default;UP_BIG_STAGES=2;UP_BIG_DGRAD=0" REPS=2 bash tools/gpu/run.sh $TAG abenv > gpurun_out/$TAG/knob_ab_736.txt 2>&1; cat gpurun_out/$TAG/knob_ab_736.txt
if [ -n "$TESTS" ]; then
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --durations=15 > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -2 gpurun_out/$TAG/pytest_gpu.log
UNIPOSE_CONV_MATH=bf16x3 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 > gpurun_out/$TAG/pytest_gpu_bf16x3.log 2>&1; echo "pytest(bf16x3 default) exit $?"
tail -2 gpurun_out/$TAG/pytest_gpu_bf16x3.log
fi
