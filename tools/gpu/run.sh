# One parametrised runner for the GPU box:  bash tools/gpu/run.sh <tag> <action> [<action> ...]
# Outputs land in gpurun_out/<tag>/.  Actions:
#   tr16         lane map of ds_read_b64_tr_b16 (tools/gpu/tr16_probe)
#   tests_bf16s  tests/test_bf16s_gpu.py           tests_all   the whole -m gpu suite
#   ab736        736^2 B=16 bf16-storage step, UP_GLDS=1 vs 0 (two alternations)
#   csv736       per-launch CSV (exclusive stream mode) of the same step for UP_GLDS=1 and 0, grouped by GEMM shape
#   abenv/csvenv VARIANTS="A=1;A=2 B=3": the 736^2 step / its per-launch CSV under each environment
#   prof736      rocprofv3 kernel stats of the 736^2 step, both stream modes
#   ab368        default fp32 step, two runs (box sanity)
#   lstm         UniPose-LSTM leg with host / wall split (tools/gpu/steps.py)
#   abenv368 / csvenv368   the same A/B forms on the headline fp32 step (368^2, B = 32);  tests_glds32  tests/test_glds32_gpu.py
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
B736="--size 736 --batch 16 --math bf16s --no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs"
B368="--no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs"
line() { tail -1 $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])
except Exception as e: print('$2', 'no json line', e)"; }
for act in "$@"; do
case $act in
tr16) ./tools/gpu/tr16_probe > $OUT/tr16.txt 2>&1; echo "tr16 exit $?"; head -20 $OUT/tr16.txt ;;
tests_bf16s) timeout 900 python -m pytest tests/test_bf16s_gpu.py -m gpu -q --timeout 600 > $OUT/pytest_bf16s.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_bf16s.log ;;
tests_all) timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=40 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -${TAILN:-15} $OUT/pytest_gpu.log ;;
ab736) for rep in 1 2; do for g in 1 0; do
  UP_GLDS=$g timeout 300 python bench.py $B736 --steps 8 --warmup 3 --no-profile > $OUT/ab736_$g.log 2>&1; line $OUT/ab736_$g.log "glds=$g"; done; done ;;
csv736) for g in 1 0; do
  UP_GLDS=$g UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/$OUT/launches736_g$g.csv timeout 300 python bench.py $B736 --steps 3 --warmup 2 > $OUT/csv736_$g.log 2>&1; line $OUT/csv736_$g.log "csv glds=$g"; done
  ls $OUT/*.csv*; A=$(ls $OUT/launches736_g1.csv* | tail -1); B=$(ls $OUT/launches736_g0.csv* | tail -1)
  python tools/gpu/csv_compare.py $A $B > $OUT/csv736_compare.txt 2>&1; head -60 $OUT/csv736_compare.txt ;;
abenv) # VARIANTS="UP_GLDS=0;UP_GLDS=1 UP_GLDS_KT=32;..."  (736^2 bf16-storage step per variant, REPS alternations, default 2)
  IFS=';' read -ra VS <<< "$VARIANTS"
  for rep in $(seq 1 ${REPS:-2}); do for v in "${VS[@]}"; do
  env $v timeout 300 python bench.py $B736 --steps 8 --warmup 3 --no-profile > $OUT/abenv.log 2>&1; line $OUT/abenv.log "$v"; done; done ;;
csvenv) # per-launch CSVs (exclusive stream mode) for each of VARIANTS, each compared with the first
  IFS=';' read -ra VS <<< "$VARIANTS"; i=0
  for v in "${VS[@]}"; do
  env $v UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/$OUT/launches_v$i.csv timeout 300 python bench.py $B736 --steps 3 --warmup 2 > $OUT/csvenv_$i.log 2>&1; line $OUT/csvenv_$i.log "csv $v"
  if [ $i -gt 0 ]; then echo "== variant $i ($v) vs variant 0 (${VS[0]})"; python tools/gpu/csv_compare.py $(ls $OUT/launches_v$i.csv* | tail -1) $(ls $OUT/launches_v0.csv* | tail -1) > $OUT/csv_compare_$i.txt 2>&1; head -${HEAD:-25} $OUT/csv_compare_$i.txt; fi
  i=$((i+1)); done ;;
prof736) # rocprofv3 kernel stats of the 736^2 step: default (two streams) and exclusive (UNIPOSE_SYNC_WGRAD=1)
  ARGS="$B736 --steps 3 --warmup 1 --no-profile"
  ( cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_736 -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_736.log 2>&1
    UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_736x -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_736x.log 2>&1 )
  python tools/rocprof_summary.py $(find $OUT/prof_736 -name "*.db" | head -1) 4 > $OUT/kernel_stats_736_bf16s.txt 2>&1
  python tools/rocprof_summary.py $(find $OUT/prof_736x -name "*.db" | head -1) 4 > $OUT/kernel_stats_736_bf16s_exclusive.txt 2>&1
  find $OUT -name "*.db" -delete; head -${HEAD:-30} $OUT/kernel_stats_736_bf16s_exclusive.txt ;;
abenv368) # VARIANTS="UP_GLDS32=0;UP_GLDS32=1 UP_GLDS32_EPI=0;..."  (the headline fp32 step per variant, REPS alternations, default 2)
  IFS=';' read -ra VS <<< "$VARIANTS"
  for rep in $(seq 1 ${REPS:-2}); do for v in "${VS[@]}"; do
  env $v timeout 300 python bench.py $B368 --steps ${STEPS:-10} --warmup 3 --no-profile > $OUT/abenv368.log 2>&1; line $OUT/abenv368.log "$v"; done; done ;;
csvenv368) # per-launch CSVs (exclusive stream mode) of the fp32 step for each of VARIANTS, each compared with the first
  IFS=';' read -ra VS <<< "$VARIANTS"; i=0
  for v in "${VS[@]}"; do
  env $v UNIPOSE_SYNC_WGRAD=1 UP_PROFILE_CSV=$GRAFT_REPO_ROOT/$OUT/launches368_v$i.csv timeout 300 python bench.py $B368 --steps 3 --warmup 2 > $OUT/csvenv368_$i.log 2>&1; line $OUT/csvenv368_$i.log "csv $v"
  if [ $i -gt 0 ]; then echo "== variant $i ($v) vs variant 0 (${VS[0]})"; python tools/gpu/csv_compare.py $(ls $OUT/launches368_v$i.csv* | tail -1) $(ls $OUT/launches368_v0.csv* | tail -1) > $OUT/csv368_compare_$i.txt 2>&1; head -${HEAD:-40} $OUT/csv368_compare_$i.txt; fi
  i=$((i+1)); done ;;
tests_glds32) timeout 600 python -m pytest tests/test_glds32_gpu.py -m gpu -q -x --timeout 300 > $OUT/pytest_glds32.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_glds32.log ;;
prof368) # rocprofv3 kernel stats of the headline fp32 step: default (two streams) and exclusive (UNIPOSE_SYNC_WGRAD=1)
  ARGS="$B368 --steps 3 --warmup 1 --no-profile"
  ( cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_368 -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_368.log 2>&1
    UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_368x -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_368x.log 2>&1 )
  python tools/rocprof_summary.py $(find $OUT/prof_368 -name "*.db" | head -1) 4 > $OUT/kernel_stats.txt 2>&1
  python tools/rocprof_summary.py $(find $OUT/prof_368x -name "*.db" | head -1) 4 > $OUT/kernel_stats_exclusive.txt 2>&1
  find $OUT -name "*.db" -delete; head -${HEAD:-30} $OUT/kernel_stats_exclusive.txt ;;
proflstm) # rocprofv3 kernel stats of the UniPose-LSTM step (K = 13, B = 8, T = 5), both stream modes
  ARGS="--model lstm --batch 8 --num-classes 13 $B368 --steps 3 --warmup 1 --no-profile"
  ( cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_lstm -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_lstm.log 2>&1
    UNIPOSE_SYNC_WGRAD=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_lstmx -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $GRAFT_REPO_ROOT/$OUT/prof_lstmx.log 2>&1 )
  python tools/rocprof_summary.py $(find $OUT/prof_lstm -name "*.db" | head -1) 4 > $OUT/kernel_stats_lstm.txt 2>&1
  python tools/rocprof_summary.py $(find $OUT/prof_lstmx -name "*.db" | head -1) 4 > $OUT/kernel_stats_lstm_exclusive.txt 2>&1
  find $OUT -name "*.db" -delete; tail -2 $OUT/prof_lstm.log | cut -c1-300; head -${HEAD:-30} $OUT/kernel_stats_lstm_exclusive.txt ;;
ab368) for rep in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs --no-profile > $OUT/ab368.log 2>&1; line $OUT/ab368.log "fp32"; done ;;
lstm) timeout 300 python tools/gpu/steps.py --model lstm --batch 8 > $OUT/lstm_steps.log 2>&1; tail -5 $OUT/lstm_steps.log ;;
*) echo "unknown action $act" ;;
esac
done
