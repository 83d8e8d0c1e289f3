cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_l; mkdir -p $OUT
B368="--no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs --no-profile"
line() { tail -1 $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])
except Exception as e: print('$2', 'no json line', e)"; }
for rep in 1 2 3; do for v in "UNIPOSE_KEEP_WGRAD_INPUTS=1" "UNIPOSE_KEEP_WGRAD_INPUTS=0"; do
  env $v timeout 300 python bench.py $B368 --steps 20 --warmup 3 --settle 10 > $OUT/ab.log 2>&1; line $OUT/ab.log "fp32 $v"
  env $v timeout 300 python bench.py $B368 --size 736 --batch 16 --math bf16s --steps 10 --warmup 3 --settle 10 > $OUT/ab.log 2>&1; line $OUT/ab.log "bf16s736 $v"
  env $v timeout 300 python bench.py $B368 --model lstm --steps 8 --warmup 3 --settle 5 > $OUT/ab.log 2>&1; line $OUT/ab.log "lstm $v"
done; done | tee $OUT/keep_ab.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -18 $OUT/pytest_gpu.log
