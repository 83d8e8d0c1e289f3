cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_h; mkdir -p $OUT
B368="--no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs --no-profile"
B736="--size 736 --batch 16 --math bf16s $B368"
line() { tail -1 $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])
except Exception as e: print('$2', 'no json line', e)"; }
timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -q --timeout 600 -k "graphed_train" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
for v in "UNIPOSE_SYNC_WGRAD=1" "UNIPOSE_SYNC_WGRAD=1 G=--graph"; do
  g=""; case "$v" in *G=--graph*) g="--graph";; esac
  env ${v/ G=--graph/} timeout 300 python bench.py $B368 --steps 10 --warmup 3 --settle 10 $g > $OUT/ab.log 2>&1; line $OUT/ab.log "fp32 $v"
  env ${v/ G=--graph/} timeout 300 python bench.py $B736 --batch 8 --steps 10 --warmup 3 --settle 10 $g > $OUT/ab.log 2>&1; line $OUT/ab.log "bf16s736 B8 $v"
  env ${v/ G=--graph/} timeout 300 python bench.py --size 368 --batch 32 --math bf16s $B368 --steps 10 --warmup 3 --settle 10 $g > $OUT/ab.log 2>&1; line $OUT/ab.log "bf16s368 B32 $v"
done | tee $OUT/graph_single_stream.txt
