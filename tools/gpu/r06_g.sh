cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_g; mkdir -p $OUT
B368="--no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs --no-profile"
B736="--size 736 --batch 16 --math bf16s $B368"
line() { tail -1 $1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'])
except Exception as e: print('$2', 'no json line', e)"; }
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -k "graphed_train or fold or hooked or any_num_classes or bias or conv_fwd_bwd" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest.log
python tools/gpu/cu_mask.py 64 1 > $OUT/cu_mask.txt 2>&1; python tools/gpu/cu_mask.py 64 4 >> $OUT/cu_mask.txt 2>&1; python tools/gpu/cu_mask.py 128 2 >> $OUT/cu_mask.txt 2>&1; cat $OUT/cu_mask.txt
for rep in 1 2; do
for v in "A=0" "UNIPOSE_SIDE_CUS=64" "UNIPOSE_SIDE_CUS=96" "UNIPOSE_SIDE_CUS=128" "UNIPOSE_SIDE_CUS=160" "UNIPOSE_SIDE_CUS=192" "UNIPOSE_SIDE_CUS=128 UNIPOSE_SIDE_CU_STRIDE=2"; do
  env $v timeout 300 python bench.py $B368 --steps 10 --warmup 3 --settle 10 > $OUT/ab.log 2>&1; line $OUT/ab.log "fp32 $v"; done; done | tee $OUT/cu_mask_ab.txt
for rep in 1 2; do
for v in "A=0" "--graph"; do
  timeout 300 python bench.py $B368 --steps 10 --warmup 3 --settle 10 ${v/A=0/} > $OUT/ab.log 2>&1; line $OUT/ab.log "fp32 $v"
  timeout 300 python bench.py $B736 --steps 10 --warmup 3 --settle 10 ${v/A=0/} > $OUT/ab.log 2>&1; line $OUT/ab.log "bf16s736 B16 $v"
  timeout 300 python bench.py $B736 --batch 8 --steps 10 --warmup 3 --settle 10 ${v/A=0/} > $OUT/ab.log 2>&1; line $OUT/ab.log "bf16s736 B8 $v"
done; done | tee $OUT/graph_ab.txt
tail -5 $OUT/ab.log | cut -c1-300
for rep in 1 2; do
for v in "A=0" "UNIPOSE_BNRED_MIN_K_BF16=384" "UNIPOSE_BNRED_MIN_K_BF16=1100" "UNIPOSE_SYNC_WGRAD=1"; do
  env $v timeout 300 python bench.py $B736 --steps 10 --warmup 3 --settle 10 > $OUT/ab.log 2>&1; line $OUT/ab.log "bf16s736 $v"; done; done | tee $OUT/bnred_ab.txt
