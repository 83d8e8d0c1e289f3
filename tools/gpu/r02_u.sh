cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_u; mkdir -p $OUT
timeout 600 python -m pytest tests/test_bf16s_gpu.py tests/test_configs_gpu.py -m gpu -q -x --timeout 600 -k "bf16 or 736" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest.log
for tw in 600 400 900; do
  echo "== UP_TILE_WANT_BF16=$tw"
  UP_TILE_WANT_BF16=$tw timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
done | tee $OUT/tile_want_bf16_736.txt
for tw in 600 1500; do
  echo "== bf16 operands at 368 UP_TILE_WANT_BF16=$tw"
  UP_TILE_WANT_BF16=$tw timeout 200 python tools/gpu/steps.py --size 368 --batch 32 --math bf16 --steps 8 2>&1 | tail -1
  UP_TILE_WANT_BF16=$tw timeout 200 python tools/gpu/steps.py --size 368 --batch 32 --math bf16s --steps 8 2>&1 | tail -1
done | tee -a $OUT/tile_want_bf16_736.txt
