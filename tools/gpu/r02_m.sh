# after folding the split-K reduce into the weight-gradient kernels: operator parity + step times
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_m; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_bf16s_gpu.py tests/test_zzz_zero_skipping.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest.log
timeout 300 python tools/gpu/tune_ab.py --rounds 3 --steps 5 base > $OUT/ab.log 2>&1; tail -1 $OUT/ab.log
UNIPOSE_SYNC_WGRAD=1 timeout 300 python tools/gpu/tune_ab.py --rounds 2 --steps 5 base > $OUT/ab_sync.log 2>&1; tail -1 $OUT/ab_sync.log
timeout 300 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 > $OUT/steps736.log 2>&1; tail -1 $OUT/steps736.log
