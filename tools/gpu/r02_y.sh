cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_y; mkdir -p $OUT
timeout 600 python -m pytest tests/test_bf16s_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -1 $OUT/pytest.log
timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
timeout 200 python tools/gpu/steps.py --size 368 --batch 32 --math bf16s --steps 8 2>&1 | tail -1
timeout 400 python tools/gpu/infer_latency.py 2>&1 | tee $OUT/infer_latency.txt | tail -14
