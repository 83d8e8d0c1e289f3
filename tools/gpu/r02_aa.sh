cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_aa; mkdir -p $OUT
timeout 600 python -m pytest tests/test_bf16s_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -1
timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
UNIPOSE_SYNC_WGRAD=1 timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
timeout 200 python tools/gpu/steps.py --size 368 --batch 32 --math bf16s --steps 8 2>&1 | tail -1
