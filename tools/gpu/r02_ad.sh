cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_ad; mkdir -p $OUT
timeout 600 python -m pytest tests/test_bf16s_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -1
timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
UNIPOSE_SYNC_WGRAD=1 timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
timeout 200 python tools/gpu/steps.py --size 368 --batch 32 --math bf16s --steps 8 2>&1 | tail -1
bash tools/gpu/pmc_sq.sh r02_ad_736 --size 736 --batch 16 --math bf16s 2>&1 | grep -E "igemm_bf16|wgrad_bf16" | cut -c1-320
