cd $GRAFT_REPO_ROOT
timeout 300 python tools/gpu/steps.py --lstm --batch 8 --steps 6 2>&1 | tail -1
UNIPOSE_SYNC_WGRAD=1 timeout 300 python tools/gpu/steps.py --lstm --batch 8 --steps 6 2>&1 | tail -1
