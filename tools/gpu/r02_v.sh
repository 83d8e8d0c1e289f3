cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_v; mkdir -p $OUT
for v in "UP_WGRAD_PER_CU=2" "UP_WGRAD_PER_CU=3" "UP_WGRAD_PER_CU=4" "UP_SHORT_K=1" "UP_SHORT_K=1100" "UNIPOSE_SYNC_WGRAD=1"; do
  echo "== $v"
  env $v timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
done | tee $OUT/knobs_736.txt
