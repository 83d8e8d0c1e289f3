cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_t; mkdir -p $OUT
for tw in 1500 1000 700 500 300; do
  echo "== UP_TILE_WANT=$tw"
  UP_TILE_WANT=$tw timeout 200 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
done | tee $OUT/tile_want_736.txt
