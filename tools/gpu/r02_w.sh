cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_w; mkdir -p $OUT
timeout 400 python tools/gpu/small_batch.py split_per_cu 1 2 3 4 2>&1 | tee $OUT/small_batch.txt | tail -24
for v in 1 3 4; do
  echo "== lstm UP_SPLIT_PER_CU=$v"
  UP_SPLIT_PER_CU=$v timeout 300 python bench.py --model lstm --num-classes 13 --batch 8 --frames 5 --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done | tee $OUT/lstm.txt
