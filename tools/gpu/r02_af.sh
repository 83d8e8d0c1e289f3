cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_af; mkdir -p $OUT
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_model_gpu.py -m gpu -q -x --timeout 600 -k "lstm or repeated or bptt or fence" 2>&1 | tail -3
for v in "" "UNIPOSE_NO_SIDE_ACCUMULATE=1"; do
  echo "== lstm $v"
  env $v timeout 300 python bench.py --model lstm --num-classes 13 --batch 8 --frames 5 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done | tee $OUT/lstm.txt
