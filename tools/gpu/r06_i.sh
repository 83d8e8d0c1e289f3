cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py -m gpu -q --timeout 600 -k "group or lstm or fold" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest.log
L="--model lstm --no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs --no-profile --steps 8 --warmup 3 --settle 5"
for rep in 1 2 3; do for v in "UP_BN_FOLD=1" "UP_BN_FOLD=0"; do
  env $v python bench.py $L 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lstm $v', d['value'], d['ms_per_step'])"
done; done | tee $OUT/lstm_fold_ab.txt
