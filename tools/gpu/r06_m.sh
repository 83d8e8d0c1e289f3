cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_glds32_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 -k "glds32 or fold or group or fused or bnred or g11 or g4" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
bash tools/gpu/run.sh r06_m ab368
B368="--no-cpu-baseline --no-stock-baseline --no-alt-math --no-other-configs --no-profile"
for i in 1 2; do python bench.py $B368 --model lstm --steps 8 --warmup 3 --settle 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lstm', d['value'], d['ms_per_step'])"; done
bash tools/gpu/acct.sh r06_m > /dev/null; sed -n 18,30p gpurun_out/r06_m/wall_accounting.txt
