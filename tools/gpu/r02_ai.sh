cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_model_gpu.py -m gpu -q -x -s --timeout 600 -k "deferred or lstm or g4 or g11" 2>&1 | grep -E "deferred|passed|failed|Error" | tail -6
timeout 300 python bench.py --model lstm --num-classes 13 --batch 8 --frames 5 --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lstm', d['ms_per_step'], d['value'])"
