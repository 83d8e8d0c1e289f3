#!/bin/bash
# round 2, call c: remaining new tests (LSTM full size, backward-exception recovery, 1-rank RCCL, folded inference, zero-skipping forms),
# bf16 weight-gradient kernel parity + the 736x736 bf16 leg.  Results under gpurun_out/r02_c/.
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_c
mkdir -p $OUT
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_checkpoint.py tests/test_zzz_zero_skipping.py "tests/test_ops_gpu.py::test_conv_bf16_operand_kernels" tests/test_model_gpu.py -m gpu -q -s > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-alt-math --steps 6 > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_c/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for o in d.get('other_configs', []):
    print(o['value'], o['ms_per_step'], o['roofline']['kernel'], o['roofline']['achieved'])
    for r in o['roofline']['by_kernel']: print('   ', r['kernel'], r['launches'], round(r['total_ms'],2), round(r['tflops'],1))
PY
