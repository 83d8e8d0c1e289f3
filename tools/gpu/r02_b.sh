#!/bin/bash
# round 2, call b: the new config-level tests + the bench line with its other_configs legs.  Results under gpurun_out/r02_b/.
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_b
mkdir -p $OUT
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_checkpoint.py tests/test_zzz_zero_skipping.py -m gpu -q -s -x > $OUT/pytest_new.log 2>&1
tail -25 $OUT/pytest_new.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; head -c 3000 $OUT/bench.json
