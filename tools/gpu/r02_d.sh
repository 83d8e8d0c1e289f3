#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_d
mkdir -p $OUT
timeout 120 python tools/gpu/steps.py --size 736 --batch 16 --math bf16 --steps 8 2>&1 | tail -1
timeout 120 python tools/gpu/steps.py --size 736 --batch 16 --math bf16 --steps 6 --sync-wgrad 2>&1 | tail -1
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/gpu/steps.py --size 736 --batch 16 --math bf16 --steps 4 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $OUT/prof -name "*.db" | head -1) 4 > $OUT/kernel_stats_736_bf16.txt 2>&1
find $OUT -name "*.db" -delete
head -22 $OUT/kernel_stats_736_bf16.txt | cut -c1-150
timeout 400 python -m pytest tests/test_configs_gpu.py -m gpu -q -s -k "lstm or rccl" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
