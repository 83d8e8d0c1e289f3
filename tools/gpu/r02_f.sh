#!/bin/bash
# round 2, call f: fp32 step with the coalesced BatchNorm finalize kernels; A/B of 64x64 weight-gradient tiles
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_f
mkdir -p $OUT
timeout 200 python tools/gpu/tune_ab.py --rounds 3 --steps 5 base wgrad_tile=64 wgrad_tile=64+wgrad_per_cu=3 base > $OUT/tune_ab.log 2>&1; tail -1 $OUT/tune_ab.log
timeout 120 python -m pytest tests/test_bf16s_gpu.py -m gpu -q -s -k small_ops > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log | cut -c1-300
export TMPDIR=/tmp
cd /tmp && UNIPOSE_SYNC_WGRAD=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $OUT/prof -name "*.db" | head -1) 4 > $OUT/kernel_stats_exclusive.txt 2>&1
find $OUT -name "*.db" -delete
head -30 $OUT/kernel_stats_exclusive.txt | cut -c1-140
