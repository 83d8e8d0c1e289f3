#!/bin/bash
# round 2, call e: bf16 STORAGE path on hardware: parity tests, then the 736x736 B=16 step (per-step times, rocprof kernel stats)
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_e
mkdir -p $OUT
timeout 600 python -m pytest tests/test_bf16s_gpu.py -m gpu -q -s > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log | cut -c1-220
grep -E "argmax agreement|cosine|held for" $OUT/pytest.log | cut -c1-250
timeout 120 python tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 8 2>&1 | tail -1
timeout 120 python tools/gpu/steps.py --size 736 --batch 16 --math bf16 --steps 6 2>&1 | tail -1
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/gpu/steps.py --size 736 --batch 16 --math bf16s --steps 4 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find $OUT/prof -name "*.db" | head -1) 4 > $OUT/kernel_stats_736_bf16s.txt 2>&1
find $OUT -name "*.db" -delete
head -24 $OUT/kernel_stats_736_bf16s.txt | cut -c1-150
