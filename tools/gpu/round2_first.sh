#!/bin/bash
# First GPU call of the next round (≈ 2 min of box time): validate the knob-gated kernel forms that were built after the
# round-1 GPU budget ran out, then A/B them inside one process.  Results under gpurun_out/r02_a/.
#   gpurun --timeout 300 -- 'bash tools/gpu/round2_first.sh'
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r02_a
mkdir -p $OUT
# 1. parity of the never-yet-on-hardware forms (tap_sort, wgrad_rect, occ64): ~20 s
timeout 120 python -m pytest tests/test_zzz_zero_skipping.py -m gpu -x -q > $OUT/pytest_zero_skipping.log 2>&1
tail -2 $OUT/pytest_zero_skipping.log
# 2. in-process A/B of every queued variant on the BASELINE configs[1] step (≈ 1 s per variant and round)
timeout 150 python tools/gpu/tune_ab.py --rounds 2 --steps 5 base tap_sort=1 wgrad_rect=1 tap_sort=1+wgrad_rect=1 occ64=7 occ64=8 \
    wgrad_single=1 main_hi=1 tap_sort=1+wgrad_rect=1+occ64=7+main_hi=1 > $OUT/tune_ab.log 2>&1
tail -1 $OUT/tune_ab.log
# 3. the north star's WASP figures with and without the tap-sorted rows
timeout 40 python bench.py --wasp-only > $OUT/wasp_default.json 2>/dev/null
UP_TAP_SORT=1 timeout 40 python bench.py --wasp-only > $OUT/wasp_tap_sort.json 2>/dev/null
cat $OUT/wasp_default.json $OUT/wasp_tap_sort.json
# 4. weight-gradient split plan in both stream modes (an unexplained 252-vs-504 workgroup difference in round 1's CSVs)
timeout 60 python tools/gpu/q_wgrad_live.py > $OUT/wgrad_plan.log 2>&1
tail -12 $OUT/wgrad_plan.log
