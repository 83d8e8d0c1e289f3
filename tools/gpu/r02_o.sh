# tap-sorted blocks dealt out over the XCDs: WASP leg + step time
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_o; mkdir -p $OUT
for v in "UP_TAP_SORT=1" "UP_TAP_SORT=0"; do
  echo "== $v"
  env $v timeout 120 python bench.py --wasp-only 2>/dev/null | tail -1 | python -c "
import json,sys
for w in json.loads(sys.stdin.read())['wasp_dilated']: print(w['dilation'], w['ms'], w['effective_mfma_frac'])"
done 2>&1 | tee $OUT/wasp_knobs.txt
timeout 300 python tools/gpu/tune_ab.py --rounds 3 --steps 5 base tap_sort=0 > $OUT/ab.log 2>&1; tail -1 $OUT/ab.log
timeout 300 python -m pytest tests/test_zzz_zero_skipping.py -m gpu -q -x --timeout 300 2>&1 | tail -1
