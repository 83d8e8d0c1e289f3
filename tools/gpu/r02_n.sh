# WASP dilated leg under the tap knobs
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_n; mkdir -p $OUT
for v in "UP_TAP_SORT=1" "UP_TAP_SORT=0" "UP_TAP_SKIP=0" "UP_TAP_SORT=1 UP_TAIL_SPLIT=0" "UP_TAP_SORT=1 UP_LDS_SWZ=0" "UP_TAP_SORT=1 UP_DB_MIN_K=100000"; do
  echo "== $v"
  env $v timeout 120 python bench.py --wasp-only 2>/dev/null | tail -1 | python -c "
import json,sys
for w in json.loads(sys.stdin.read())['wasp_dilated']: print(w['dilation'], w['ms'], w['effective_mfma_frac'])"
done 2>&1 | tee $OUT/wasp_knobs.txt
