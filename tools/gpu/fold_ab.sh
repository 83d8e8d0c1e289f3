# round 6: BatchNorm finalize folded into the producing launches — tests + whole-step A/B (UP_BN_FOLD=1 vs 0), three alternations
cd $GRAFT_REPO_ROOT
TAG=${1:-r06_c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x --timeout 600 -k "fold" > $OUT/pytest_fold.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_fold.log
VARIANTS="UP_BN_FOLD=1;UP_BN_FOLD=0" REPS=3 STEPS=20 bash tools/gpu/run.sh $TAG abenv368 2>&1 | tee $OUT/fold_ab.txt
