cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_glds32_gpu.py -m gpu -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 600 python tools/gpu/shape_ab.py base breg=1 > $OUT/shape_ab_breg.txt 2>&1; tail -40 $OUT/shape_ab_breg.txt
VARIANTS="UP_BREG=0;UP_BREG=1" REPS=3 STEPS=20 bash tools/gpu/run.sh r06_j abenv368 2>&1 | tee $OUT/breg_ab.txt
