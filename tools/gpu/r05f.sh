# round 5, GPU call F: async re-pack A/B + tests of the stale-weights fix
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py "tests/test_configs_gpu.py::test_g16_adam_trajectory_vs_reference_golden" tests/test_bf16s_gpu.py -m gpu -q --timeout 600 > $OUT/pytest_new.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_new.log
VARIANTS="UNIPOSE_ASYNC_REPACK=1;UNIPOSE_ASYNC_REPACK=0" REPS=3 STEPS=20 bash tools/gpu/run.sh r05_f abenv368
VARIANTS="UNIPOSE_ASYNC_REPACK=1;UNIPOSE_ASYNC_REPACK=0" REPS=2 bash tools/gpu/run.sh r05_f abenv
