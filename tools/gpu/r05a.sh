# round 5, GPU call A: new tests (G16 trajectory, RCCL child, fine tail parts / persistent forms) + per-shape and whole-step A/B
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_a; mkdir -p $OUT
python -c "
from unipose_amd.build import library_is_current, source_hash
print('libunipose_hip.so built from this tree:', library_is_current(), source_hash())" | tee $OUT/build_identity.txt
timeout 900 python -m pytest tests/test_glds32_gpu.py "tests/test_configs_gpu.py::test_g16_adam_trajectory_vs_reference_golden" "tests/test_configs_gpu.py::test_one_rank_rccl_gradient_exchange" "tests/test_configs_gpu.py::test_g11_train_b8_vs_reference_golden" "tests/test_model_gpu.py::test_g4_train_128_vs_reference_golden" "tests/test_model_gpu.py::test_train_step_368_vs_oracle_yardstick" -m gpu -q -x -s --timeout 600 > $OUT/pytest_new.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_new.log
grep -h "g16\|stem\|backbone.conv1.weight\|RCCL" $OUT/pytest_new.log | head -40
timeout 600 python tools/gpu/shape_ab.py base tail_per_cu=2 tail_per_cu=4 persist=1 persist=2 persist=1+stagger=2 persist=1+stagger=4 persist=1+stagger=8 stagger=4 tail_per_cu=4+persist=1+stagger=4 > $OUT/shape_ab.txt 2>&1; echo "shape_ab exit $?"; grep -v "^{" $OUT/shape_ab.txt | tail -16
timeout 600 python tools/gpu/tune_ab.py --rounds 3 --steps 5 base tail_per_cu=2 tail_per_cu=4 persist=1 persist=2 persist=1+stagger=4 stagger=4 tail_per_cu=4+persist=1+stagger=4 tail_per_cu=4+persist=2 > $OUT/tune_ab.txt 2>&1; echo "tune_ab exit $?"; tail -12 $OUT/tune_ab.txt | cut -c1-600
