cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_x; mkdir -p $OUT
timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -q -x -s --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?"; grep -E "hipGraph|passed|failed|Error|error" $OUT/pytest.log | head -20
