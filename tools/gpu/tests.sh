cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu12.log 2>&1; echo "pytest exit $?"
tail -1 gpurun_out/pytest_gpu12.log
