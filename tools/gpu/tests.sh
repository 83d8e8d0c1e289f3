# the -m gpu suite (pass test paths to narrow it)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
[ $# -eq 0 ] && set -- tests
timeout 900 python -m pytest "$@" -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu_last.log 2>&1; echo "pytest exit $?"
tail -1 gpurun_out/pytest_gpu_last.log
