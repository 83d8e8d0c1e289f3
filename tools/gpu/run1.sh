set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -5
nproc
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -5 gpurun_out/bench.log
