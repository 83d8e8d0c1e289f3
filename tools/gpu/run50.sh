cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 100 tools/gpu/igemm_probe loops > gpurun_out/probe_loops.log 2>&1; echo "probe exit $?"
grep -v "production" gpurun_out/probe_loops.log
