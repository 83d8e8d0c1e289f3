cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/gpu/igemm_probe wgrad > gpurun_out/probe_wgrad.log 2>&1; echo "probe exit $?"
grep -v "first blocks" gpurun_out/probe_wgrad.log
