cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 100 tools/gpu/igemm_probe wasp > gpurun_out/probe_wasp.log 2>&1; echo "probe exit $?"
cat gpurun_out/probe_wasp.log
