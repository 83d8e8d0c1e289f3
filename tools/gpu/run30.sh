cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/gpu/igemm_probe > gpurun_out/probe_caps.log 2>&1; echo "probe exit $?"
cat gpurun_out/probe_caps.log
timeout 200 tools/gpu/igemm_probe tiles > gpurun_out/probe_caps2.log 2>&1; echo "probe exit $?"
cat gpurun_out/probe_caps2.log
