cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 tools/gpu/igemm_probe > gpurun_out/probe_prio.log 2>&1; echo "probe exit $?"
grep -v "first blocks\|XCD finish\|CU residency\|starts p10\|timeline" gpurun_out/probe_prio.log
