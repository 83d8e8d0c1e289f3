cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/probe/p1 -o p --output-format csv -- $R/tools/gpu/igemm_probe > $R/gpurun_out/probe/p1.log 2>&1; echo "exit $?"
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum -d $R/gpurun_out/probe/p2 -o p --output-format csv -- $R/tools/gpu/igemm_probe > $R/gpurun_out/probe/p2.log 2>&1; echo "exit $?"
tail -3 $R/gpurun_out/probe/p2.log
