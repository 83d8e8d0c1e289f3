# SQ-level counter passes (one counter set per pass, kernel trace only) of one workload, reduced per kernel: where do the
# waves of the MFMA kernels spend their cycles?   bash tools/gpu/pmc_sq.sh <tag> <bench.py arguments of the workload>
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/sq_$TAG
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES" \
         "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
         "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" \
         "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES"; do
  i=$((i+1))
  UNIPOSE_SYNC_WGRAD=1 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/sq_$TAG/p$i -o pmc --output-format csv -- python $R/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline --no-stock-baseline --no-profile --no-alt-math --no-other-configs > $R/gpurun_out/sq_$TAG/p$i.log 2>&1; echo "pass $i exit $?"
done
cd $R
python tools/pmc_sq_summary.py gpurun_out/sq_$TAG > gpurun_out/sq_${TAG}_summary.txt 2>&1
head -40 gpurun_out/sq_${TAG}_summary.txt
find gpurun_out/sq_$TAG -name "*.csv" -delete
