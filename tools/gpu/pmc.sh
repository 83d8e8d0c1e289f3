# separate --pmc passes (one counter set per pass, kernel-trace only) of the exclusive-mode bench command
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
TAG=${1:-r01_h}
mkdir -p gpurun_out/pmc_$TAG
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  UNIPOSE_SYNC_WGRAD=1 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_$TAG/$tag -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --settle 0 --no-cpu-baseline --no-stock-baseline --no-profile --no-alt-math --no-other-configs > $R/gpurun_out/pmc_$TAG/$tag.log 2>&1; echo "$tag exit $?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_${TAG}_summary.txt 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_$TAG $TAG > gpurun_out/pmc_traffic.json
head -14 gpurun_out/pmc_${TAG}_summary.txt
find gpurun_out/pmc_$TAG -name "*.csv" -delete
