cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
grep -i -E "MFMA|FETCH_SIZE|WRITE_SIZE|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|LDS_BANK" $R/gpurun_out/pmc/counters.txt | head -40
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc/$tag -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/pmc/$tag.log 2>&1; echo "$tag exit $?"
  ls $R/gpurun_out/pmc/$tag | head -5
done
