cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_f
timeout 300 python bench.py --model lstm --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_lstm2.log 2>&1; echo "lstm exit $?"
tail -1 gpurun_out/bench_lstm2.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('lstm', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --size 736 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_736b.log 2>&1; echo "736 exit $?"
tail -1 gpurun_out/bench_736b.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('736', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --force-dp --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-profile > gpurun_out/bench_dp2.log 2>&1; echo "dp exit $?"
tail -1 gpurun_out/bench_dp2.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('force-dp', d['value'], d['ms_per_step'])"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  UNIPOSE_SYNC_WGRAD=1 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_f/$tag -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math > $R/gpurun_out/pmc_f/$tag.log 2>&1; echo "$tag exit $?"
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_f > gpurun_out/pmc_f_summary.txt 2>&1
head -16 gpurun_out/pmc_f_summary.txt
find gpurun_out/pmc_f -name "*.csv" -size +3M -delete
