# rocprofv3 kernel stats of the UniPose-LSTM step (K=13, B=8, T=5) — where do the 11x11 head convolutions stand?
cd $GRAFT_REPO_ROOT
TAG=${1:-r02_k}
mkdir -p gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --model lstm --num-classes 13 --batch 8 --frames 5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-alt-math --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/$TAG/prof.log | cut -c1-400
python tools/rocprof_summary.py $(find gpurun_out/$TAG/prof -name "*.db" | head -1) 4 > gpurun_out/$TAG/kernel_stats_lstm.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
head -30 gpurun_out/$TAG/kernel_stats_lstm.txt
UP_PROFILE_CSV=gpurun_out/$TAG/lstm_launches.csv timeout 300 python bench.py --model lstm --num-classes 13 --batch 8 --frames 5 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-math --no-other-configs > gpurun_out/$TAG/bench_lstm.log 2>&1
tail -1 gpurun_out/$TAG/bench_lstm.log | cut -c1-300
ls gpurun_out/$TAG
