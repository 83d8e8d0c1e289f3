cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --model lstm --steps 4 --warmup 2 --no-profile > gpurun_out/bench_lstm.log 2>&1; echo "lstm exit $?"; tail -2 gpurun_out/bench_lstm.log | cut -c1-700
timeout 300 python bench.py --size 736 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/bench_736.log 2>&1; echo "736 f32 exit $?"; tail -1 gpurun_out/bench_736.log | cut -c1-200
timeout 300 python bench.py --size 736 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-profile --math bf16 > gpurun_out/bench_736_bf16.log 2>&1; echo "736 bf16 exit $?"; tail -1 gpurun_out/bench_736_bf16.log | cut -c1-200
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
