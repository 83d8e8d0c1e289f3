cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_ao
timeout 400 python bench.py --stock-baseline-only --model lstm 2>gpurun_out/r02_ao/lstm.err | tail -1 | tee gpurun_out/r02_ao/stock_lstm.json | cut -c1-400
timeout 400 python bench.py --stock-baseline-only --size 736 --batch 16 2>gpurun_out/r02_ao/736.err | tail -1 | tee gpurun_out/r02_ao/stock_736_f32.json | cut -c1-400
tail -2 gpurun_out/r02_ao/lstm.err
