cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline --no-profile --force-dp > gpurun_out/bench_dp1.log 2>&1; echo "dp1 exit $?"; tail -3 gpurun_out/bench_dp1.log | cut -c1-400
timeout 300 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/bench_nodp.log 2>&1; tail -1 gpurun_out/bench_nodp.log | cut -c1-200
