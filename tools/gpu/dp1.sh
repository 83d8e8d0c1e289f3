cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dp --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > gpurun_out/bench_dp_final.log 2>&1; echo "exit $?"
tail -1 gpurun_out/bench_dp_final.log | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('torchrun 1 rank, forced exchange:', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'])"
