cd $GRAFT_REPO_ROOT
echo "A torchrun, force-dp (hwq 8 + early side stream)"; timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-profile --force-dp 2>&1 | tail -1 | cut -c90-200
echo "B same with GPU_MAX_HW_QUEUES=4"; GPU_MAX_HW_QUEUES=4 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-profile --force-dp 2>&1 | tail -1 | cut -c90-200
echo "D plain"; timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c90-200
