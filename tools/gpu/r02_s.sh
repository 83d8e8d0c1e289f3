cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_s; mkdir -p $OUT
timeout 500 python - > $OUT/stock.log 2>&1 <<'PY'
import time, torch, json, sys
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
for bm in (False, True):
    t0 = time.time()
    r = bench.stock_gpu_baseline(dev, 16, 368, 32, steps=5, warmup=3, benchmark=bm)
    print(bm, round(time.time() - t0, 1), "s", json.dumps(r), flush=True)
print("reserved GB", torch.cuda.memory_reserved() / 2**30)
PY
echo "exit $?"; tail -5 $OUT/stock.log
