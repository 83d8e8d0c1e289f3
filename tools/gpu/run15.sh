cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --size 64 --batch 2 2>&1 | tail -1 | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-profile --force-dp --size 64 --batch 2 2>&1 | tail -1 | cut -c1-200
python - <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from model.unipose import unipose
from unipose_amd import ops
dev = torch.device('cuda:0')
m = unipose('MPII', num_classes=16).to(dev).train()
x = torch.randn(2,3,64,64, device=dev); t = torch.rand(2,17,8,8, device=dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
for it in range(3):
    opt.zero_grad(); l = ops.mse_loss(m(x), t); l.backward(); opt.step()
torch.cuda.synchronize()
t0=time.perf_counter(); 
for it in range(10):
    opt.zero_grad()
    ta=time.perf_counter(); y = m(x); tb=time.perf_counter(); l = ops.mse_loss(y, t); l.backward(); tc=time.perf_counter(); opt.step(); td=time.perf_counter()
torch.cuda.synchronize()
print('host ms: fwd %.1f bwd %.1f opt %.1f total/step %.1f' % ((tb-ta)*1e3, (tc-tb)*1e3, (td-tc)*1e3, (time.perf_counter()-t0)*100))
PY
