cd $GRAFT_REPO_ROOT
cat > /tmp/t.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']+'/tests')
import torch, model_cases as mc
from unipose_amd import ops
from oracle import unipose_oracle as O
orig = mc.yardstick
worst = {}
def ys(ours, r32, r64, slack=10.0, floor=2e-5):
    ok, eo, er = orig(ours, r32, r64, slack, floor)
    worst['max'] = max(worst.get('max', 0), eo / (er + floor))
    if not ok: print('   FAIL', eo, er)
    return True, eo, er
mc.yardstick = ys
for mode in (0, 1, 0, 1):
    ops.ASYNC_WGRAD = bool(mode)
    worst.clear()
    mc.lstm_case(torch.device('cuda:0'), size=96, T=3, B=2, train=True)
    print('async', mode, 'worst ratio', worst['max'], flush=True)
PY
timeout 600 python /tmp/t.py 2>&1 | grep -v amdgpu.ids | tail -12
