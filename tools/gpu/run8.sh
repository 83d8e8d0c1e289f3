cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cat > /tmp/t.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']+'/tests')
import torch, model_cases as mc
from unipose_amd import ops, _C
ops.ASYNC_WGRAD = False
real = _C.load()
class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)
        if not name.startswith('up_') or name in ('up_last_error',): return fn
        def w(*a):
            descs = []
            for x in a:
                try:
                    d = x._obj
                    descs.append({f[0]: getattr(d, f[0]) for f in d._fields_ if isinstance(getattr(d, f[0]), int)})
                except Exception: pass
            print('CALL', name, descs, flush=True)
            r = fn(*a)
            torch.cuda.synchronize()
            return r
        return w
_C._lib = Proxy()
mc.lstm_case(torch.device('cuda:0'), size=96, T=3, B=2, train=True)
print('ok')
PY
timeout 300 python /tmp/t.py > gpurun_out/trace.log 2>&1
tail -6 gpurun_out/trace.log | cut -c1-400
