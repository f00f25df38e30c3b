"""Host cost of one launch through the ctypes layer (enqueue only / with drain), with and without ops.stream_scope."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from complex_yolov4_pytorch_amd.ops import View, CY_F16
x = View.alloc(1, 8, 8, 64, CY_F16); y = View.alloc(1, 8, 8, 64, CY_F16)
sc = torch.ones(64, device='cuda'); sh = torch.zeros(64, device='cuda')
for _ in range(100): ops.bn_act_fwd(x, y, None, sc, sh, 0)
torch.cuda.synchronize()
for scoped in (False, True):
    ctx = ops.stream_scope(torch.cuda.current_stream()) if scoped else None
    if ctx: ctx.__enter__()
    t0 = time.perf_counter()
    for _ in range(2000): ops.bn_act_fwd(x, y, None, sc, sh, 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if ctx: ctx.__exit__()
    print('scoped=%s: enqueue %.2f us/launch, with drain %.2f us/launch' % (scoped, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
raw = ops.lib().raw('cy_version')
t0 = time.perf_counter()
for _ in range(20000): raw()
print('bare ctypes call %.2f us' % ((time.perf_counter() - t0) / 20000 * 1e6))
