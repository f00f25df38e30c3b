"""BN + activation pass micro-benchmark (forward apply, backward reduce, backward apply): achieved HBM GB/s per shape.
usage: python tools/bn_micro.py [batch]   (shapes = the BN layers of complex_yolov4.cfg at 608x608)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from complex_yolov4_pytorch_amd.ops import CY_F16, View

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = [(608, 32), (304, 64), (152, 64), (152, 128), (76, 128), (76, 256), (38, 256), (38, 512), (19, 512), (19, 1024)]
if os.environ.get('BN_SHAPES'):      # e.g. BN_SHAPES=19x512,38x256 (for rocprofv3 runs: GPU-side durations of single shapes)
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in os.environ['BN_SHAPES'].split(',')]
ACTS = {'mish': ops.ACT['mish'], 'leaky': ops.ACT['leaky']}


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


for act_name, act in ACTS.items():
    for H, C in SHAPES:
        x = View.alloc(B, H, H, C, CY_F16); x.buf.normal_()
        y = View.alloc(B, H, H, C, CY_F16)
        dy = View.alloc(B, H, H, C, CY_F16); dy.buf.normal_()
        dx = View.alloc(B, H, H, C, CY_F16)
        f = lambda: torch.randn(C, device='cuda')
        mean, invstd, scale, shift = f() * 0.1, f().abs() + 0.5, f().abs() + 0.5, f() * 0.1
        dgs, dbs = f(), f()
        rows = ops.bn_bwd_rows(x.M, C, CY_F16)
        part = torch.zeros(rows * 2 * C, device='cuda')
        nbytes = x.M * C * 2
        t_f = timeit(lambda: ops.bn_act_fwd(x, y, None, scale, shift, act))
        t_r = timeit(lambda: ops.bn_act_bwd_reduce(x, dy, mean, invstd, scale, shift, act, part))
        t_a = timeit(lambda: ops.bn_act_bwd_apply(x, dy, dx, None, False, mean, invstd, scale, shift, dgs, dbs, act))
        print('%-5s %3dx%-3d C=%-4d %6.1f MB | fwd %7.1f us %5.0f GB/s | bwd_reduce %7.1f us %5.0f GB/s | bwd_apply %7.1f us %5.0f GB/s'
              % (act_name, H, H, C, nbytes / 1e6, t_f, 2 * nbytes / t_f / 1e3, t_r, 2 * nbytes / t_r / 1e3, t_a, 3 * nbytes / t_a / 1e3))

# ceiling: a plain device copy of the same tensors (1 read + 1 write) through torch's copy kernel
print('copy ceiling (torch .copy_, 1 read + 1 write):')
for H, C in SHAPES:
    a = torch.empty(B * H * H * C, dtype=torch.float16, device='cuda').normal_()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print('      %3dx%-3d C=%-4d %6.1f MB | %7.1f us %5.0f GB/s' % (H, H, C, a.numel() * 2 / 1e6, t, 2 * a.numel() * 2 / t / 1e3))
