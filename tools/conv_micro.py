"""Single-shape conv kernel micro-benchmark (forward igemm / dgrad / wgrad), for A/B work and PMC profiling.
usage: python tools/conv_micro.py N Cin Cout ks stride H [iters] [kinds=fwd,dgrad,wgrad]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from complex_yolov4_pytorch_amd.ops import CY_F16, View

N, Ci, Co, ks, st, H = [int(v) for v in sys.argv[1:7]]
if os.environ.get('PIPE_CFG'):      # "mode,cap,bn,variant,bm_eff" -> cy_conv_pipe_config (A/B of the two conv kernels)
    ops.conv_pipe_config(*[int(v) for v in os.environ['PIPE_CFG'].split(',')])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
kinds = (sys.argv[8] if len(sys.argv) > 8 else 'fwd,dgrad,wgrad').split(',')
pad = (ks - 1) // 2
OH = (H + 2 * pad - ks) // st + 1
dt = CY_F16
x = View.alloc(N, H, H, Ci, dt); x.buf.normal_()
y = View.alloc(N, OH, OH, Co, dt)
dy = View.alloc(N, OH, OH, Co, dt); dy.buf.normal_()
dx = View.alloc(N, H, H, Ci, dt)
w = torch.randn(Co, Ci, ks, ks, device='cuda') * 0.05
wf, wd = ops.pack_weights(w, Co, Ci, dt)
M = N * OH * OH
rows = ops.conv_stats_rows(M, Co)
stats = torch.zeros((rows + ops.bn_scratch_rows()) * 2 * Co, device='cuda')
split = ops.wgrad_split(M, Co, Ci, ks)
part = torch.empty(split * Co * ks * ks * Ci, device='cuda')
flops = 2.0 * M * Co * ks * ks * Ci


def run(kind):
    if kind == 'fwd':
        ops.conv_igemm(x, wf, Co, y, ks, st, pad, flags=ops.CONV_STATS, stats=stats)
    elif kind == 'dgrad':
        ops.conv_igemm(dy, wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED)
    else:
        ops.conv_wgrad(dy, x, ks, st, pad, part, split)


for kind in kinds:
    for _ in range(3):
        run(kind)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run(kind)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    print('%-5s N=%d %d->%d k%d s%d H=%d: %.1f us  %.1f TFLOP/s' % (kind, N, Ci, Co, ks, st, H, us, flops / us / 1e6))
