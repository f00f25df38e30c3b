"""Same-process A/B of the conv kernels on v4's 3x3 / stride-1 layers at batch 16 (608 x 608): every kernel / tile hint
(1: 4-wave kernels, 2-9: the pipelined kernel and its loader split, 12 / 13: the slab kernel) timed in interleaved rounds
(guide rule 24), forward with BN statistics and dgrad.  usage: python tools/slab_micro.py [rounds] [reps] [shapes=all|i,j]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from complex_yolov4_pytorch_amd.ops import CY_F16, View

SHAPES = [(16, 512, 1024, 19), (16, 512, 512, 19), (16, 256, 512, 38), (16, 256, 256, 38), (16, 128, 256, 76), (16, 128, 128, 76),
          (16, 1024, 512, 19), (16, 512, 256, 38), (16, 256, 128, 76)]     # the last three: the dgrads of 512->1024 etc.
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sel = sys.argv[3] if len(sys.argv) > 3 else 'all'
shapes = SHAPES if sel == 'all' else [SHAPES[int(i)] for i in sel.split(',')]
HINTS = [1, 3, 4, 7, 8, 9, 12, 13, 112, 113, 212, 213, 312, 313]     # 1xx: slab loaders variant, 2xx: K-split 3-stage, 3xx: 4-stage
if os.environ.get('SLAB_HINTS'):
    HINTS = [int(v) for v in os.environ['SLAB_HINTS'].split(',')]
dt = CY_F16
for (N, Ci, Co, H) in shapes:
    x = View.alloc(N, H, H, Ci, dt); x.buf.normal_()
    if os.environ.get('ZERO') == '1':
        x.buf.zero_()
    y = View.alloc(N, H, H, Co, dt)
    w = torch.randn(Co, Ci, 3, 3, device='cuda') * (0.0 if os.environ.get('ZERO') == '1' else 0.05)
    wf, wd = ops.pack_weights(w, Co, Ci, dt)
    dy = View.alloc(N, H, H, Co, dt); dy.buf.normal_()
    if os.environ.get('ZERO') == '1':
        dy.buf.zero_()
    dx = View.alloc(N, H, H, Ci, dt)
    M = N * H * H
    stats = torch.zeros((ops.conv_stats_rows(M, Co) + ops.bn_scratch_rows()) * 2 * Co, device='cuda')
    flops = 2.0 * M * Co * 9 * Ci
    for kind in ('fwd', 'dgrad'):
        def run(h):
            if h > 100:
                ops.conv_slab_config({1: 3, 2: 5, 3: 9}[h // 100], 0)
                h = h % 100
            else:
                ops.conv_slab_config(1, 0)
            if kind == 'fwd':
                ops.conv_igemm(x, wf, Co, y, 3, 1, 1, flags=ops.CONV_STATS, stats=stats, tile=h)
            else:
                ops.conv_igemm(dy, wd, Ci, dx, 3, 1, 1, flags=ops.CONV_TRANSPOSED, tile=h)
        best = {}
        hints = [h for h in HINTS if not (h % 100 in (12, 13) and (Co if kind == 'fwd' else Ci) <= 64)]
        for h in hints:
            run(h)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for h in hints:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    run(h)
                e.record()
                e.synchronize()
                us = s.elapsed_time(e) * 1e3 / reps
                best.setdefault(h, []).append(us)
        med = {h: sorted(v)[len(v) // 2] for h, v in best.items()}
        old = min([med[h] for h in hints if h < 11] or [float('nan')])
        line = '%-5s %4d->%4d @%2d: ' % (kind, Ci if kind == 'fwd' else Co, Co if kind == 'fwd' else Ci, H)
        line += ' '.join('h%d %.1f' % (h, med[h]) for h in hints)
        new = min([med[h] for h in hints if h > 11] or [float('nan')])
        line += ' | best old %.1f us (%.0f TF) slab %.1f us (%.0f TF) x%.3f' % (old, flops / old / 1e6, new, flops / new / 1e6, old / new)
        print(line, flush=True)
