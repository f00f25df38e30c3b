"""A/B of the pipelined conv kernel (csrc/conv_pipe.hip) against the 4-wave kernels (csrc/conv_igemm.hip) on the layer
shapes of complex_yolov4.cfg at batch N: every tile capacity / variant is checked against the 4-wave result on the same
inputs (output tensor, BN statistics) and timed.
usage: python tools/pipe_probe.py [N=16] [dtype=f16] [quick]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from complex_yolov4_pytorch_amd.ops import View

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dt = ops.dtype_code(sys.argv[2]) if len(sys.argv) > 2 else ops.CY_F16
quick = len(sys.argv) > 3
ITERS = 10

# (Cin, Cout, ks, stride, H_in)
SHAPES = [(128, 128, 3, 1, 76), (128, 256, 3, 1, 76), (256, 256, 3, 1, 38), (256, 512, 3, 1, 38), (512, 512, 3, 1, 19),
          (512, 1024, 3, 1, 19), (64, 64, 3, 1, 152), (64, 128, 3, 2, 304), (128, 256, 3, 2, 152), (256, 512, 3, 2, 76),
          (64, 64, 1, 1, 304), (128, 64, 1, 1, 304), (128, 128, 1, 1, 152), (256, 128, 1, 1, 76), (128, 128, 1, 1, 76),
          (512, 256, 1, 1, 38), (256, 256, 1, 1, 38), (1024, 512, 1, 1, 19), (2048, 512, 1, 1, 19)]
if quick:
    SHAPES = SHAPES[:2] + SHAPES[5:6] + SHAPES[12:13]
CONFIGS = [('old', dict(mode=0)), ('policy', dict(mode=2)),
           ('c128', dict(mode=2, cap=128, bn=0)), ('c192', dict(mode=2, cap=192, bn=0)), ('c256', dict(mode=2, cap=256, bn=0)),
           ('c384', dict(mode=2, cap=384, bn=0)),
           ('c128/lc', dict(mode=2, cap=128, bn=0, variant=3)), ('c192/lc', dict(mode=2, cap=192, bn=0, variant=3)),
           ('c256/lc', dict(mode=2, cap=256, bn=0, variant=3))]


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / ITERS


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)


for Ci, Co, ks, st, H in SHAPES:
    pad = (ks - 1) // 2
    OH = (H + 2 * pad - ks) // st + 1
    M = N * OH * OH
    torch.manual_seed(Ci * 7 + Co)
    x = View.alloc(N, H, H, Ci, dt); x.buf.normal_()
    y = View.alloc(N, OH, OH, Co, dt)
    dy = View.alloc(N, OH, OH, Co, dt); dy.buf.normal_()
    dx = View.alloc(N, H, H, Ci, dt)
    res = View.alloc(N, OH, OH, Co, dt); res.buf.normal_()
    w = torch.randn(Co, Ci, ks, ks, device='cuda') * (1.0 / (ks * ks * Ci) ** 0.5)
    wf, wd = ops.pack_weights(w, Co, Ci, dt)
    rows = ops.conv_stats_rows(M, Co)
    stats = torch.zeros(rows * 2 * Co, device='cuda')
    scale, shift = torch.rand(Co, device='cuda') + 0.5, torch.randn(Co, device='cuda') * 0.1
    flops = 2.0 * M * Co * ks * ks * Ci

    def fwd():
        ops.conv_igemm(x, wf, Co, y, ks, st, pad, flags=ops.CONV_STATS, stats=stats)

    def dgrad():
        ops.conv_igemm(dy, wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED)

    def dgrad_acc():
        ops.conv_igemm(dy, wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED | ops.CONV_ACCUM)

    def evalf():
        ops.conv_bn_act_eval(x, wf, Co, y, ks, st, pad, scale, shift, ops.ACT['mish'], res)

    ref = {}
    print('== N=%d %d->%d k%d s%d H=%d  (M=%d, %.1f GFLOP)' % (N, Ci, Co, ks, st, H, M, flops / 1e9))
    for name, cfg in CONFIGS:
        c = dict(cfg)
        if 'bn' in c:
            c['bn'] = 128 if Co > 64 else 64
            if c['cap'] == 192 and c['bn'] == 64:
                continue
        ops.conv_pipe_config(**c)
        line = '   %-12s' % name
        try:
            for kind, fn, out in (('fwd', fwd, y), ('dgrad', dgrad, dx), ('dacc', dgrad_acc, dx), ('eval', evalf, y)):
                if quick and kind in ('dacc',):
                    continue
                n0 = ops.pipe_launches()
                stats.zero_()
                if kind == 'dacc':
                    dx.buf.fill_(0.25)
                fn()
                torch.cuda.synchronize()
                used = ops.pipe_launches() - n0
                got = out.buf.clone()
                st_sum = stats.view(rows, 2, Co).sum(0).clone() if kind == 'fwd' else None
                if name == 'old':
                    ref[kind] = (got, st_sum)
                    err = serr = 0.0
                else:
                    err = relerr(got, ref[kind][0])
                    serr = relerr(st_sum, ref[kind][1]) if st_sum is not None else 0.0
                stats.zero_()
                us = timed(fn) if kind != 'dacc' else 0.0
                stats.zero_()
                flag = '' if (err < 4e-3 and serr < 1e-3) else ' **MISMATCH**'
                if kind == 'dacc':
                    line += ' | dacc err %.1e%s' % (err, flag)
                else:
                    line += ' | %s %6.1f us %4.0f TF [%d] err %.1e%s%s' % (kind, us, flops / us / 1e6, used, err,
                                                                         (' st %.0e' % serr) if kind == 'fwd' else '', flag)
        except Exception as ex:  # noqa: BLE001
            line += ' | FAILED: %s' % ex
        print(line, flush=True)
ops.conv_pipe_config(mode=1)
