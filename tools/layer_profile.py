"""Per-layer achieved rates of the conv kernels (forward igemm, dgrad igemm, wgrad) for one train step of
complex_yolov4.cfg: HIP events around every launch, grouped by conv shape.  Usage: python tools/layer_profile.py [batch] [size]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models import engine as eng_mod
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 608
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
model = Darknet(cfg, use_giou_loss=True, dtype='f16').cuda().train()
x, tg = syn.bev_images(B, S, seed=0).cuda(), syn.targets(B, 6, S, seed=0).cuda()
cur = {}
orig = eng_mod.Engine._conv_work


def tagged(self, rec):
    cur['rec'] = rec
    return orig(self, rec)


eng_mod.Engine._conv_work = tagged


class Prof(ops.LaunchProfiler):
    def bracket(self, kind, flops, nbytes):
        r = cur['rec']
        phase = 'bwd' if torch.is_grad_enabled() and getattr(self, 'in_bwd', False) else 'fwd'
        key = (kind if kind == 'wgrad' else ('dgrad' if self.in_bwd else 'fwd'), r['cin'], r['cout'], r['ks'], r['stride'], r['H'])
        return ops._Bracket(self, key, flops, nbytes)


for _ in range(2):
    loss, _ = model(x, tg); loss.backward()
p = Prof(); p.in_bwd = False
ops.PROFILER = p
loss, _ = model(x, tg)
p.in_bwd = True
loss.backward()
ops.PROFILER = None
summ = p.summary()
rows = sorted(summ.items(), key=lambda kv: -kv[1]['ms'])
tot = sum(v['ms'] for v in summ.values())
print('total conv-kernel ms/step %.2f (batch %d, %dx%d)' % (tot, B, S, S))
# "floor": the launch at 4.5 TB/s of algorithmic bytes or 1500 TFLOP/s, whichever is longer (what a well-fed kernel reaches
# on this chip); "lost" = time above it, summed over the shape's launches -- where the remaining conv time sits
print('%-6s %5s %5s %2s %1s %4s | %3s %8s %8s %8s %6s %8s %8s' % ('kind', 'cin', 'cout', 'k', 's', 'H', 'n', 'ms', 'TFLOP/s', 'GB/s', '%time', 'us/launch', 'lost ms'))
lost_tot = 0.0
for (kind, cin, cout, ks, st, H), v in rows:
    floor = max(v['bytes'] / 4.5e12, v['flops'] / 1.5e15) * 1e3
    lost = v['ms'] - floor
    lost_tot += lost
    print('%-6s %5d %5d %2d %1d %4d | %3d %8.3f %8.1f %8.1f %6.2f %8.1f %8.3f' % (kind, cin, cout, ks, st, H, v['launches'], v['ms'],
          v['flops'] / v['ms'] / 1e9, v['bytes'] / v['ms'] / 1e6, 100 * v['ms'] / tot, 1e3 * v['ms'] / v['launches'], lost))
print('time above the floor: %.2f ms of %.2f' % (lost_tot, tot))
for kind in ('fwd', 'dgrad', 'wgrad'):
    ms = sum(v['ms'] for k, v in summ.items() if k[0] == kind); fl = sum(v['flops'] for k, v in summ.items() if k[0] == kind)
    print('%s: %.2f ms, %.1f TFLOP/s' % (kind, ms, fl / ms / 1e9))
