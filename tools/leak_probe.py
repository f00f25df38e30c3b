"""Steady state check: device memory and step time over a few hundred train steps (nothing may grow)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.optim import DynamicLossScale, FusedAdam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
model = Darknet(cfg, use_giou_loss=True, dtype='f16').cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-4)
scaler = DynamicLossScale(model, opt, init_scale=16.0, growth_interval=100)
x, tg = syn.bev_images(16, 608, seed=0).cuda(), syn.targets(16, 6, 608, seed=0).cuda()
marks = {}
for i in range(n):
    if i in (20, n - 20):
        torch.cuda.synchronize()
        marks[i] = (time.perf_counter(), torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    opt.zero_grad(set_to_none=True)
    loss, _ = model(x, tg)
    loss.backward()
    scaler.check()
    opt.step()
    scaler.update()
torch.cuda.synchronize()
(t0, a0, r0), (t1, a1, r1) = marks[20], marks[n - 20]
print('steps %d..%d: %.2f ms/step; allocated %.3f -> %.3f GB, reserved %.3f -> %.3f GB; loss %.3f scale %g skipped %d'
      % (20, n - 20, (t1 - t0) / (n - 40) * 1e3, a0 / 2**30, a1 / 2**30, r0 / 2**30, r1 / 2**30,
         float(loss.detach().reshape(-1)[0]), scaler.scale, scaler.skipped))
