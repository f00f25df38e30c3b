"""Does it train?  N optimizer steps of complex_yolov4.cfg on one fixed synthetic batch (over-fitting it), fp16
performance mode with DynamicLossScale vs fp32 parity mode: prints the loss trajectory of both.
usage: python tools/train_probe.py [steps=100] [batch=8] [size=608]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.optim import DynamicLossScale, FusedAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 608
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
x, tg = syn.bev_images(B, S, seed=0).cuda(), syn.targets(B, 6, S, seed=0).cuda()


def run(dtype, n, scaler_on):
    torch.manual_seed(0)
    model = Darknet(cfg, use_giou_loss=True, dtype=dtype).cuda().train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    scaler = DynamicLossScale(model, opt, init_scale=64.0, growth_interval=50) if scaler_on else None
    out = []
    for i in range(n):
        opt.zero_grad(set_to_none=True)
        loss, _ = model(x, tg)
        loss.backward()
        if scaler:
            scaler.check()
        opt.step()
        if scaler:
            scaler.update()
        if i % max(1, n // 10) == 0 or i == n - 1:
            out.append((i, float(loss.detach().reshape(-1)[0]), scaler.scale if scaler else 1.0))
    return out, (scaler.skipped if scaler else 0)


for dtype, n, sc in (('f16', steps, True), ('f16', steps, False), ('f32', min(steps, 30), False)):
    traj, skipped = run(dtype, n, sc)
    print('%s %s: %s  (skipped %d)' % (dtype, 'dynamic-scale' if sc else 'scale 1      ',
                                       '  '.join('%d:%.2f@%g' % t for t in traj), skipped))
