import sys, os; sys.path.insert(0, os.getcwd())
import torch, numpy as np
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
CFG = 'complex-yolov4-pytorch_amd/config/cfg/'
for cfg, dt, B, S in (('complex_yolov4_tiny.cfg', 'f32', 4, 608), ('complex_yolov4_tiny.cfg', 'f16', 4, 608), ('complex_yolov4.cfg', 'f32', 4, 416), ('complex_yolov4.cfg', 'f16', 4, 416), ('complex_yolov4.cfg', 'f16', 16, 608)):
    torch.manual_seed(0)
    m = Darknet(CFG + cfg, use_giou_loss=True, dtype=dt).cuda().train()
    x, tg = syn.bev_images(B, S, seed=21), syn.targets(B, 6, S, seed=21)
    res = []
    for trial in range(3):
        for p in m.parameters(): p.grad = None
        if trial == 0 or trial == 2:
            xx, tt = x, tg
        else:
            perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)); inv = torch.empty_like(perm); inv[perm] = torch.arange(B)
            xx = x[perm]; tt = tg.clone(); tt[:, 0] = inv[tg[:, 0].long()].float()
        loss, _ = m(xx.cuda(), tt.cuda()); loss.backward()
        res.append((float(loss.detach()), m.flat_grad.clone()))
    d_same = float((res[2][1] - res[0][1]).norm() / res[0][1].norm())
    d_perm = float((res[1][1] - res[0][1]).norm() / res[0][1].norm())
    print('%s %s B=%d S=%d: loss %.4f / perm %.4f / repeat %.4f | grad rel diff: repeat %.2e, permuted %.2e, |g| %.3e' % (cfg, dt, B, S, res[0][0], res[1][0], res[2][0], d_same, d_perm, float(res[0][1].norm())))
