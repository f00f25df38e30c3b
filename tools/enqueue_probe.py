import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.optim import FusedAdam
cfg = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
model = Darknet(cfg, use_giou_loss=True, dtype='f16').cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-4)
x, tg = syn.bev_images(16, 608, seed=0).cuda(), syn.targets(16, 6, 608, seed=0).cuda()
def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model(x, tg); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.2f ms/step, total %.2f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
