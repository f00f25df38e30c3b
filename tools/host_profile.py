"""Where does the host time of one train step go?  cProfile over a few steps (top functions by own time)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.optim import FusedAdam

cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
model = Darknet(cfg, use_giou_loss=True, dtype='f16').cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-4)
x, tg = syn.bev_images(16, 608, seed=0).cuda(), syn.targets(16, 6, 608, seed=0).cuda()


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model(x, tg)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
