"""Host enqueue cost of the train step by phase (forward call, loss.backward(), optimizer step) against the GPU time of the
whole step: each phase is timed on the host clock starting from an idle GPU queue, so the figure is pure enqueue + Python.
usage: python tools/enqueue_probe.py [dtype=f16]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.optim import FusedAdam

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16'
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
model = Darknet(cfg, use_giou_loss=True, dtype=dtype).cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-4)
x, tg = syn.bev_images(16, 608, seed=0).cuda(), syn.targets(16, 6, 608, seed=0).cuda()


def step(times=None):
    sync = torch.cuda.synchronize if times is not None else (lambda: None)
    sync(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss, _ = model(x, tg)
    t1 = time.perf_counter(); sync(); t1b = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter(); sync(); t2b = time.perf_counter()
    opt.step()
    t3 = time.perf_counter(); sync()
    if times is not None:
        times.append((t1 - t0, t2 - t1b, t3 - t2b))


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('free-running: host enqueue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
times = []
for _ in range(10):
    step(times)
f, b, o = [sum(t[i] for t in times) / len(times) * 1e3 for i in range(3)]
print('host time per phase from an idle queue: forward %.2f ms, backward %.2f ms, optimizer %.2f ms (sum %.2f ms)' % (f, b, o, f + b + o))

# GPU-side bubble between the end of the forward pass and the first backward kernel, free-running (no host syncs): events
# recorded after model(...) returns and at the entry of Engine.backward
from complex_yolov4_pytorch_amd.models import engine as E

evs = []
orig = E.Engine.backward


def patched(self, *a, **k):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    return orig(self, *a, **k)


E.Engine.backward = patched
pairs = []
for _ in range(10):
    opt.zero_grad(set_to_none=True)
    loss, _ = model(x, tg)
    ea = torch.cuda.Event(enable_timing=True)
    ea.record()
    loss.backward()
    pairs.append((ea, evs[-1]))
    opt.step()
torch.cuda.synchronize()
print('GPU time between forward end and backward entry (free-running): ' + ' '.join('%.0f' % (1e3 * a.elapsed_time(b)) for a, b in pairs) + ' us')
