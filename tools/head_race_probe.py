"""Is cy_yolo_loss reproducible while other kernels run beside it on another HIP stream?  Repeats the loss of one head on a
side stream, with and without a long conv kernel in flight on the main stream, and compares metrics / d(logits) bitwise.
usage: python tools/head_race_probe.py [iters=300]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.ops import CY_F16, View

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, G, A, C, S = 16, 76, 3, 3, 608
anchors = [(11, 14, 0, 1), (11, 14, -3.14, 1), (11, 14, 0.5, 0.8)]
torch.manual_seed(0)
logits = (torch.randn(B * G * G * A * (7 + C), device='cuda') * 0.5).contiguous()
tg = syn.targets(B, 6, S, seed=5).cuda()
need = ops.yolo_loss_workspace(B, G, A, C, tg.shape[0])
ws = torch.empty(need, dtype=torch.uint8, device='cuda')
met = torch.zeros(20, device='cuda')
dl = torch.empty_like(logits)
side = torch.cuda.Stream()
x = View.alloc(16, 152, 152, 128, CY_F16); x.buf.normal_()
y = View.alloc(16, 152, 152, 128, CY_F16)
w = torch.randn(128, 128, 3, 3, device='cuda') * 0.03
wf, _ = ops.pack_weights(w, 128, 128, CY_F16)


def loss_on_side():
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    side.wait_event(ev)
    with torch.cuda.stream(side), ops.stream_scope(side):
        ops.yolo_loss(logits, B, G, A, C, tg, anchors, S, 0.7, True, ws, met, dl)


for busy in (False, True):
    ref, bad = None, 0
    for it in range(iters):
        if busy:
            for _ in range(3):
                ops.conv_igemm(x, wf, 128, y, 3, 1, 1)
        loss_on_side()
        if busy:
            for _ in range(3):
                ops.conv_igemm(x, wf, 128, y, 3, 1, 1)
        torch.cuda.synchronize()
        cur = (met.clone(), dl.clone())
        if ref is None:
            ref = cur
        elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
            bad += 1
            if bad == 1:
                d = [(i, float(cur[0][i]), float(ref[0][i])) for i in range(20) if cur[0][i] != ref[0][i]]
                print('   first mismatch at iteration %d: metrics %s; dlogits max |d| %.3e' % (it, d, float((cur[1] - ref[1]).abs().max())))
    print('conv kernels in flight on the main stream: %s -> %d of %d repeats differ' % (busy, bad, iters - 1))
