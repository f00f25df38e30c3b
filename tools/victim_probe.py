"""Are OTHER kernels of the step disturbed by the conv instantiations that disturb the GIoU head kernels
(tools/head_race_probe2.py: igemm_fast<192,128> = tile hint 1 on a 152x152x128 layer, the pipelined 384x128 tile = hint 5)?
Victim kernels run on a side stream, the aggressor on the main stream; each victim's output is compared bit for bit with its
first run.   usage: python tools/victim_probe.py [iters=1500] [aggressor hint=1]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from complex_yolov4_pytorch_amd.ops import CY_F16, View

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
hint = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
side = torch.cuda.Stream()
x = View.alloc(16, 152, 152, 128, CY_F16); x.buf.normal_()
y = View.alloc(16, 152, 152, 128, CY_F16)
w = torch.randn(128, 128, 3, 3, device='cuda') * 0.03
wf, wd = ops.pack_weights(w, 128, 128, CY_F16)
# victims (small launches, like the layers of the 19 / 38 grids)
vx = View.alloc(16, 38, 38, 256, CY_F16); vx.buf.normal_()
vy = View.alloc(16, 38, 38, 256, CY_F16)
vg = View.alloc(16, 38, 38, 256, CY_F16); vg.buf.normal_()
vo = View.alloc(16, 38, 38, 256, CY_F16)
sc = (torch.rand(256, device='cuda') + 0.5); sh = torch.randn(256, device='cuda') * 0.2
mean = torch.randn(256, device='cuda') * 0.1; inv = torch.rand(256, device='cuda') + 0.5
dgs = torch.randn(256, device='cuda') * 0.01; dbs = torch.randn(256, device='cuda') * 0.01
w2 = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
wf2, wd2 = ops.pack_weights(w2, 256, 256, CY_F16)
cy = View.alloc(16, 38, 38, 256, CY_F16)
part = torch.empty(4 * 256 * 9 * 256, device='cuda')

VICTIMS = {
    'bn_act_fwd(mish)': (lambda: ops.bn_act_fwd(vx, vy, None, sc, sh, 2), lambda: vy.buf),
    'bn_act_bwd_apply(mish)': (lambda: ops.bn_act_bwd_apply(vx, vg, vo, None, False, mean, inv, sc, sh, dgs, dbs, 2), lambda: vo.buf),
    'conv3x3 256 @38 (pipelined, hint 4)': (lambda: ops.conv_igemm(vx, wf2, 256, cy, 3, 1, 1, tile=4), lambda: cy.buf),
    'conv3x3 256 @38 (4-wave, hint 1)': (lambda: ops.conv_igemm(vx, wf2, 256, cy, 3, 1, 1, tile=1), lambda: cy.buf),
    'wgrad 256x256 3x3 @38': (lambda: ops.conv_wgrad(vg, vx, 3, 1, 1, part, 4), lambda: part),
}
for name, (run, out) in VICTIMS.items():
    for busy in (False, True):
        ref, bad = None, 0
        for it in range(iters):
            if busy:
                for _ in range(3):
                    ops.conv_igemm(x, wf, 128, y, 3, 1, 1, tile=hint)
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); side.wait_event(ev)
            with torch.cuda.stream(side), ops.stream_scope(side):
                run()
            if busy:
                for _ in range(3):
                    ops.conv_igemm(x, wf, 128, y, 3, 1, 1, tile=hint)
            torch.cuda.synchronize()
            cur = out().clone()
            if ref is None:
                ref = cur
            elif not torch.equal(cur.view(torch.int16) if cur.element_size() == 2 else cur.view(torch.int32),
                                 ref.view(torch.int16) if ref.element_size() == 2 else ref.view(torch.int32)):
                bad += 1
        print('%-40s aggressor hint %d in flight: %-5s -> %d of %d repeats differ' % (name, hint, busy, bad, iters - 1), flush=True)
