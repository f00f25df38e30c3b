"""Per-layer decision of the BN-backward-sums fusion (models/engine.py::_autotune_dgrad): time of the best plain dgrad, of
the separate reduce pass and of the best dgrad with the sums in its epilogue, for every layer the plan marks.
usage: python tools/dgrad_sums_probe.py [batch=16] [size=608] [dtype=f16]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models import engine as E
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 608
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16'
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
torch.manual_seed(0)
m = Darknet(cfg, use_giou_loss=True, dtype=dtype).cuda()
m.train()
x, tg = syn.bev_images(B, S, seed=0).cuda(), syn.targets(B, 6, S, seed=0).cuda()
loss, _ = m(x, tg)
loss.sum().backward()
torch.cuda.synchronize()
eng = next(iter(m._engines.values()))
memo = E._CONV_TUNE_MEMO
plain = {k[1:]: v for k, v in memo.items() if k[0] == 'dgrad'}
tot_plain = tot_fused = 0.0
print('fused layers: %d of %d marked' % (len(eng._sums_fused), sum(len(b.get('dx_sums', {})) for b in eng.plan.bwd)))
for k, (h, t) in sorted(memo.items(), key=lambda kv: str(kv[0])):
    if k[0] != 'dgrad+sums':
        continue
    act, rawld, key = k[1], k[2], k[3:]
    ph, pt = plain[key]
    red = [v for kk, v in memo.items() if kk[0] == 'bn_bwd_reduce' and kk[1] == act and kk[3] == key[1] * key[6] * key[7] and kk[4] == key[8]]
    rt = red[0][1] if red else float('nan')
    if h is None:
        print('%-60s not taken by the pipelined kernel' % (key,))
        continue
    win = t < pt + rt
    print('%-64s plain h%d %6.1f us + reduce %6.1f us = %6.1f | fused h%d %6.1f us  %s' % (
        key, ph, pt / 3 * 1e3, rt / 3 * 1e3, (pt + rt) / 3 * 1e3, h, t / 3 * 1e3, 'FUSED' if win else ''))
