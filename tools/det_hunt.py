"""Hunt for a rare run-to-run difference of the deterministic mode: repeat the same train step N times in one process,
hash every activation / gradient storage after each pass and report the first storage (in plan order) whose hash differs
from the first run's.  usage: python tools/det_hunt.py [dtype=bf16] [runs=60]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
torch.manual_seed(0)
model = Darknet(cfg, use_giou_loss=True, dtype=dtype, deterministic=True).cuda().train()
x, tg = syn.bev_images(16, 608, seed=5).cuda(), syn.targets(16, 6, 608, seed=5).cuda()


def digest(t):
    # cheap order-sensitive checksum on the device (exact equality of two tensors <=> equal with overwhelming probability)
    v = t.view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64)
    w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 65521
    return int((v * w).sum())


ref = None
for it in range(runs):
    model.zero_grad(set_to_none=True)
    loss, out = model(x, tg)
    eng = next(iter(model._engines.values()))
    torch.cuda.synchronize()
    fwd = {sid: digest(t) for sid, t in eng.act.items()}
    mets = [m.clone() for m in eng.metrics]
    loss.backward()
    torch.cuda.synchronize()
    bwd = {sid: digest(t) for sid, t in eng.gact.items()}
    cur = (float(loss.detach()), fwd, bwd, digest(model.flat_grad), mets)
    if ref is None:
        ref = cur
        continue
    if cur[0] != ref[0] or cur[1] != ref[1] or cur[2] != ref[2] or cur[3] != ref[3]:
        print('run %d differs from run 0: loss %r vs %r' % (it, cur[0], ref[0]))
        for rec in eng.plan.fwd:
            for key in ('raw', 'out', 'logits'):
                r = rec.get(key)
                if r is not None and r.st.sid in fwd and fwd[r.st.sid] != ref[1][r.st.sid]:
                    print('   first forward storage that differs: module %s op %s key %s (%s)' % (rec.get('idx'), rec['op'], key, r.st))
                    break
            else:
                continue
            break
        else:
            print('   all forward storages identical')
            for h, (a, b) in enumerate(zip(cur[4], ref[4])):
                if not torch.equal(a, b):
                    print('   head %d metrics differ at' % h, [(i, float(a[i]), float(b[i])) for i in range(20) if a[i] != b[i]])
        nb = [sid for sid in bwd if bwd[sid] != ref[2][sid]]
        print('   gradient storages differing: %d; flat_grad equal: %s' % (len(nb), cur[3] == ref[3]))
        break
else:
    print('%d runs identical' % runs)
