"""Where do two repeats of the same deterministic-mode train step first differ?  Compares every activation / gradient
storage of the engine after two identical forward+backward passes.  usage: python tools/det_probe.py [dtype=f16] [B=16]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
torch.manual_seed(0)
model = Darknet(cfg, use_giou_loss=True, dtype=dtype, deterministic=True).cuda().train()
x, tg = syn.bev_images(B, 608, seed=5).cuda(), syn.targets(B, 6, 608, seed=5).cuda()
snaps = []
for it in range(int(os.environ.get("DET_RUNS", "3"))):
    model.zero_grad(set_to_none=True)
    loss, out = model(x, tg)
    eng = next(iter(model._engines.values()))
    torch.cuda.synchronize()
    fwd = {sid: t.clone() for sid, t in eng.act.items()}
    loss.backward()
    torch.cuda.synchronize()
    bwd = {sid: t.clone() for sid, t in eng.gact.items()}
    snaps.append((float(loss), fwd, bwd, model.flat_grad.clone(), {k: v.clone() for k, v in eng.bnvec.items()}))
    print('run', it, 'loss %.6f' % float(loss), 'tiles fwd', sorted(set(eng._fwd_tile.values())), 'dgrad', sorted(set(eng._dgrad_tile.values())))
plan = eng.plan
for a, b in [(i, i + 1) for i in range(len(snaps) - 1)]:
    print('--- run %d vs run %d' % (a, b))
    first = None
    for rec in plan.fwd:
        if rec['op'] != 'conv':
            continue
        for key in ('raw', 'out'):
            ref = rec.get(key)
            if ref is None:
                continue
            sid = ref.st.sid
            if sid in snaps[a][1] and not torch.equal(snaps[a][1][sid], snaps[b][1][sid]):
                d = (snaps[a][1][sid].float() - snaps[b][1][sid].float()).abs().max()
                print('  fwd differs: module %d (%s) %s cin %d cout %d k%d s%d H%d tile %s: max |d| %.3e' % (
                    rec['idx'], key, ref.st, rec['cin'], rec['cout'], rec['ks'], rec['stride'], rec['H'],
                    eng._fwd_tile.get(rec['idx']), float(d)))
                first = first or rec['idx']
                break
        if first is not None:
            break
    if first is None:
        print('  forward storages identical')
        nb = sum(1 for sid in snaps[a][2] if not torch.equal(snaps[a][2][sid], snaps[b][2][sid]))
        print('  backward storages differing: %d of %d;  flat_grad equal: %s' % (nb, len(snaps[a][2]), torch.equal(snaps[a][3], snaps[b][3])))
        if nb:
            for bb in plan.bwd:
                if bb['op'] != 'conv_bwd':
                    continue
                rec = bb['fwd']
                sid = rec['out'].st.sid
                if sid in snaps[a][2] and not torch.equal(snaps[a][2][sid], snaps[b][2][sid]):
                    print('  first differing gradient storage (backward order): module %d %s' % (rec['idx'], rec['out'].st))
                    break
names, off = [], 0
for n, p_ in model.named_parameters():
    names.append((n, off, p_.numel()))
    off += p_.numel()
a, b = snaps[1][3], snaps[2][3]
bad = [(n, float((a[o:o + c] - b[o:o + c]).abs().max()), float(a[o:o + c].abs().max())) for n, o, c in names if not torch.equal(a[o:o + c], b[o:o + c])]
print('parameter gradients differing between run 1 and 2: %d of %d' % (len(bad), len(names)))
for n, d, m in bad[:12]:
    print('   %-40s max |d| %.3e (max |g| %.3e)' % (n, d, m))
kinds = {}
for n, d, m in bad:
    k = n.split('.')[-2][:4] + '.' + n.split('.')[-1]
    kinds[k] = kinds.get(k, 0) + 1
print(kinds)
