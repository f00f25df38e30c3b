"""Where does the fp16 backward lose accuracy?  Compare d(loss)/d(pre-BN conv output) of every BN conv of the mini
cfg (engine grad storages after backward) with the oracle's, in backward order."""
import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
from oracle import darknet_ref
from tests.util import mini_cfg_path
cfg = mini_cfg_path()
x, tg = syn.bev_images(2, 64, seed=4, sparsity=0.5), syn.targets(2, 3, 64, seed=4, collide=True)
net = darknet_ref.DarknetRef(parse_cfg(cfg)); ps, bs = net.param_shapes()
params = {k: v.requires_grad_(True) for k, v in syn.fill_state_dict(ps).items()}
keep = {}
o_ref, l_ref, _ = net.forward(params, x, tg, True, True, syn.fill_state_dict(bs), keep=keep)
for v in keep.values(): v.retain_grad()
l_ref.sum().backward()
for dt in ('f32', 'f16'):
    m = Darknet(cfg, use_giou_loss=True, dtype=dt)
    sd = m.state_dict(); sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point}); m.load_state_dict(sd)
    m.cuda().train(); loss, out = m(x.cuda(), tg.cuda()); loss.backward()
    eng = list(m._engines.values())[0]
    print('==', dt)
    for rec in reversed(eng.plan.convs):
        if not rec['bn']: continue
        i = rec['idx']
        g = eng.view(rec['out'], grad=True).to_nchw().cpu()
        ref = keep[('raw', i)].grad
        raw = eng.view(rec['raw']).to_nchw().cpu()
        act = eng.view(rec['out']).to_nchw().cpu()
        ref_act = keep[eng.plan.fused_into.get(i, i)]
        e = float((g - ref).abs().max()) / float(ref.abs().max())
        e2 = float((g - ref).norm()) / float(ref.norm())
        print('module %2d %-6s dRaw: max-rel %.3e  l2-rel %.3e | fwd raw l2-rel %.2e act l2-rel %.2e | |dRaw|max %.2e' % (
            i, rec['act'], e, e2, float((raw - keep[('raw', i)].detach()).norm()) / float(keep[('raw', i)].norm()),
            float((act - ref_act.detach()).norm()) / float(ref_act.norm()), float(ref.abs().max())))
