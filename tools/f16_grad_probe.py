import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
from oracle import darknet_ref
from tests.util import grad_rel_errors, mini_cfg_path
cfg = mini_cfg_path()
x, tg = syn.bev_images(2, 64, seed=4, sparsity=0.5), syn.targets(2, 3, 64, seed=4, collide=True)
net = darknet_ref.DarknetRef(parse_cfg(cfg)); ps, bs = net.param_shapes()
params = {k: v.requires_grad_(True) for k, v in syn.fill_state_dict(ps).items()}
o_ref, l_ref, _ = net.forward(params, x, tg, True, True, syn.fill_state_dict(bs)); l_ref.sum().backward()
for ls in (1.0, 64.0, 1024.0, 16384.0):
    m = Darknet(cfg, use_giou_loss=True, dtype='f16', loss_scale=ls)
    sd = m.state_dict(); sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point}); m.load_state_dict(sd)
    m.cuda().train(); loss, out = m(x.cuda(), tg.cuda()); loss.backward()
    errs = grad_rel_errors([(n, p.grad.cpu()) for n, p in m.named_parameters()], {k: v.grad for k, v in params.items()})
    v = np.asarray(list(errs.values()))
    print('loss_scale %g: loss rel %.2e grad err median %.3e max %.3e' % (ls, abs(float(loss.detach())-float(l_ref.detach()))/float(l_ref.detach()), np.median(v), v.max()))
    names = list(errs)
    print('   per-module:', ' '.join('%s:%.2f' % (n.split('.')[1]+n.split('.')[2][:2], errs[n]) for n in names if n.endswith('weight') and 'conv' in n))
