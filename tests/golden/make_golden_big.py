#!/usr/bin/env python3
"""tests/golden/darknet_big.npz: THE REFERENCE ITSELF (imported from /root/reference/src as in make_golden.py) on
BASELINE.json's own shapes -- complex_yolov4.cfg, 608x608 batch 16 (configs[1]) and 1024x1024 batch 2 (configs[4]'s
resolution) -- one fp32 train step each on the seeded synthetic batch.  Holds outputs only (loss, every 97th decoded row,
the 18 metrics per head, per-tensor gradient norms, BatchNorm running statistics heads); takes a few minutes of CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_big.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from tests.golden.make_golden import METRIC_KEYS, ROOT, import_reference  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402

CASES = (('b16_608', 16, 608, 21), ('b2_1024', 2, 1024, 21))


def main():
    d2p = import_reference()[0]
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
    out = {}
    for tag, B, S, seed in CASES:
        torch.manual_seed(0)
        model = d2p.Darknet(cfgfile=cfg, use_giou_loss=True)
        sd = model.state_dict()
        sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
        model.load_state_dict(sd)
        model.train()
        x, tg = syn.bev_images(B, S, seed=seed), syn.targets(B, 6, S, seed=seed)
        loss, outputs = model(x, tg)
        loss.sum().backward()
        key = tag + '_'
        out[key + 'loss'] = loss.detach().numpy().reshape(-1)
        out[key + 'out_rows'] = outputs[:, ::97].detach().numpy()
        out[key + 'out_shape'] = np.asarray(outputs.shape)
        out[key + 'metrics'] = np.asarray([[yl_.metrics[k] for k in METRIC_KEYS] for yl_ in model.yolo_layers])
        out[key + 'names'] = np.asarray([n for n, _ in model.named_parameters()])
        out[key + 'grad_norm'] = np.asarray([float(p.grad.double().norm()) for _, p in model.named_parameters()])
        out[key + 'grad_head'] = np.stack([p.grad.reshape(-1)[:8].numpy() for _, p in model.named_parameters()])
        bn = [(k, v) for k, v in model.state_dict().items() if k.endswith('running_mean') or k.endswith('running_var')]
        out[key + 'bn_names'] = np.asarray([k for k, _ in bn])
        out[key + 'bn_head'] = np.stack([v[:8].numpy() for _, v in bn])
        print(tag, 'loss', out[key + 'loss'], flush=True)
    np.savez_compressed(os.path.join(HERE, 'darknet_big.npz'), **out)
    print('darknet_big.npz', len(out), 'arrays')


if __name__ == '__main__':
    main()
