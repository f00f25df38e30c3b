#!/usr/bin/env python3
"""tests/golden/darknet_big.npz: THE REFERENCE ITSELF (imported from /root/reference/src as in make_golden.py) on
BASELINE.json's own shapes -- complex_yolov4.cfg, 608x608 batch 16 (configs[1]), 1024x1024 batch 2 (configs[4]'s
resolution) and a batch of two 1216x1216 MOSAIC canvases (configs[2], "mosaic aug on": each canvas assembled from four
608x608 maps by the reference's own KittiDataset.load_mosaic under a fixed ``random`` seed, as in make_golden_aug.py) --
one fp32 train step each on the seeded synthetic batch.  Holds outputs only (loss, every 97th decoded row, the 18 metrics
per head, per-tensor gradient norms and heads, BatchNorm running statistics heads); takes a few minutes of CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_big.py            # everything
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_big.py --only b2_1216m   # (re)make one case, keep the rest
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from tests.golden.make_golden import METRIC_KEYS, ROOT, import_reference  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402

CASES = (('b16_608', 16, 608, 21), ('b2_1024', 2, 1024, 21), ('b2_1216m', 2, 1216, 60))
MOSAIC_SEEDS = (3, 11)        # ``random.seed`` per canvas of the b2_1216m case (random centre, kitti_dataset.py:128-131)


def mosaic_batch(seed0, B=2, half=608, nt=6):
    """B canvases from the reference's load_mosaic: canvas b is made of the synthetic tiles seeded seed0 + 10 b + k,
    k = 0..3 (6 targets each), under random.seed(MOSAIC_SEEDS[b]).  -> (images [B,3,2 half,2 half], targets [n,8])."""
    import random
    from tests.golden.make_golden import REF
    sys.path.insert(0, REF)
    cwd = os.getcwd(); os.chdir(REF)
    try:
        from data_process.kitti_dataset import KittiDataset
    finally:
        os.chdir(cwd)
    canvases, rows = [], []
    for b in range(B):
        ds = object.__new__(KittiDataset)
        ds.img_size, ds.random_padding, ds.mosaic_border, ds.num_samples = half, True, [-half // 2, -half // 2], 4
        tl = [syn.bev_images(1, half, seed=seed0 + 10 * b + k)[0] for k in range(4)]
        tt = [syn.targets(1, nt, half, seed=seed0 + 10 * b + k) for k in range(4)]
        calls = []

        def fake(index, tl=tl, tt=tt, calls=calls):
            k = len(calls); calls.append(index)
            return 'tile%d' % k, tl[k], tt[k].clone()
        ds.load_img_with_targets = fake
        random.seed(MOSAIC_SEEDS[b])
        _, canvas, targets = ds.load_mosaic(0)
        targets = targets.clone(); targets[:, 0] = b
        canvases.append(canvas); rows.append(targets)
    return torch.stack(canvases).float(), torch.cat(rows, 0).float()


def main():
    d2p = import_reference()[0]
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
    out = {}
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None
    path = os.path.join(HERE, 'darknet_big.npz')
    if only:
        old = np.load(path, allow_pickle=False)
        out = {k: old[k] for k in old.files if not k.startswith(only + '_')}
    for tag, B, S, seed in CASES:
        if only and tag != only:
            continue
        torch.manual_seed(0)
        model = d2p.Darknet(cfgfile=cfg, use_giou_loss=True)
        sd = model.state_dict()
        sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
        model.load_state_dict(sd)
        model.train()
        if tag.endswith('m'):
            x, tg = mosaic_batch(seed, B, S // 2)
            out[tag + '_targets'] = tg.numpy()
            out[tag + '_canvas_rows'] = x[:, :, ::97, ::89].numpy()
        else:
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
    np.savez_compressed(path, **out)
    print('darknet_big.npz', len(out), 'arrays')


if __name__ == '__main__':
    main()
