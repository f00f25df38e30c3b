#!/usr/bin/env python3
"""tests/golden/aug.npz: the reference's image-space augmentation run UNMODIFIED on seeded synthetic maps --
transformation.Horizontal_Flip / Cutout (src/data_process/transformation.py:376-437) and KittiDataset.load_mosaic
(src/data_process/kitti_dataset.py:123-173; the dataset object is created without its file-reading __init__ and its
``load_img_with_targets`` hands out the synthetic tiles).  Images are frozen as sha256 of their float32 bytes plus a sparse
sample (they are pure copies: bit-exactness is the test), targets in full.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_aug.py
"""
import hashlib
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from tests.golden.make_golden import REF  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy(), dtype=np.float32).tobytes()).hexdigest()


def tiles(seed0, n=4, size=608, nt=6):
    return ([syn.bev_images(1, size, seed=seed0 + k)[0] for k in range(n)],
            [syn.targets(1, nt, size, seed=seed0 + k) for k in range(n)])


def cutout_seed(tr, img, tg):
    """first numpy seed for which the reference's Cutout(3, 0.25) drops at least one of the targets"""
    for s in range(1000):
        np.random.seed(s)
        _, t = tr(img.clone(), tg.clone())
        if 0 < t.shape[0] < tg.shape[0]:
            return s
    raise RuntimeError('no seed drops a target')


def main():
    sys.modules['cv2'] = types.ModuleType('cv2')
    sys.path.insert(0, REF)
    cwd = os.getcwd(); os.chdir(REF)
    try:
        from data_process import transformation as T
        from data_process.kitti_dataset import KittiDataset
    finally:
        os.chdir(cwd)
    out = {}
    img, tg = syn.bev_images(1, 608, seed=51)[0], syn.targets(1, 6, 608, seed=51)
    # flip
    np.random.seed(1)
    fi, ft = T.Horizontal_Flip(p=1.0)(img.clone(), tg.clone())
    out['flip_sha'] = np.asarray(sha(fi)); out['flip_rows'] = fi[:, ::61, ::53].numpy(); out['flip_targets'] = ft.numpy()
    # cut-out
    cut = T.Cutout(n_holes=3, ratio=0.25, fill_value=0.5, p=1.0)
    s = cutout_seed(cut, img, tg)
    np.random.seed(s)
    ci, ct = cut(img.clone(), tg.clone())
    out['cut_seed'] = np.asarray([s]); out['cut_sha'] = np.asarray(sha(ci)); out['cut_rows'] = ci[:, ::61, ::53].numpy()
    out['cut_targets'] = ct.numpy()
    # flip + cut-out through Compose, 12 holes (more than one kernel launch's worth)
    comp = T.Compose([T.Horizontal_Flip(p=1.0), T.Cutout(n_holes=12, ratio=0.1, fill_value=0.0, p=1.0)], p=1.0)
    np.random.seed(77)
    pi, pt = comp(img.clone(), tg.clone())
    out['comp_sha'] = np.asarray(sha(pi)); out['comp_targets'] = pt.numpy()
    # mosaic, fixed centre and random centres
    for tag, rp, seed in (('mosaic_fixed', False, 3), ('mosaic_rand_a', True, 3), ('mosaic_rand_b', True, 11)):
        ds = object.__new__(KittiDataset)
        ds.img_size, ds.random_padding, ds.mosaic_border, ds.num_samples = 608, rp, [-304, -304], 4
        tl, tt = tiles(60)
        calls = []

        def fake(index, tl=tl, tt=tt, calls=calls):
            k = len(calls); calls.append(index)
            return 'tile%d' % k, tl[k], tt[k].clone()
        ds.load_img_with_targets = fake
        random.seed(seed)
        _, canvas, targets = ds.load_mosaic(0)
        out[tag + '_sha'] = np.asarray(sha(canvas)); out[tag + '_rows'] = canvas[:, ::97, ::89].numpy()
        out[tag + '_targets'] = targets.numpy(); out[tag + '_seed'] = np.asarray([seed])
        print(tag, tuple(canvas.shape), targets.shape, float(canvas.sum()))
    np.savez_compressed(os.path.join(HERE, 'aug.npz'), **out)
    print('aug.npz', len(out), 'arrays; cut-out seed', s, 'targets kept', ct.shape[0], 'of', tg.shape[0])


if __name__ == '__main__':
    main()
