"""Horizontal_Flip / Cutout / mosaic on the HIP device -- drop-ins for the image-space augmentation of the reference
(src/data_process/transformation.py:376-437 and KittiDataset.load_mosaic, src/data_process/kitti_dataset.py:123-173),
SURVEY.md section 8f row 1.  The classes keep the reference's constructor arguments and ``(img, targets) -> (img, targets)``
call; random draws are made on the host with the same ``np.random`` / ``random`` calls in the same order as the
reference, so a seeded run picks the same flips, holes and mosaic centres; pixels and target rows move on the device
(cy_bev_flip_cutout / cy_bev_mosaic / cy_bev_mosaic_targets) and come out bit-identical.

``img``: float32 [3, H, W] device tensor (what ``kitti_bev_utils.makeBVFeature`` returns for a device input);
``targets``: float32 [n, 8] = (sample, class, x, y, w, l, im, re), host or device.  Every transform returns the targets on
the image's device whether or not its random gate fired (a collate that concatenates samples must not see a host / device
mix that depends on the draws); when nothing fires and the targets already live there, the very same objects come back."""
import random

import numpy as np
import torch

from .. import ops


def _on(img, targets):
    """targets on the image's device as contiguous float32 (the same object when it already is)."""
    ops.check_device_tensor(img, 'augmentation')
    return targets.to(img.device).float().contiguous()


class Compose(object):
    """reference transformation.py:315-324: transforms applied with a common probability gate per call."""

    def __init__(self, transforms, p=1.0):
        self.transforms = transforms
        self.p = p

    def __call__(self, img, targets):
        targets = _on(img, targets)
        if np.random.random() <= self.p:
            for t in self.transforms:
                img, targets = t(img, targets)
        return img, targets


class OneOf(object):
    """reference transformation.py:327-337: one of the transforms, drawn uniformly."""

    def __init__(self, transforms, p=1.0):
        self.transforms = transforms
        self.p = p

    def __call__(self, img, targets):
        targets = _on(img, targets)
        if np.random.random() <= self.p:
            choice = np.random.randint(low=0, high=len(self.transforms))
            img, targets = self.transforms[choice](img, targets)
        return img, targets


class Horizontal_Flip(object):
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img, targets):
        targets = _on(img, targets)
        if np.random.random() <= self.p:
            img, _ = ops.bev_flip_cutout(img.float().contiguous(), True, [], 0.0, targets)
        return img, targets


class Cutout(object):
    """n_holes square-ish patches of side ratio * size set to fill_value; targets whose centre falls into a patch are
    dropped (reference transformation.py:388-437, draws included: one ``random()`` for p, then ``randint(h)``,
    ``randint(w)`` per hole)."""

    def __init__(self, n_holes, ratio, fill_value=0., p=1.0):
        assert 0. <= fill_value <= 1., "the fill value is in a range of 0 to 1"
        self.n_holes, self.ratio, self.fill_value, self.p = n_holes, ratio, fill_value, p

    def __call__(self, img, targets):
        targets = _on(img, targets)
        if np.random.random() <= self.p:
            h, w = img.size(1), img.size(2)
            h_cutout, w_cutout = int(self.ratio * h), int(self.ratio * w)
            holes = []
            for _ in range(self.n_holes):
                y, x = np.random.randint(h), np.random.randint(w)
                holes.append((int(np.clip(y - h_cutout // 2, 0, h)), int(np.clip(y + h_cutout // 2, 0, h)),
                              int(np.clip(x - w_cutout // 2, 0, w)), int(np.clip(x + w_cutout // 2, 0, w))))
            # the kernel takes at most 8 holes per launch
            keep_all = None
            for k in range(0, max(len(holes), 1), 8):
                img, keep = ops.bev_flip_cutout(img.float().contiguous(), False, holes[k:k + 8], self.fill_value, targets,
                                                want_keep=True)
                if keep is not None:
                    keep_all = keep if keep_all is None else (keep_all & keep)
            if keep_all is not None:
                targets = targets[keep_all.bool()]
        return img, targets


def mosaic_geometry(img_size, h, w, yc, xc):
    """Destination / source rectangles of the four tiles around the centre (yc, xc) on the 2*img_size canvas
    (kitti_dataset.py:141-155) -> [(x1a, y1a, x2a, y2a, x1b, y1b)] * 4 and [(padw, padh)] * 4."""
    S2 = img_size * 2
    rects = []
    x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
    rects.append((x1a, y1a, x2a, y2a, w - (x2a - x1a), h - (y2a - y1a)))
    x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, S2), yc
    rects.append((x1a, y1a, x2a, y2a, 0, h - (y2a - y1a)))
    x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(S2, yc + h)
    rects.append((x1a, y1a, x2a, y2a, w - (x2a - x1a), 0))
    x1a, y1a, x2a, y2a = xc, yc, min(xc + w, S2), min(S2, yc + h)
    rects.append((x1a, y1a, x2a, y2a, 0, 0))
    pads = [(r[0] - r[4], r[1] - r[5]) for r in rects]
    return rects, pads


def make_mosaic(tiles, targets_list, img_size, random_padding=False, mosaic_border=None):
    """The image half of KittiDataset.load_mosaic: four (img [3,h,w], targets [n,8]) pairs -> ([3, 2 S, 2 S] canvas filled
    with 0.5, targets renormalised to the canvas and concatenated).  The centre is drawn like the reference does
    (``random.uniform`` per axis) when ``random_padding`` is set, else it is (S, S)."""
    border = mosaic_border if mosaic_border is not None else [-img_size // 2, -img_size // 2]
    if random_padding:
        yc, xc = [int(random.uniform(-x, 2 * img_size + x)) for x in border]
    else:
        yc, xc = img_size, img_size
    C, h, w = tiles[0].shape
    rects, pads = mosaic_geometry(img_size, h, w, yc, xc)
    canvas = ops.bev_mosaic(tiles, img_size, rects, 0.5)
    dev = canvas.device
    rows = [t.to(dev).float() for t in targets_list]
    tile_of = torch.cat([torch.full((t.shape[0],), k, dtype=torch.int32, device=dev) for k, t in enumerate(rows)])
    targets = torch.cat(rows, 0).contiguous()
    ops.bev_mosaic_targets(targets, tile_of, h, w, pads, img_size)
    return canvas, targets
