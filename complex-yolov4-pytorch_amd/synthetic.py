"""Deterministic synthetic BEV batches and parameter fills.

There is no KITTI data in the container; every parity test, the golden generator and bench.py
draw their inputs from here so that the reference, the oracle and the HIP path see identical
tensors.  Target rows follow the reference's collate format
(reference src/data_process/kitti_dataset.py:216-233): (sample_idx, class, x, y, w, l, im, re),
x,y,w,l normalised to [0,1], (im, re) = (sin yaw, cos yaw).
"""
import math
import zlib

import torch

# per-class (w, l) of KITTI boxes in BEV at 50 m / 608 px, normalised (Car, Pedestrian, Cyclist)
_CLASS_WL = ((0.038, 0.084), (0.018, 0.022), (0.018, 0.041))


def bev_images(batch, size, seed=0, sparsity=0.08):
    """[B,3,S,S] float32 in [0,1]; BEV-like: a pixel is non-empty with probability ``sparsity``."""
    g = torch.Generator().manual_seed(1000 + seed)
    vals = torch.rand(batch, 3, size, size, generator=g)
    if sparsity >= 1.0:
        return vals
    occ = (torch.rand(batch, 1, size, size, generator=g) < sparsity).float()
    return vals * occ


def targets(batch, per_image=6, size=608, seed=0, collide=False):
    """[nT,8] float32 target rows; nT = batch*per_image (+2 when ``collide``: two extra boxes that
    share a grid cell with box 0 at every head, to exercise last-writer-wins)."""
    g = torch.Generator().manual_seed(2000 + seed)
    rows = []
    for b in range(batch):
        for _ in range(per_image):
            cls = int(torch.randint(0, 3, (1,), generator=g))
            u = torch.rand(5, generator=g)
            w0, l0 = _CLASS_WL[cls]
            x = 0.05 + 0.9 * float(u[0])
            y = 0.05 + 0.9 * float(u[1])
            w = w0 * (0.85 + 0.3 * float(u[2]))
            l = l0 * (0.85 + 0.3 * float(u[3]))
            yaw = (2 * float(u[4]) - 1) * math.pi
            lim = 1 - 0.5 / size
            rows.append([b, cls, min(x, lim), min(y, lim), w, l, math.sin(yaw), math.cos(yaw)])
    if collide:
        b0 = rows[0]
        rows.append([b0[0], (int(b0[1]) + 2) % 3, b0[2] + 1e-4, b0[3] + 1e-4, b0[4] * 1.05, b0[5] * 0.97,
                     math.sin(0.3), math.cos(0.3)])
        rows.append([b0[0], int(b0[1]), b0[2] + 2e-4, b0[3] - 1e-4, b0[4], b0[5], math.sin(-1.1), math.cos(-1.1)])
    return torch.tensor(rows, dtype=torch.float32)


def fill_tensor(name, shape, seed=0):
    """Deterministic value for the state-dict entry ``name`` (independent of construction order)."""
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7fffffff)
    if name.endswith('running_var'):
        return 0.5 + torch.rand(shape, generator=g)
    if name.endswith('running_mean'):
        return 0.1 * torch.randn(shape, generator=g)
    if '.bn' in name and name.endswith('.weight'):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if name.endswith('.bias'):
        return 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)


def fill_state_dict(shapes, seed=0):
    return {k: fill_tensor(k, tuple(s), seed) for k, s in shapes.items()}


def nms_predictions(batch, n_rows, n_above, seed=0, n_centres=50, size=608.0, conf_thresh=0.5):
    """[B,n_rows,10] float32 decoded-prediction tensors for the NMS path: ``n_above`` rows per image
    have objectness >= conf_thresh with boxes clustered around ``n_centres`` centres (so suppression
    happens); scores are tie-free."""
    g = torch.Generator().manual_seed(3000 + seed)
    pred = torch.zeros(batch, n_rows, 10)
    pred[..., 0:2] = torch.rand(batch, n_rows, 2, generator=g) * size
    pred[..., 2] = 10 + 20 * torch.rand(batch, n_rows, generator=g)
    pred[..., 3] = 20 + 40 * torch.rand(batch, n_rows, generator=g)
    yaw = (2 * torch.rand(batch, n_rows, generator=g) - 1) * math.pi
    pred[..., 4], pred[..., 5] = torch.sin(yaw), torch.cos(yaw)
    pred[..., 6] = 0.4 * torch.rand(batch, n_rows, generator=g)
    pred[..., 7:] = torch.rand(batch, n_rows, 3, generator=g)
    for b in range(batch):
        rows = torch.randperm(n_rows, generator=g)[:n_above]
        cen = torch.rand(n_centres, 2, generator=g) * (size - 100) + 50
        cyaw = (2 * torch.rand(n_centres, generator=g) - 1) * math.pi
        ccls = torch.randint(0, 3, (n_centres,), generator=g)
        which = torch.randint(0, n_centres, (n_above,), generator=g)
        pred[b, rows, 0:2] = cen[which] + 6 * torch.randn(n_above, 2, generator=g)
        jyaw = cyaw[which] + 0.15 * torch.randn(n_above, generator=g)
        pred[b, rows, 2] = 22 + 4 * torch.rand(n_above, generator=g)
        pred[b, rows, 3] = 48 + 8 * torch.rand(n_above, generator=g)
        pred[b, rows, 4], pred[b, rows, 5] = torch.sin(jyaw), torch.cos(jyaw)
        pred[b, rows, 6] = conf_thresh + (1 - conf_thresh) * torch.rand(n_above, generator=g)
        cls = 0.2 * torch.rand(n_above, 3, generator=g)
        cls[torch.arange(n_above), ccls[which]] = 0.6 + 0.4 * torch.rand(n_above, generator=g)
        pred[b, rows, 7:] = cls
    return pred
