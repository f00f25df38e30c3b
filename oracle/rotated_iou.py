"""ORACLE (test infrastructure).  Rotated-box IoU / GIoU as the reference computes it.

Restates, in torch-CPU float32 with explicit loops:
  * corner generation             reference src/utils/iou_rotated_boxes_utils.py:34-61
  * anchor-vs-target IoU          reference src/utils/iou_rotated_boxes_utils.py:64-95   (float64 clip)
  * pred-vs-target IoU / GIoU     reference src/utils/iou_rotated_boxes_utils.py:98-142
  * the float32 polygon clip      reference src/utils/cal_intersection_rotated_boxes.py:16-96
including the reference's behaviours a clean implementation would not have:
  #0  the clip stops at the first target edge that rejects the whole running polygon and then
      returns the area of the polygon clipped SO FAR (disjoint pairs get a non-zero area);
  #11 edge-crossing points are constants for autograd, ``ious`` is returned detached;
  #12 inside <=> a*x+b*y+c <= 0, crossing <=> s*t < 0 (strict).
Gradients come from torch autograd over this restatement, so they have the reference's partial
gradient by construction.
"""
import math

import numpy as np
import torch

from . import clip


def box_corners(x, y, w, l, yaw):
    """[n] float32 tensors -> [n,4,2]; order front-left, rear-left, rear-right, front-right."""
    c, s = torch.cos(yaw), torch.sin(yaw)
    hw_c, hw_s = w / 2 * c, w / 2 * s
    hl_c, hl_s = l / 2 * c, l / 2 * s
    fl = torch.stack((x - hw_c - hl_s, y - hw_s + hl_c), -1)
    rl = torch.stack((x - hw_c + hl_s, y - hw_s - hl_c), -1)
    rr = torch.stack((x + hw_c + hl_s, y + hw_s - hl_c), -1)
    fr = torch.stack((x + hw_c - hl_s, y + hw_s + hl_c), -1)
    return torch.stack((fl, rl, rr, fr), 1)


def shape_corners_fixed_centre(wlir, centre=100.0):
    """(w, l, im, re)[n,4] -> corners [n,4,2] of the box placed at (centre, centre), and w*l."""
    w, l, im, re = wlir.t()
    cx = torch.full_like(w, centre)
    return box_corners(cx, cx, w, l, torch.atan2(im, re)), w * l


def anchors_vs_targets_iou(anchor_wlir, target_wlir):
    """[nA,4],[nT,4] -> [nA,nT] float32 IoU of shapes sharing a centre (position-free matching)."""
    ac, aa = shape_corners_fixed_centre(anchor_wlir.float())
    tc, ta = shape_corners_fixed_centre(target_wlir.float())
    inter = clip.inter_matrix(ac.detach().numpy(), tc.detach().numpy())
    out = torch.zeros(ac.shape[0], tc.shape[0], dtype=torch.float32)
    for a in range(ac.shape[0]):
        for t in range(tc.shape[0]):
            i = float(inter[a, t])
            # python float / float32 tensor, as in the reference (float32 result)
            out[a, t] = i / (aa[a] + ta[t] - i + 1e-16)
    return out


def _shoelace(vs):
    n = len(vs)
    acc = None
    for i in range(n):
        j = (i + 1) % n
        term = vs[i][0] * vs[j][1] - vs[i][1] * vs[j][0]
        acc = term if acc is None else acc + term
    return acc.abs() * 0.5


def clip_area_refsem(subject, clipper):
    """float32 clip of quad ``subject`` [4,2] by the 4 edges of quad ``clipper`` [4,2] with the
    reference's exact control flow.  Returns a 0-d tensor (autograd-connected to the kept
    subject vertices) or the python float 0.0."""
    poly = [subject[i] for i in range(4)]
    for e in range(4):
        if len(poly) <= 2:
            break
        p, q = clipper[e], clipper[(e + 1) % 4]
        a = q[1] - p[1]
        b = p[0] - q[0]
        c = q[0] * p[1] - q[1] * p[0]
        vals = [a * v[0] + b * v[1] + c for v in poly]
        kept = []
        n = len(poly)
        for i in range(n):
            j = (i + 1) % n
            if bool(vals[i] <= 0):
                kept.append(poly[i])
            if bool(vals[i] * vals[j] < 0):
                s, t = poly[i], poly[j]
                a2 = t[1] - s[1]
                b2 = s[0] - t[0]
                c2 = t[0] * s[1] - t[1] * s[0]
                wdet = a * b2 - b * a2
                kept.append(torch.stack(((b * c2 - c * b2) / wdet, (c * a2 - a * c2) / wdet)).detach())
        if not kept:
            break  # reference quirk: keeps the polygon clipped so far
        poly = kept
    if len(poly) <= 2:
        return 0.0
    return _shoelace(poly)


def hull_indices(pts):
    """Indices of the convex-hull vertices of pts[n,2] (float64 monotone chain, CCW)."""
    p = np.asarray(pts, dtype=np.float64)
    order = sorted(range(len(p)), key=lambda i: (p[i, 0], p[i, 1]))

    def cross(o, a, b):
        return (p[a, 0] - p[o, 0]) * (p[b, 1] - p[o, 1]) - (p[a, 1] - p[o, 1]) * (p[b, 0] - p[o, 0])

    lower, upper = [], []
    for i in order:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], i) <= 0:
            lower.pop()
        lower.append(i)
    for i in reversed(order):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], i) <= 0:
            upper.pop()
        upper.append(i)
    return lower[:-1] + upper[:-1]


def pred_vs_target(pred, target, giou=False):
    """pred[n,6], target[n,6] rows (x,y,w,l,im,re) -> (ious[n] detached float32, loss_sum[1]).

    giou=True : float32 clip with reference semantics + hull area, loss += 1-(IoU-(C-U)/C)
    giou=False: float64 exact clip (no grad), loss += 1-IoU (grad only through pred w*l)."""
    assert pred.shape == target.shape
    tx, ty, tw, tl, tim, tre = target.t()
    t_c = box_corners(tx, ty, tw, tl, torch.atan2(tim, tre))
    t_area = tw * tl
    px, py, pw, pl, pim, pre = pred.t()
    p_c = box_corners(px, py, pw, pl, torch.atan2(pim, pre))
    p_area = pw * pl
    ious = []
    loss = torch.zeros(1, dtype=torch.float32)
    for k in range(pred.shape[0]):
        if giou:
            inter = clip_area_refsem(p_c[k], t_c[k])
        else:
            inter = clip.inter_area(p_c[k].detach().numpy(), t_c[k].detach().numpy())
        union = p_area[k] + t_area[k] - inter
        iou = inter / (union + 1e-16)
        if giou:
            both = torch.cat((p_c[k], t_c[k]), 0)
            hv = hull_indices(both.detach().numpy())
            c_area = _shoelace([both[i] for i in hv])
            loss = loss + (1. - (iou - (c_area - union) / (c_area + 1e-16)))
        else:
            loss = loss + (1. - iou)
        ious.append(float(iou.detach()) if torch.is_tensor(iou) else float(iou))
    return torch.tensor(ious, dtype=torch.float32), loss
