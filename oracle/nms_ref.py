"""ORACLE (test infrastructure).  Rotated NMS / post-processing as the reference does it on CPU.

Restates reference src/utils/evaluation_utils.py:
  iou_rotated_single_vs_multi_boxes_cpu :193-218   (float32 corners, float64 clip, eps 1e-16)
  nms_cpu                               :250-276   (class-agnostic greedy, keep IoU <= thr, eps 1e-12)
  post_processing_v2                    :321-357   (obj filter, score sort, same-class merge-NMS)
Ties in the score sort are implementation-defined in the reference (unstable argsort); the
oracle fixes them as "lower original index first" (SURVEY.md App. A #17).
"""
import numpy as np
import torch

from . import clip


def corners_np(boxes):
    """boxes[n,6] float32 (x,y,w,l,im,re) -> float32 corners [n,4,2] (float32 arithmetic throughout)."""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 6)
    x, y, w, l, im, re = b.T
    yaw = np.arctan2(im, re)
    c, s = np.cos(yaw), np.sin(yaw)
    out = np.zeros((b.shape[0], 4, 2), dtype=np.float32)
    out[:, 0, 0] = x - w / 2 * c - l / 2 * s
    out[:, 0, 1] = y - w / 2 * s + l / 2 * c
    out[:, 1, 0] = x - w / 2 * c + l / 2 * s
    out[:, 1, 1] = y - w / 2 * s - l / 2 * c
    out[:, 2, 0] = x + w / 2 * c + l / 2 * s
    out[:, 2, 1] = y + w / 2 * s - l / 2 * c
    out[:, 3, 0] = x + w / 2 * c - l / 2 * s
    out[:, 3, 1] = y + w / 2 * s + l / 2 * c
    return out


def iou_matrix(boxes_a, boxes_b, eps=1e-16):
    """float32 IoU matrix [na,nb].  The clip is float64; everything after it is float32 arithmetic, as
    in the reference where a python-float intersection meets float32 tensors / numpy-2 scalars."""
    a = np.asarray(boxes_a, dtype=np.float32).reshape(-1, 6)
    b = np.asarray(boxes_b, dtype=np.float32).reshape(-1, 6)
    inter = clip.inter_matrix(corners_np(a), corners_np(b))
    area_a = (a[:, 2] * a[:, 3])[:, None]
    area_b = (b[:, 2] * b[:, 3])[None, :]
    i32 = inter.astype(np.float32)
    return i32 / (((area_a + area_b) - i32) + np.float32(eps))


def greedy_nms(boxes, confs, nms_thresh=0.5):
    """Class-agnostic greedy rotated NMS -> kept indices (int64), highest confidence first."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 6)
    confs = np.asarray(confs).reshape(-1)
    order = np.argsort(-confs, kind='stable')
    iou = iou_matrix(boxes, boxes, eps=1e-12)
    alive = np.ones(len(order), dtype=bool)
    keep = []
    for pos, i in enumerate(order):
        if not alive[pos]:
            continue
        keep.append(int(i))
        rest = order[pos + 1:]
        alive[pos + 1:] &= iou[i, rest] <= nms_thresh
    return np.asarray(keep, dtype=np.int64)


def post_process_v2(prediction, conf_thresh=0.95, nms_thresh=0.4):
    """prediction[B,N,7+C] -> list of [K,9] float32 tensors (x,y,w,l,im,re,obj,cls_conf,cls_id) / None.
    Also returns, per image, the original row index of every emitted detection (for index parity)."""
    pred = torch.as_tensor(prediction).float()
    outs, idxs = [], []
    for img in pred:
        rows = torch.nonzero(img[:, 6] >= conf_thresh).reshape(-1)
        if rows.numel() == 0:
            outs.append(None); idxs.append(None)
            continue
        cand = img[rows]
        cls_conf, cls_id = cand[:, 7:].max(1)
        score = cand[:, 6] * cls_conf
        order = torch.from_numpy(np.argsort(-score.numpy(), kind='stable'))
        cand, cls_conf, cls_id, rows = cand[order], cls_conf[order], cls_id[order], rows[order]
        iou = torch.from_numpy(iou_matrix(cand[:, :6].numpy(), cand[:, :6].numpy()))
        alive = torch.ones(cand.shape[0], dtype=torch.bool)
        det, src = [], []
        for i in range(cand.shape[0]):
            if not alive[i]:
                continue
            grp = alive & (iou[i] > nms_thresh) & (cls_id == cls_id[i])
            wgt = cand[grp, 6:7]
            merged = (wgt * cand[grp, :6]).sum(0) / wgt.sum()
            det.append(torch.cat((merged, cand[i, 6:7], cls_conf[i:i + 1], cls_id[i:i + 1].float())))
            src.append(int(rows[i]))
            alive &= ~grp
            alive[i] = False  # guard: a degenerate box has self-IoU 0 and would never leave
        outs.append(torch.stack(det)); idxs.append(np.asarray(src, dtype=np.int64))
    return outs, idxs
