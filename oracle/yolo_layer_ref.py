"""ORACLE (test infrastructure).  YoloLayer head: decode, target assignment, loss, metrics.

Restates reference src/models/yolo_layer.py:
  decode              :144-193     (sigmoid x,y,conf,cls; exp(w,h).clamp(max=1e3)*anchor; im,re raw)
  grid/anchor tables  :53-67       (stride := img_size / grid_size)
  build_targets       :69-142      (best anchor by position-free rotated IoU, obj/noobj masks, ignore
                                    threshold, tx..tre, multi-hot tcls, class_mask, pred-vs-target IoU)
  loss + 18 metrics   :199-251
Index collisions (two targets in one (b,anchor,gj,gi)) follow CPU ``index_put`` order: the LAST
target in row order wins for scalar maps, class one-hots accumulate.
"""
import torch
import torch.nn.functional as F

from . import rotated_iou

LOSS_SCALES = dict(noobj=100.0, obj=1.0, lgiou=3.54, leular=3.54, lobj=64.3, lcls=37.4)


def decode(x, anchors, num_classes, img_size):
    """x[B, A*(7+C), G, G] -> dict of decoded maps [B,A,G,G,...] and output [B, A*G*G, 7+C]."""
    B, _, G, _ = x.shape
    A = len(anchors)
    stride = img_size / G
    p = x.view(B, A, num_classes + 7, G, G).permute(0, 1, 3, 4, 2).contiguous()
    sx, sy = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1])
    raw_w, raw_h, im, re = p[..., 2], p[..., 3], p[..., 4], p[..., 5]
    conf, cls = torch.sigmoid(p[..., 6]), torch.sigmoid(p[..., 7:])
    cols = torch.arange(G, dtype=torch.float32).view(1, 1, 1, G)
    rows = torch.arange(G, dtype=torch.float32).view(1, 1, G, 1)
    sa = torch.tensor([(aw / stride, ah / stride, i, r) for aw, ah, i, r in anchors], dtype=torch.float32)
    aw = sa[:, 0].view(1, A, 1, 1)
    ah = sa[:, 1].view(1, A, 1, 1)
    boxes = torch.stack((sx + cols, sy + rows, torch.exp(raw_w).clamp(max=1e3) * aw,
                         torch.exp(raw_h).clamp(max=1e3) * ah, im, re), -1)
    out = torch.cat((boxes[..., :4].reshape(B, -1, 4) * stride, boxes[..., 4:].reshape(B, -1, 2),
                     conf.reshape(B, -1, 1), cls.reshape(B, -1, num_classes)), -1)
    return dict(sx=sx, sy=sy, raw_w=raw_w, raw_h=raw_h, im=im, re=re, conf=conf, cls=cls, boxes=boxes,
                scaled_anchors=sa, output=out, stride=stride)


def assign_targets(d, targets, ignore_thresh, use_giou):
    boxes, cls = d['boxes'], d['cls']
    B, A, G, _, C = cls.shape
    sa = d['scaled_anchors']
    z = lambda *s: torch.zeros(*s, dtype=torch.float32)
    obj = torch.zeros(B, A, G, G, dtype=torch.bool)
    noobj = torch.ones(B, A, G, G, dtype=torch.bool)
    m = dict(class_mask=z(B, A, G, G), iou_scores=z(B, A, G, G), tx=z(B, A, G, G), ty=z(B, A, G, G),
             tw=z(B, A, G, G), th=z(B, A, G, G), tim=z(B, A, G, G), tre=z(B, A, G, G), tcls=z(B, A, G, G, C))
    giou_loss = torch.zeros(1)
    nT = targets.shape[0]
    if nT > 0:
        b = targets[:, 0].long()
        lab = targets[:, 1].long()
        tb = torch.cat((targets[:, 2:6] * G, targets[:, 6:8]), -1)
        ious_at = rotated_iou.anchors_vs_targets_iou(sa, tb[:, 2:6])  # [A, nT]
        best = ious_at.argmax(0)
        gi, gj = tb[:, 0].long(), tb[:, 1].long()
        for k in range(nT):  # row order == CPU index_put order
            bb, a, j, i = int(b[k]), int(best[k]), int(gj[k]), int(gi[k])
            obj[bb, a, j, i] = True
            noobj[bb, a, j, i] = False
            noobj[bb, ious_at[:, k] > ignore_thresh, j, i] = False
            m['tx'][bb, a, j, i] = tb[k, 0] - tb[k, 0].floor()
            m['ty'][bb, a, j, i] = tb[k, 1] - tb[k, 1].floor()
            m['tw'][bb, a, j, i] = torch.log(tb[k, 2] / sa[a, 0] + 1e-16)
            m['th'][bb, a, j, i] = torch.log(tb[k, 3] / sa[a, 1] + 1e-16)
            m['tim'][bb, a, j, i] = tb[k, 4]
            m['tre'][bb, a, j, i] = tb[k, 5]
            m['tcls'][bb, a, j, i, int(lab[k])] = 1.0
            m['class_mask'][bb, a, j, i] = float(int(cls[bb, a, j, i].argmax()) == int(lab[k]))
        ious, giou_loss = rotated_iou.pred_vs_target(boxes[b, best, gj, gi], tb, giou=use_giou)
        for k in range(nT):
            m['iou_scores'][int(b[k]), int(best[k]), int(gj[k]), int(gi[k])] = ious[k]
        giou_loss = giou_loss / nT
    m.update(obj=obj, noobj=noobj, tconf=obj.float(), giou_loss=giou_loss)
    return m


def head_forward(x, targets, anchors, num_classes, ignore_thresh, img_size, use_giou_loss):
    """Returns (output[B,A*G*G,7+C], total_loss or 0, metrics dict of python floats or {})."""
    d = decode(x, anchors, num_classes, img_size)
    if targets is None:
        return d['output'], 0, {}
    m = assign_targets(d, targets, ignore_thresh, use_giou_loss)
    obj, noobj, tconf = m['obj'], m['noobj'], m['tconf']
    mse = lambda a, t: F.mse_loss(a[obj], t[obj])
    loss_x, loss_y = mse(d['sx'], m['tx']), mse(d['sy'], m['ty'])
    loss_w, loss_h = mse(d['raw_w'], m['tw']), mse(d['raw_h'], m['th'])
    loss_im, loss_re = mse(d['im'], m['tim']), mse(d['re'], m['tre'])
    unit = ((1. - torch.sqrt(d['im'][obj] ** 2 + d['re'][obj] ** 2)) ** 2).mean()
    loss_eular = loss_im + loss_re + unit
    conf = d['conf']
    l_conf_obj = F.binary_cross_entropy(conf[obj], tconf[obj])
    l_conf_noobj = F.binary_cross_entropy(conf[noobj], tconf[noobj])
    loss_cls = F.binary_cross_entropy(d['cls'][obj], m['tcls'][obj])
    S = LOSS_SCALES
    if use_giou_loss:
        loss_obj = l_conf_obj + l_conf_noobj
        total = m['giou_loss'] * S['lgiou'] + loss_eular * S['leular'] + loss_obj * S['lobj'] + loss_cls * S['lcls']
    else:
        loss_obj = S['obj'] * l_conf_obj + S['noobj'] * l_conf_noobj
        total = loss_x + loss_y + loss_w + loss_h + loss_eular + loss_obj + loss_cls
    conf50 = (conf > 0.5).float()
    iou50 = (m['iou_scores'] > 0.5).float()
    iou75 = (m['iou_scores'] > 0.75).float()
    det = conf50 * m['class_mask'] * tconf
    f = lambda t: float(t.detach().reshape(-1)[0]) if t.numel() else float('nan')
    metrics = {
        'loss': f(total), 'iou_score': f(m['iou_scores'][obj].mean()), 'giou_loss': f(m['giou_loss']),
        'loss_x': f(loss_x), 'loss_y': f(loss_y), 'loss_w': f(loss_w), 'loss_h': f(loss_h),
        'loss_eular': f(loss_eular), 'loss_im': f(loss_im), 'loss_re': f(loss_re),
        'loss_obj': f(loss_obj), 'loss_cls': f(loss_cls),
        'cls_acc': f(100 * m['class_mask'][obj].mean()),
        'recall50': f((iou50 * det).sum() / (obj.sum() + 1e-16)),
        'recall75': f((iou75 * det).sum() / (obj.sum() + 1e-16)),
        'precision': f((iou50 * det).sum() / (conf50.sum() + 1e-16)),
        'conf_obj': f(conf[obj].mean()), 'conf_noobj': f(conf[noobj].mean()),
    }
    return d['output'], total, metrics
