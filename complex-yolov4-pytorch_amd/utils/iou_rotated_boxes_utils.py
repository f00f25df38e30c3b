"""iou_pred_vs_target_boxes / iou_rotated_boxes_targets_vs_anchors / get_polygons_areas_fix_xy -- drop-ins for
reference src/utils/iou_rotated_boxes_utils.py (:64-142) on the HIP device (cy_riou_pairs, cy_riou_anchors).

The reference materialises shapely polygons on the host; here "polygons" are just the (w, l, im, re) rows on
the device and the float64 clip happens in the kernel.  ``iou_pred_vs_target_boxes`` keeps the reference's
return convention: (ious [n] detached, summed loss [1]) with the reference's partial gradient (App. A #11)."""
import torch

from .. import ops


class _PairLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, giou):
        ious, terms, g = ops.riou_pairs(pred.detach(), target.detach(), giou)
        ctx.save_for_backward(g)
        ctx.mark_non_differentiable(ious)
        return ious, terms.sum().reshape(1)

    @staticmethod
    def backward(ctx, _gi, gloss):
        (g,) = ctx.saved_tensors
        return g * gloss.reshape(1, 1), None, None


def iou_pred_vs_target_boxes(pred_boxes, target_boxes, GIoU=False, DIoU=False, CIoU=False):
    assert pred_boxes.size() == target_boxes.size(), "Unmatch size of pred_boxes and target_boxes"
    if DIoU or CIoU:
        raise NotImplementedError
    ops.check_device_tensor(pred_boxes, 'iou_pred_vs_target_boxes')
    return _PairLoss.apply(pred_boxes.float(), target_boxes.to(pred_boxes.device).float(), bool(GIoU))


def get_polygons_areas_fix_xy(boxes, fix_xy=100.):
    """boxes [n,4] = (w, l, im, re) -> (the same rows on the device standing in for the polygon list, areas)."""
    ops.check_device_tensor(boxes, 'get_polygons_areas_fix_xy')
    b = boxes.float()
    return b, b[:, 0] * b[:, 1]


def iou_rotated_boxes_targets_vs_anchors(anchors_polygons, anchors_areas, targets_polygons, targets_areas):
    """-> [nA, nT] float32 IoU of shapes sharing a centre (position-free anchor matching)."""
    return ops.riou_anchors(anchors_polygons, targets_polygons)
