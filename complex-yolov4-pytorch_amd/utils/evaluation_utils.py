"""post_processing_v2 / nms_cpu / iou_rotated_single_vs_multi_boxes_cpu -- drop-ins for the NMS half of
reference src/utils/evaluation_utils.py (:193-218, :250-276, :321-357), running on the HIP device
(cy_pp2_select / cy_pp2_merge / cy_rnms_greedy / cy_riou_matrix).  Names keep the reference's ``_cpu``
suffix for drop-in imports although nothing here runs on the CPU.  The mAP bookkeeping of that file
(ap_per_class, compute_ap, get_batch_statistics_rotated_bbox) is host-side numpy and out of scope
(SURVEY.md section 2 row 5).

Ties in the score sort are implementation-defined in the reference (unstable argsort, App. A #17); here equal
scores keep the lower original row first.  A degenerate (zero-area) box has self-IoU 0 and makes the
reference loop forever (section 8a row I); here every detection always belongs to its own group."""
import numpy as np
import torch

from .. import ops


def _dev(t):
    t = torch.as_tensor(t)
    return t if t.is_cuda else t.to('cuda')


def post_processing_v2(prediction, conf_thresh=0.95, nms_thresh=0.4):
    """prediction [B, N, 7+C] (CPU or device) -> list of [K, 9] CPU tensors
    (x, y, w, l, im, re, object_conf, class_score, class_pred) or None, as the reference returns."""
    pred = _dev(prediction).float()
    outs, _ = ops.pp2(pred, conf_thresh, nms_thresh)
    return [None if o is None else o.cpu() for o in outs]


def post_processing_v2_device(prediction, conf_thresh=0.95, nms_thresh=0.4):
    """Same, results stay on the device; also returns the source row of every detection."""
    return ops.pp2(_dev(prediction).float(), conf_thresh, nms_thresh)


def nms_cpu(boxes, confs, nms_thresh=0.5):
    """boxes [K,6] (x,y,w,l,im,re), confs [K] -> np.ndarray of kept indices (highest confidence first)."""
    b = _dev(np.ascontiguousarray(boxes, dtype=np.float32) if not torch.is_tensor(boxes) else boxes)
    c = _dev(np.ascontiguousarray(confs, dtype=np.float32) if not torch.is_tensor(confs) else confs)
    return ops.rnms_greedy(b.float(), c.float(), nms_thresh).cpu().numpy().astype(np.int64)


def iou_rotated_single_vs_multi_boxes_cpu(single_box, multi_boxes):
    """[6], [K,6] -> float32 tensor [K] of rotated IoUs (CPU tensor, like the reference)."""
    s = _dev(torch.as_tensor(np.asarray(single_box, dtype=np.float32) if not torch.is_tensor(single_box) else single_box))
    m = _dev(torch.as_tensor(np.asarray(multi_boxes, dtype=np.float32) if not torch.is_tensor(multi_boxes) else multi_boxes))
    return ops.riou_matrix(s.reshape(1, 6).float(), m.reshape(-1, 6).float(), 1e-16)[0].cpu()
