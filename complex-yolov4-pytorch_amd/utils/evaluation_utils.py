"""post_processing_v2 / nms_cpu / iou_rotated_single_vs_multi_boxes_cpu -- drop-ins for the NMS half of
reference src/utils/evaluation_utils.py (:193-218, :250-276, :321-357), running on the HIP device
(cy_pp2_select / cy_pp2_merge / cy_rnms_greedy / cy_riou_matrix).  Names keep the reference's ``_cpu``
suffix for drop-in imports although nothing here runs on the CPU.  The mAP bookkeeping of that file
(SURVEY.md section 8f row 2): get_batch_statistics_rotated_bbox (:152-190) takes its detection x target IoUs from the
device kernel (cy_riou_matrix) and keeps the short sequential claim loop on the host; ap_per_class / compute_ap
(:70-149) are host-side numpy, as in the reference.

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


def load_classes(path):
    """Class names, one per line (reference evaluation_utils.py:43-49; host-only helper kept for import compatibility)."""
    with open(path, 'r') as fp:
        return fp.read().split('\n')[:-1]


def post_processing(outputs, conf_thresh=0.95, nms_thresh=0.4):
    """The reference's first post-processing variant (evaluation_utils.py:279-318): per image keep the rows whose
    objectness x best class score exceeds ``conf_thresh``, greedy rotated NMS on that score (``nms_cpu`` = cy_rnms_greedy),
    rows (x, y, w, l, im, re, object_conf, score, class).  The reference indexes its 2-D objectness array with three
    subscripts (:305) and raises IndexError on every call; this is the function it evidently meant, kept so that
    ``from utils.evaluation_utils import post_processing`` (evaluate.py:20) resolves.  -> list of np.ndarray [K, 9] / None."""
    out = outputs.detach().cpu().numpy() if torch.is_tensor(outputs) else np.asarray(outputs)
    confs = out[:, :, 6:7] * out[:, :, 7:]
    max_conf, max_id = confs.max(axis=2), confs.argmax(axis=2)
    result = [None] * out.shape[0]
    for i in range(out.shape[0]):
        sel = max_conf[i] > conf_thresh
        if not sel.any():
            continue
        boxes, obj, score, cls = out[i, sel, :6], out[i, sel, 6], max_conf[i, sel], max_id[i, sel]
        keep = nms_cpu(boxes, score, nms_thresh=nms_thresh)
        if keep.size > 0:
            result[i] = np.concatenate((boxes[keep], obj[keep].reshape(-1, 1), score[keep].reshape(-1, 1),
                                        cls[keep].reshape(-1, 1).astype(out.dtype)), axis=-1)
    return result


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


def get_batch_statistics_rotated_bbox(outputs, targets, iou_threshold):
    """outputs: list (one per image) of [K,9] tensors (x,y,w,l,im,re,obj_conf,cls_score,cls_pred) or None, as returned by
    post_processing_v2; targets [T,8] = (image index, class, x,y,w,l,im,re) in the same pixel units.
    -> [[true_positives np.ndarray [K], pred_scores tensor [K], pred_labels tensor [K]], ...] for the images that have
    detections (reference evaluation_utils.py:152-190).  A detection claims the target of its highest rotated IoU (over
    all targets of the image, any class) when that IoU >= iou_threshold, its label occurs among the image's target
    labels and the target is still free; the walk over the detections stops when every target is claimed."""
    targets = torch.as_tensor(targets).float().cpu()
    metrics = []
    for i, output in enumerate(outputs):
        if output is None:
            continue
        output = torch.as_tensor(output).float()
        K = output.shape[0]
        tp = np.zeros(K)
        ann = targets[targets[:, 0] == i][:, 1:]
        if len(ann) > 0 and K > 0:
            iou = ops.riou_matrix(_dev(output[:, :6]).contiguous(), _dev(ann[:, 1:7]).contiguous(), 1e-16)
            best_iou, best_idx = iou.max(dim=1)
            best_iou, best_idx = best_iou.cpu().numpy(), best_idx.cpu().numpy()
            labels = output[:, -1].cpu().numpy()
            label_ok = np.isin(labels, ann[:, 0].numpy())
            claimed = set()
            for k in range(K):
                if len(claimed) == len(ann):
                    break
                if not label_ok[k]:
                    continue
                if best_iou[k] >= iou_threshold and int(best_idx[k]) not in claimed:
                    tp[k] = 1
                    claimed.add(int(best_idx[k]))
        metrics.append([tp, output[:, 6].cpu(), output[:, -1].cpu()])
    return metrics


def compute_ap(recall, precision):
    """Area under the monotone precision envelope over the recall steps (py-faster-rcnn AP; reference :127-149)."""
    mrec = np.concatenate(([0.0], np.asarray(recall, dtype=np.float64), [1.0]))
    mpre = np.concatenate(([0.0], np.asarray(precision, dtype=np.float64), [0.0]))
    for k in range(mpre.size - 1, 0, -1):
        mpre[k - 1] = max(mpre[k - 1], mpre[k])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    """tp, conf, pred_cls: per-detection arrays concatenated over the dataset; target_cls: class of every ground-truth box.
    -> (precision, recall, AP, f1, classes) per class present in target_cls (reference :70-124)."""
    tp, conf, pred_cls = np.asarray(tp), np.asarray(conf), np.asarray(pred_cls)
    target_cls = np.asarray(target_cls)
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = np.unique(target_cls)
    ap, p, r = [], [], []
    for c in classes:
        sel = pred_cls == c
        n_gt, n_p = (target_cls == c).sum(), sel.sum()
        if n_p == 0 and n_gt == 0:
            continue
        if n_p == 0 or n_gt == 0:
            ap.append(0); r.append(0); p.append(0)
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(), tp[sel].cumsum()
        recall_curve = tpc / (n_gt + 1e-16)
        precision_curve = tpc / (tpc + fpc)
        r.append(recall_curve[-1]); p.append(precision_curve[-1])
        ap.append(compute_ap(recall_curve, precision_curve))
    p, r, ap = np.array(p), np.array(r), np.array(ap)
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, classes.astype('int32')
