"""YoloLayer -- drop-in for reference src/models/yolo_layer.py (same constructor, ``forward(x, targets,
img_size, use_giou_loss) -> (output, loss | 0)`` and ``.metrics`` with the reference's 18 keys), backed by
the fused HIP head kernels (cy_yolo_decode / cy_yolo_loss).

Inside ``Darknet`` the engine calls the kernels directly on the NHWC fp32 logits of the head conv; this
module's own ``forward`` serves standalone use on an NCHW tensor, as the reference's tests would.
Behavioural notes kept from the reference: every target is assigned in every head (#1);
``scale_x_y`` is stored and unused (#3); an index collision keeps the last target in row order (#7);
nT == 0 yields NaN means (#6).  A target that falls outside the grid (x or y == 1.0, #8) raises IndexError
in the reference; here the row is skipped and counted in ``metrics_raw[19]``.
"""
import torch
import torch.nn as nn

from .. import ops

METRIC_KEYS = ('loss', 'iou_score', 'giou_loss', 'loss_x', 'loss_y', 'loss_w', 'loss_h', 'loss_eular', 'loss_im',
               'loss_re', 'loss_obj', 'loss_cls', 'cls_acc', 'recall50', 'recall75', 'precision', 'conf_obj',
               'conf_noobj')


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layer, targets, img_size, use_giou):
        B, _, G, _ = x.shape
        A, C = layer.num_anchors, layer.num_classes
        logits = x.detach().float().permute(0, 2, 3, 1).contiguous()
        out = torch.empty(B, A * G * G, 7 + C, device=x.device)
        ops.yolo_decode(logits, B, G, A, C, layer.anchors, img_size, out, A * G * G, 0)
        ws = torch.empty(ops.yolo_loss_workspace(B, G, A, C, targets.shape[0]), dtype=torch.uint8, device=x.device)
        metrics = torch.zeros(20, device=x.device)
        dl = torch.empty_like(logits)
        ops.yolo_loss(logits, B, G, A, C, targets.float().contiguous(), layer.anchors, img_size, layer.ignore_thresh,
                      use_giou, ws, metrics, dl)
        ctx.save_for_backward(dl)
        layer._metrics_dev = metrics
        ctx.mark_non_differentiable(out)
        loss = metrics[0:1].clone() if use_giou else metrics[0].clone()   # shapes [1] / [] as the reference (App. A #10)
        return out, loss

    @staticmethod
    def backward(ctx, _gout, gloss):
        (dl,) = ctx.saved_tensors
        return (dl * gloss.reshape(())).permute(0, 3, 1, 2), None, None, None, None


class YoloLayer(nn.Module):
    """Yolo layer (reference yolo_layer.py:27-51 for the constructor contract)."""

    def __init__(self, num_classes, anchors, stride, scale_x_y, ignore_thresh):
        super(YoloLayer, self).__init__()
        self.num_classes = num_classes
        self.anchors = anchors                  # [(w, h, sin yaw, cos yaw)] in input pixels
        self.num_anchors = len(anchors)
        self.stride = stride
        self.scale_x_y = scale_x_y
        self.ignore_thresh = ignore_thresh
        self.noobj_scale, self.obj_scale = 100, 1
        self.lgiou_scale, self.leular_scale, self.lobj_scale, self.lcls_scale = 3.54, 3.54, 64.3, 37.4
        self.seen = 0
        self.grid_size = 0
        self.img_size = 0
        self._metrics_dev = None

    @property
    def metrics_raw(self):
        """The 20 device floats of cy_yolo_loss (18 metrics, nObj, rejected-target count) on the host."""
        return None if self._metrics_dev is None else self._metrics_dev.detach().cpu()

    @property
    def metrics(self):
        """dict with the reference's 18 keys (yolo_layer.py:232-251); one D2H copy per access instead of the
        reference's 18 ``.item()`` synchronisations per head per step."""
        raw = self.metrics_raw
        if raw is None:
            return {}
        vals = raw.tolist()
        return {k: vals[i] for i, k in enumerate(METRIC_KEYS)}

    def forward(self, x, targets=None, img_size=608, use_giou_loss=False):
        ops.check_device_tensor(x, 'YoloLayer')
        self.img_size = img_size
        self.grid_size = x.size(3)
        self.stride = img_size / self.grid_size
        if targets is None:
            B, _, G, _ = x.shape
            logits = x.detach().float().permute(0, 2, 3, 1).contiguous()
            out = torch.empty(B, self.num_anchors * G * G, 7 + self.num_classes, device=x.device)
            ops.yolo_decode(logits, B, G, self.num_anchors, self.num_classes, self.anchors, img_size, out,
                            self.num_anchors * G * G, 0)
            return out, 0
        out, loss = _HeadFn.apply(x, self, targets.to(x.device), img_size, bool(use_giou_loss))
        return out, loss
