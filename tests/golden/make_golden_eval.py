#!/usr/bin/env python3
"""tests/golden/darknet_eval.npz: THE REFERENCE's inference path on BASELINE.json configs[3] (complex_yolov4.cfg, batch 32,
608x608; reference src/evaluate.py:32-45): ``model.eval()(imgs)`` followed by ``post_processing_v2``.

The seeded random-init fill of the other goldens saturates every sigmoid in eval mode (its BatchNorm running statistics have
nothing to do with the activations: half of the 22,743 rows per image come out with objectness exactly 1.0), which would
compare nothing.  So the model is CALIBRATED first, by the reference itself: one ``model.train()`` forward of a seeded
calibration batch with BatchNorm momentum 1.0 sets every running mean / variance to that batch's statistics; the eval outputs
then look like a network's (objectness spread around 0.5).  The calibrated running statistics are part of the golden -- the
device model loads them, so both sides run the same function.

Stored: the running statistics; every 97th decoded row of the batch-32 eval output; a confidence threshold picked in the
widest gap of the objectness values near the rank that leaves ~64 candidates per image; objectness of every row within 0.02 of
it; the reference's candidate rows (index + 10 values); the reference's post_processing_v2 detections on them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py           # ~2 minutes of CPU
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from tests.golden.make_golden import ROOT, import_reference  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402

B, S, SEED, CAL_B, CAL_SEED, PER_IMAGE, NMS_THRESH = 32, 608, 33, 4, 34, 64, 0.4


def main():
    d2p, _, _, _, ev = import_reference()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
    torch.manual_seed(0)
    model = d2p.Darknet(cfgfile=cfg, use_giou_loss=True)
    sd = model.state_dict()
    sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
    model.load_state_dict(sd)
    # calibration by the reference: train-mode forward, momentum 1 -> running statistics = this batch's statistics
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    with torch.no_grad():
        model(syn.bev_images(CAL_B, S, seed=CAL_SEED))
    model.eval()
    bn = [(k, v.clone()) for k, v in model.state_dict().items() if k.endswith('running_mean') or k.endswith('running_var')]
    out = {'bn_names': np.asarray([k for k, _ in bn]), 'bn_values': np.concatenate([v.numpy().reshape(-1) for _, v in bn]),
           'bn_sizes': np.asarray([v.numel() for _, v in bn])}
    with torch.no_grad():
        y = model(syn.bev_images(B, S, seed=SEED))
    print('eval output', tuple(y.shape), 'objectness percentiles', np.percentile(y[..., 6].numpy(), [1, 50, 99, 99.9]))
    out['out_shape'] = np.asarray(y.shape)
    out['out_rows'] = y[:, ::97].numpy()
    obj = y[..., 6].reshape(-1).numpy()
    order = np.sort(obj)[::-1]
    k0 = B * PER_IMAGE
    win = order[k0 - 200:k0 + 200]
    gaps = win[:-1] - win[1:]
    j = int(np.argmax(gaps))
    thr = np.float32((np.float64(win[j]) + np.float64(win[j + 1])) / 2)
    print('threshold %.7f in a gap of %.2e; candidates %d' % (thr, gaps[j], int((obj >= thr).sum())))
    out['conf_thresh'] = np.asarray([thr], dtype=np.float32)
    out['gap'] = np.asarray([gaps[j]], dtype=np.float32)
    near = np.nonzero(np.abs(obj - thr) < 0.02)[0]
    out['near_idx'], out['near_obj'] = near.astype(np.int64), obj[near]
    cand = np.nonzero(obj >= thr)[0]
    out['cand_idx'] = cand.astype(np.int64)                    # flat index b * N + row
    out['cand_rows'] = y.reshape(-1, y.shape[-1])[cand].numpy()
    dets = ev.post_processing_v2(y, conf_thresh=float(thr), nms_thresh=NMS_THRESH)
    out['nms_thresh'] = np.asarray([NMS_THRESH], dtype=np.float32)
    out['det_count'] = np.asarray([0 if d is None else d.shape[0] for d in dets])
    out['det'] = np.concatenate([d.numpy() for d in dets if d is not None], 0).astype(np.float32)
    print('detections per image: min %d median %d max %d' % (out['det_count'].min(), np.median(out['det_count']), out['det_count'].max()))
    np.savez_compressed(os.path.join(HERE, 'darknet_eval.npz'), **out)
    print('darknet_eval.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
