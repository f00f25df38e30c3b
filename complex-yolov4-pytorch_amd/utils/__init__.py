"""Host-side mirrors of the reference utils: evaluation (NMS, mAP), rotated IoU, training helpers."""
