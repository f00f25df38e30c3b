#!/usr/bin/env python3
"""Emit the Darknet cfg files the hot path consumes.

The reference ships its model definitions as Darknet cfg text
(/root/reference/src/config/cfg/complex_yolov4.cfg, complex_yolov4_tiny.cfg, complex_yolov3.cfg, complex_yolov3_tiny.cfg).  Only a handful of
keys are consumed by the model builder (reference src/models/darknet2pytorch.py:245-397); this
script writes cfgs carrying exactly those keys from a compact description of the four
architectures (CSPDarknet53 + SPP + PANet, the CSP-tiny variant, Darknet-53 + FPN, and the v3-tiny stack), so the files under
complex-yolov4-pytorch_amd/config/cfg/ are generated artefacts, not copies.

tests/test_cfg.py checks the generated graphs against SURVEY.md Appendix B and, when
/root/reference is present, block-for-block against the reference cfgs on the consumed keys.
"""
import os
import sys

ANCHORS_V4 = "11, 15, 0, 10, 24, 0, 11, 25, 0, 23, 49, 0, 23, 55, 0, 24, 53, 0, 24, 60, 0, 27, 63, 0, 29, 74, 0"
ANCHORS_TINY = "11, 15, 0, 11, 25, 0, 23, 49, 0, 23, 55, 0, 24, 53, 0, 25, 61, 0"
ANCHORS_V3 = "11,14,-3.14, 11,14,0, 11,14,3.14, 11,25,-3.14, 11,25,0, 11,25,3.14, 23,51,-3.14, 23,51,0, 23,51,3.14"
ANCHORS_V3_TINY = "16,16,-3.14, 16,16,0, 16,16,3.14, 23,51,-3.14, 23,51,0, 23,51,3.14"


class Cfg:
    def __init__(self):
        self.sections = []

    def add(self, kind, **kv):
        self.sections.append((kind, kv))
        return len(self.sections) - 2  # module index (the [net] section is not a module)

    def conv(self, filters, size, stride=1, act="mish", bn=1):
        kv = {}
        if bn:
            kv["batch_normalize"] = 1
        kv.update(filters=filters, size=size, stride=stride, pad=1, activation=act)
        return self.add("convolutional", **kv)

    def route(self, *layers, **extra):
        return self.add("route", layers=",".join(str(l) for l in layers), **extra)

    def shortcut(self, frm):
        return self.add("shortcut", **{"from": frm, "activation": "linear"})

    def maxpool(self, size, stride):
        return self.add("maxpool", stride=stride, size=size)

    def upsample(self, stride=2):
        return self.add("upsample", stride=stride)

    def yolo(self, mask, anchors, num, scale_x_y):
        kv = dict(mask=",".join(str(m) for m in mask), anchors=anchors, classes=3, num=num, ignore_thresh=.7)
        if scale_x_y is not None:
            kv["scale_x_y"] = scale_x_y
        return self.add("yolo", **kv)

    def text(self):
        out = []
        for kind, kv in self.sections:
            out.append("[%s]" % kind)
            for k, v in kv.items():
                out.append("%s=%s" % (k, v))
            out.append("")
        return "\n".join(out)


def csp_stage(c, ch, n_res, first=False):
    """One CSP stage of CSPDarknet53: stride-2 conv, split, n residual units, merge."""
    c.conv(ch, 3, 2)
    half = ch if first else ch // 2
    c.conv(half, 1)
    c.route(-2)
    c.conv(half, 1)
    for _ in range(n_res):
        c.conv(ch // 2 if first else half, 1)
        c.conv(half, 3)
        c.shortcut(-3)
    c.conv(half, 1)
    c.route(-1, -(4 + 3 * n_res))
    c.conv(ch, 1)


def five(c, lo, hi, act="leaky"):
    for i in range(5):
        c.conv(lo if i % 2 == 0 else hi, 1 if i % 2 == 0 else 3, act=act)


def build_v4():
    c = Cfg()
    c.add("net", width=608, height=608, channels=3)
    c.conv(32, 3, 1)
    csp_stage(c, 64, 1, first=True)
    csp_stage(c, 128, 2)
    csp_stage(c, 256, 8)
    csp_stage(c, 512, 8)
    csp_stage(c, 1024, 4)
    # SPP neck
    c.conv(512, 1, act="leaky"); c.conv(1024, 3, act="leaky"); c.conv(512, 1, act="leaky")
    c.maxpool(5, 1); c.route(-2); c.maxpool(9, 1); c.route(-4); c.maxpool(13, 1)
    c.route(-1, -3, -5, -6)
    c.conv(512, 1, act="leaky"); c.conv(1024, 3, act="leaky"); c.conv(512, 1, act="leaky")
    # PANet top-down
    c.conv(256, 1, act="leaky"); c.upsample(); c.route(85); c.conv(256, 1, act="leaky"); c.route(-1, -3)
    five(c, 256, 512)
    c.conv(128, 1, act="leaky"); c.upsample(); c.route(54); c.conv(128, 1, act="leaky"); c.route(-1, -3)
    five(c, 128, 256)
    # head, stride 8
    c.conv(256, 3, act="leaky"); c.conv(30, 1, act="linear", bn=0)
    c.yolo((0, 1, 2), ANCHORS_V4, 9, 1.2)
    # bottom-up, stride 16
    c.route(-4); c.conv(256, 3, 2, act="leaky"); c.route(-1, -16)
    five(c, 256, 512)
    c.conv(512, 3, act="leaky"); c.conv(30, 1, act="linear", bn=0)
    c.yolo((3, 4, 5), ANCHORS_V4, 9, 1.1)
    # stride 32
    c.route(-4); c.conv(512, 3, 2, act="leaky"); c.route(-1, -37)
    five(c, 512, 1024)
    c.conv(1024, 3, act="leaky"); c.conv(30, 1, act="linear", bn=0)
    c.yolo((6, 7, 8), ANCHORS_V4, 9, 1.05)
    return c


def tiny_block(c, ch):
    c.conv(ch, 3, act="leaky")
    c.route(-1, groups=2, group_id=1)
    c.conv(ch // 2, 3, act="leaky"); c.conv(ch // 2, 3, act="leaky")
    c.route(-1, -2)
    c.conv(ch, 1, act="leaky")
    c.route(-6, -1)
    c.maxpool(2, 2)


def build_tiny():
    c = Cfg()
    c.add("net", width=416, height=416, channels=3)
    c.conv(32, 3, 2, act="leaky"); c.conv(64, 3, 2, act="leaky")
    tiny_block(c, 64); tiny_block(c, 128); tiny_block(c, 256)
    c.conv(512, 3, act="leaky"); c.conv(256, 1, act="leaky"); c.conv(512, 3, act="leaky")
    c.conv(30, 1, act="linear", bn=0)
    c.yolo((3, 4, 5), ANCHORS_TINY, 6, 1.05)
    c.route(-4); c.conv(128, 1, act="leaky"); c.upsample(); c.route(-1, 23)
    c.conv(256, 3, act="leaky"); c.conv(30, 1, act="linear", bn=0)
    c.yolo((0, 1, 2), ANCHORS_TINY, 6, 1.05)
    return c


def build_v3():
    """Darknet-53 + FPN (reference complex_yolov3.cfg): leaky everywhere, residual units of (1x1, 3x3, shortcut)."""
    c = Cfg()
    c.add("net", width=608, height=608, channels=3)
    c.conv(32, 3, 1, act="leaky")
    for ch, n_res in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        c.conv(ch, 3, 2, act="leaky")
        for _ in range(n_res):
            c.conv(ch // 2, 1, act="leaky"); c.conv(ch, 3, act="leaky"); c.shortcut(-3)
    for ch, mask, lateral in ((512, (6, 7, 8), 61), (256, (3, 4, 5), 36), (128, (0, 1, 2), None)):
        for _ in range(3):
            c.conv(ch, 1, act="leaky"); c.conv(2 * ch, 3, act="leaky")
        c.conv(30, 1, act="linear", bn=0)
        c.yolo(mask, ANCHORS_V3, 9, "1.")
        if lateral is not None:
            c.route(-4); c.conv(ch // 2, 1, act="leaky"); c.upsample(); c.route(-1, lateral)
    return c


def build_v3_tiny():
    """reference complex_yolov3_tiny.cfg: six conv + max-pool steps, the last pool being size 2 / STRIDE 1 (the reference's
    MaxPoolDark, darknet2pytorch.py:30-59), two heads."""
    c = Cfg()
    c.add("net", width=608, height=608, channels=3)
    for ch in (16, 32, 64, 128, 256):
        c.conv(ch, 3, act="leaky"); c.maxpool(2, 2)
    c.conv(512, 3, act="leaky"); c.maxpool(2, 1)
    c.conv(1024, 3, act="leaky"); c.conv(256, 1, act="leaky"); c.conv(512, 3, act="leaky")
    c.conv(30, 1, act="linear", bn=0)
    c.yolo((3, 4, 5), ANCHORS_V3_TINY, 6, None)
    c.route(-4); c.conv(128, 1, act="leaky"); c.upsample(); c.route(-1, 8)
    c.conv(256, 3, act="leaky"); c.conv(30, 1, act="linear", bn=0)
    c.yolo((0, 1, 2), ANCHORS_V3_TINY, 6, None)
    return c


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "complex-yolov4-pytorch_amd", "config", "cfg")
    os.makedirs(out, exist_ok=True)
    for name, b in (("complex_yolov4.cfg", build_v4), ("complex_yolov4_tiny.cfg", build_tiny), ("complex_yolov3.cfg", build_v3),
                    ("complex_yolov3_tiny.cfg", build_v3_tiny)):
        with open(os.path.join(out, name), "w") as f:
            f.write("# generated by tools/gen_cfg.py -- do not edit\n" + b().text())
        print("wrote", name)


if __name__ == "__main__":
    sys.exit(main())
