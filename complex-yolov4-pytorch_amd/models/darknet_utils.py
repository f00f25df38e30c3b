"""Darknet cfg text -> list of block dicts.

Drop-in for the reference's ``parse_cfg`` (reference src/models/darknet_utils.py:17-47): same return
convention -- one dict per ``[section]``, key ``type`` holding the section name, every other key
a stripped string, ``[convolutional]`` blocks defaulting ``batch_normalize`` to 0 and a literal
``type=`` key inside a section renamed ``_type``.  Also the Darknet ``.weights`` (de)serialisers
load_conv_bn / load_conv / save_conv_bn / save_conv (reference :199-246).
"""
import torch

__all__ = ['parse_cfg', 'print_cfg', 'load_conv_bn', 'load_conv', 'save_conv_bn', 'save_conv']


def parse_cfg(cfgfile):
    blocks = []
    current = None
    with open(cfgfile, 'r') as fp:
        for raw in fp:
            line = raw.rstrip()
            if not line or line.startswith('#'):
                continue
            if line.startswith('['):
                if current:
                    blocks.append(current)
                current = {'type': line.lstrip('[').rstrip(']')}
                if current['type'] == 'convolutional':
                    current['batch_normalize'] = 0
                continue
            key, value = line.split('=')
            key = key.strip()
            current['_type' if key == 'type' else key] = value.strip()
    if current:
        blocks.append(current)
    return blocks


def print_cfg(blocks, width=None, height=None):
    """Human-readable layer table (role of reference darknet_utils.py:50-196; format is ours)."""
    from .graph import trace_shapes
    net = blocks[0]
    w = int(width or net.get('width', 608))
    h = int(height or net.get('height', 608))
    print('idx   type            out C x H x W')
    for i, (kind, shape) in enumerate(trace_shapes(blocks, h, w)):
        print('%4d  %-14s  %d x %d x %d' % (i, kind, shape[0], shape[1], shape[2]))


# ---- Darknet .weights interop (reference darknet_utils.py:199-261) -------------------------
# File layout: int32[5] header (major, minor, revision, seen lo, seen hi as the reference reads it: 5 int32, seen =
# header[3]) followed by flat float32: per conv block with batch_normalize  bn.bias, bn.weight, bn.running_mean,
# bn.running_var, conv.weight ; without  conv.bias, conv.weight.  The master parameters of this package are ordinary
# fp32 OIHW tensors (the packed fp16 device matrices are rebuilt from them every step), so interop is a flat copy.
def _take(buf, start, t):
    n = t.numel()
    if start + n > buf.size:
        # the reference fails here with torch's RuntimeError (shape mismatch in copy_)
        raise RuntimeError('weights file too short: need %d floats at offset %d, file has %d' % (n, start, buf.size))
    t.copy_(torch.from_numpy(buf[start:start + n].copy()).reshape(t.shape))
    return start + n


def load_conv_bn(buf, start, conv_model, bn_model):
    with torch.no_grad():
        start = _take(buf, start, bn_model.bias.data)
        start = _take(buf, start, bn_model.weight.data)
        start = _take(buf, start, bn_model.running_mean)
        start = _take(buf, start, bn_model.running_var)
        start = _take(buf, start, conv_model.weight.data)
    return start


def load_conv(buf, start, conv_model):
    with torch.no_grad():
        start = _take(buf, start, conv_model.bias.data)
        start = _take(buf, start, conv_model.weight.data)
    return start


def _put(fp, t):
    t.detach().to('cpu', torch.float32).contiguous().numpy().tofile(fp)


def save_conv_bn(fp, conv_model, bn_model):
    for t in (bn_model.bias.data, bn_model.weight.data, bn_model.running_mean, bn_model.running_var, conv_model.weight.data):
        _put(fp, t)


def save_conv(fp, conv_model):
    _put(fp, conv_model.bias.data)
    _put(fp, conv_model.weight.data)
