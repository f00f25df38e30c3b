"""Darknet cfg text -> list of block dicts.

Drop-in for the reference's ``parse_cfg`` (reference src/models/darknet_utils.py:17-47): same return
convention -- one dict per ``[section]``, key ``type`` holding the section name, every other key
a stripped string, ``[convolutional]`` blocks defaulting ``batch_normalize`` to 0 and a literal
``type=`` key inside a section renamed ``_type``.  Darknet ``.weights`` (de)serialisers are out of
scope (never called by train/evaluate).
"""

__all__ = ['parse_cfg', 'print_cfg']


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
