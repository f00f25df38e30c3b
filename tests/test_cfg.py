"""The generated cfgs and the drop-in parser (CPU)."""
import os

import pytest

from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
from oracle.darknet_ref import DarknetRef

CFG = os.path.join(os.path.dirname(__file__), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg')
REF_CFG = '/root/reference/src/config/cfg'

USED = {'net': ['channels'], 'convolutional': ['batch_normalize', 'filters', 'size', 'stride', 'pad', 'activation'],
        'maxpool': ['size', 'stride'], 'route': ['layers', 'groups', 'group_id'], 'shortcut': ['from', 'activation'],
        'upsample': ['stride'], 'yolo': ['mask', 'anchors', 'classes', 'num', 'scale_x_y', 'ignore_thresh']}


def _norm(k, v):
    v = str(v).replace(' ', '')
    if k in ('ignore_thresh', 'scale_x_y'):
        return float(v)
    return v


@pytest.mark.parametrize('name,nblocks,nconv,nparams', [('complex_yolov4.cfg', 163, 110, 63959226),
                                                       ('complex_yolov4_tiny.cfg', 39, 21, 5883356)])
def test_generated_cfg_structure(name, nblocks, nconv, nparams):
    blocks = parse_cfg(os.path.join(CFG, name))
    assert len(blocks) == nblocks and blocks[0]['type'] == 'net'
    net = DarknetRef(blocks)
    convs = [m for m in net.mods if m['type'] == 'convolutional']
    assert len(convs) == nconv
    pshapes, _ = net.param_shapes()
    total = 0
    for s in pshapes.values():
        n = 1
        for d in s:
            n *= d
        total += n
    assert total == nparams          # SURVEY.md section 8a row A [probe]


def test_parse_cfg_conventions(tmp_path):
    p = tmp_path / 'x.cfg'
    p.write_text('# c\n[net]\nwidth = 32\n\n[convolutional]\nfilters=4\n[cost]\ntype=sse\n')
    b = parse_cfg(str(p))
    assert b[0] == {'type': 'net', 'width': '32'}
    assert b[1]['batch_normalize'] == 0 and b[1]['filters'] == '4'
    assert b[2] == {'type': 'cost', '_type': 'sse'}


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('name', ['complex_yolov4.cfg', 'complex_yolov4_tiny.cfg'])
def test_generated_cfg_matches_reference_on_consumed_keys(name):
    ours, ref = parse_cfg(os.path.join(CFG, name)), parse_cfg(os.path.join(REF_CFG, name))
    assert len(ours) == len(ref)
    for i, (a, b) in enumerate(zip(ours, ref)):
        assert a['type'] == b['type'], i
        for k in USED[a['type']]:
            assert (k in a) == (k in b), (i, k)
            if k in a:
                assert _norm(k, a[k]) == _norm(k, b[k]), (i, k)
