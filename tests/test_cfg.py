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


def _mini_model():
    import warnings
    from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
    from tests.util import mini_cfg_path
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f32')


def test_darknet_weights_file_matches_reference(golden, tmp_path):
    """Darknet.load_weights against the reference's own loader (tests/golden/weights.npz: checksums of every state-dict
    entry after the reference loaded the same synthetic file), the short-file error, the block-boundary prefix, and the
    save_weights round trip."""
    import numpy as np
    import torch
    from tests.golden.make_golden import weights_file
    g = golden('weights')
    n = int(g['n_floats'][0])
    for tag, count in (('full', n), ('prefix', int(g['prefix_floats'][0]))):
        path = str(tmp_path / (tag + '.weights'))
        weights_file(path, count)
        m = _mini_model()
        for p_ in m.parameters():
            p_.data.fill_(0.25)
        used = m.load_weights(path)
        assert used == count and int(m.seen) == int(g[tag + '_seen'][0]) == 12345
        sd = m.state_dict()
        keys = [k for k in g.files if k.startswith(tag + '/')]
        assert len(keys) == len([k for k in sd if 'num_batches_tracked' not in k])
        for k in keys:
            v = sd[k[len(tag) + 1:]].double().reshape(-1)
            np.testing.assert_allclose([float(v.sum()), float((v * v).sum()), float(v[0]), float(v[-1])], g[k], rtol=1e-12, atol=1e-12)
    assert int(g['short_error'][0]) == 1
    path = str(tmp_path / 'short.weights')
    weights_file(path, n // 2)
    with pytest.raises(RuntimeError):
        _mini_model().load_weights(path)
    # round trip: save_weights writes what load_weights reads
    path = str(tmp_path / 'full.weights')
    m = _mini_model(); m.load_weights(path)
    out = str(tmp_path / 'resaved.weights')
    m.save_weights(out)
    assert open(out, 'rb').read() == open(path, 'rb').read()


def test_checkpoint_helpers(tmp_path):
    import torch
    from complex_yolov4_pytorch_amd.utils.train_utils import get_saved_state, save_checkpoint
    m = _mini_model()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 1.0)
    msd, usd = get_saved_state(m, opt, sched, 3, {'lr': 1e-3})
    assert set(usd) == {'epoch', 'configs', 'optimizer', 'lr_scheduler'} and usd['epoch'] == 3
    save_checkpoint(str(tmp_path), 'run', msd, usd, 3)
    back = torch.load(str(tmp_path / 'Model_run_epoch_3.pth'))
    assert list(back) == list(m.state_dict())
    m2 = _mini_model(); m2.load_state_dict(back)
    assert (tmp_path / 'Utils_run_epoch_3.pth').exists()
