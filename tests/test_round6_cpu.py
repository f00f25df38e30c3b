"""Round 6 host logic on CPU (operator layer = tests/opsim.py): the byte-budgeted engine cache of Darknet (the reference changes
img_size every 10 batches and doubles it under mosaic, kitti_dataset.py:42-43,144,225-230) and the bounded list of superseded
engine tables."""
import pytest
import torch

import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
from complex_yolov4_pytorch_amd.models.engine import Engine
from tests import opsim
from tests.util import mini_cfg_path


def _mini(monkeypatch):
    opsim.install(monkeypatch)
    torch.manual_seed(3)
    return Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f32')


def test_engine_cache_evicts_least_recently_used_within_its_budget(monkeypatch):
    """Multiscale training in one process: engines of geometries not used recently are dropped when the cache exceeds its byte
    budget, the newest engine always stays, a revisited geometry is rebuilt and gives the same loss as before."""
    m = _mini(monkeypatch)
    monkeypatch.setattr(Engine, 'nbytes', lambda self: 1000 * self.plan.H)       # (the simulator's tensors live on the host)
    monkeypatch.setattr(Darknet, '_engine_budget', lambda self, device: self.engine_budget_bytes)
    m.engine_budget_bytes = 1000 * (64 + 96) + 1
    m.train()
    losses = {}
    for size in (64, 96, 128, 64, 96):
        x, tg = syn.bev_images(2, size, seed=4, sparsity=0.5), syn.targets(2, 3, size, seed=4)
        loss, _ = m(x, tg)
        loss.backward()
        if size in losses:
            assert float(loss) == losses[size]          # the rebuilt engine computes what the evicted one did
        losses.setdefault(size, float(loss))
        sizes = [k[1] for k in m._engines]
        assert sizes[-1] == size and m.engine_bytes() <= m.engine_budget_bytes or len(sizes) == 1, sizes
    # 64, 96 fit together; 128 evicts both (128 alone is over the budget but the newest engine always stays); 64 then evicts 128; 96 joins 64
    assert [k[1] for k in m._engines] == [64, 96] and m.engine_evictions == 3
    # a recently used engine is kept in preference to an older one
    m(syn.bev_images(2, 64, seed=4, sparsity=0.5), syn.targets(2, 3, 64, seed=4))
    m.engine_budget_bytes = 1000 * (64 + 32) + 1
    m(syn.bev_images(2, 32, seed=4, sparsity=0.5), syn.targets(2, 3, 32, seed=4))
    assert [k[1] for k in m._engines] == [64, 32]
    # an engine a captured hipGraph points into is never evicted
    m._engines[next(iter(m._engines))].pin_retired = True
    m.engine_budget_bytes = 1
    m(syn.bev_images(2, 96, seed=4, sparsity=0.5), syn.targets(2, 3, 96, seed=4))
    assert [k[1] for k in m._engines] == [64, 96]


def test_superseded_engine_tables_are_bounded(monkeypatch):
    """parameters that move every epoch (EMA swap, load_state_dict(assign=True)) used to leave one pack table per move
    in Engine._retired_ws for the engine's lifetime; now the list is bounded unless a hipGraph capture pinned it."""
    m = _mini(monkeypatch)
    m.train()
    x, tg = syn.bev_images(2, 64, seed=4, sparsity=0.5), syn.targets(2, 3, 64, seed=4)
    m(x, tg)[0].backward()
    eng = next(iter(m._engines.values()))
    for _ in range(3 * Engine.MAX_RETIRED):
        with torch.no_grad():
            for p in m.parameters():
                p.data = p.data.clone()          # every parameter at a new address
        m(x, tg)[0].backward()
    assert len(eng._retired_ws) <= Engine.MAX_RETIRED
    eng.pin_retired = True
    n0 = len(eng._retired_ws)
    for _ in range(3):
        with torch.no_grad():
            for p in m.parameters():
                p.data = p.data.clone()
        m(x, tg)[0].backward()
    assert len(eng._retired_ws) > n0


def test_sibling_convs_fused_equal_the_unfused_plan(monkeypatch):
    """The two 1x1 convs over one input of every CSP stage (complex_yolov4.cfg:44-64 and the same pattern in the four later
    stages) as ONE forward conv (joint pre-BN buffer and statistics table), ONE weight gradient and ONE input gradient: the
    engine with CY_SIBLING_FUSE=2 against the same engine with the pairs switched off, on the CPU operator simulator (fp32):
    five pairs found, same loss, outputs, running statistics and gradients up to float32 summation order."""
    import os
    cfg = os.path.join(os.path.dirname(__file__), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
    opsim.install(monkeypatch)
    x, tg = syn.bev_images(2, 96, seed=7, sparsity=0.5), syn.targets(2, 4, 96, seed=7)
    res = {}
    for mode in ('0', '2'):
        monkeypatch.setenv('CY_SIBLING_FUSE', mode)
        torch.manual_seed(0)
        m = Darknet(cfg, use_giou_loss=True, dtype='f32')
        sd = m.state_dict()
        sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
        m.load_state_dict(sd)
        m.train()
        loss, out = m(x, tg)
        loss.backward()
        eng = next(iter(m._engines.values()))
        res[mode] = (float(loss), out.clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                     {k: v.clone() for k, v in m.state_dict().items() if 'running' in k or 'tracked' in k}, len(eng._sib))
    assert res['0'][4] == 0 and res['2'][4] == 5
    assert abs(res['0'][0] - res['2'][0]) <= 1e-5 * abs(res['0'][0])
    torch.testing.assert_close(res['2'][1], res['0'][1], rtol=1e-4, atol=1e-5)
    for k, v in res['0'][3].items():
        torch.testing.assert_close(res['2'][3][k], v, rtol=1e-5, atol=1e-6)
    worst = 0.0
    for k, g0 in res['0'][2].items():
        g2 = res['2'][2][k]
        worst = max(worst, float((g2 - g0).norm() / (g0.norm() + 1e-12)))
    assert worst < 1e-3, worst


@pytest.mark.parametrize('sib', ['0', '2'])
def test_concat_producers_sums_in_the_closing_dgrad_equal_the_reduce_passes(monkeypatch, sib):
    """The closing 1x1 conv of every CSP stage reads [branch | A] (route layers=-1,-7, complex_yolov4.cfg): its input gradient is the
    last writer of both producers' output gradients, so its epilogue takes BOTH layers' BatchNorm-backward sums (graph.py
    'dx_sums_cat'; engine: one pre-BN buffer [branch | A (| B)], one [4][C1 + C2] vector block, a sums table of its own that the two
    BatchNorm backward passes read through a column window).  CY_CAT_SUMS=0 (two reduce passes) against the default on the CPU
    operator simulator, with and without the sibling fusion (3-wide / 2-wide pre-BN buffers): five concatenations taken, ten
    layers without a reduce pass, the same loss, outputs, running statistics and gradients."""
    import os
    cfg = os.path.join(os.path.dirname(__file__), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
    opsim.install(monkeypatch)
    monkeypatch.setenv('CY_SIBLING_FUSE', sib)
    monkeypatch.setenv('CY_DGRAD_BN_SUMS', '2')      # every marked launch takes its sums, untimed (nothing to time on the CPU)
    x, tg = syn.bev_images(2, 96, seed=11, sparsity=0.5), syn.targets(2, 4, 96, seed=11)
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('CY_CAT_SUMS', mode)
        torch.manual_seed(0)
        m = Darknet(cfg, use_giou_loss=True, dtype='f32')
        sd = m.state_dict()
        sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
        m.load_state_dict(sd)
        m.train()
        for _ in range(2):          # (twice: the tables of their own must be zero again at the second backward)
            m.zero_grad()
            loss, out = m(x, tg)
            loss.backward()
        eng = next(iter(m._engines.values()))
        res[mode] = (float(loss), out.clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                     {k: v.clone() for k, v in m.state_dict().items() if 'running' in k or 'tracked' in k},
                     len(eng._cat), len(eng._cat_on), {L: c0 for L, (ct, c0) in eng._cat_on.items()})
    assert res['0'][4] == 0 and res['1'][4] == 5 and res['1'][5] == 10
    assert sorted(res['1'][6].values()) == [0] * 5 + [64, 64, 128, 256, 512]
    assert abs(res['0'][0] - res['1'][0]) <= 1e-6 * abs(res['0'][0])
    torch.testing.assert_close(res['1'][1], res['0'][1], rtol=1e-5, atol=1e-6)
    for k, v in res['0'][3].items():
        torch.testing.assert_close(res['1'][3][k], v, rtol=1e-6, atol=1e-7)
    worst = 0.0
    for k, g0 in res['0'][2].items():
        worst = max(worst, float((res['1'][2][k] - g0).norm() / (g0.norm() + 1e-12)))
    assert worst < 1e-4, worst
