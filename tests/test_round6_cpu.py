"""Round 6 host logic on CPU (operator layer = tests/opsim.py): the byte-budgeted engine cache of Darknet (the reference changes
img_size every 10 batches and doubles it under mosaic, kitti_dataset.py:42-43,144,225-230) and the bounded list of superseded
engine tables."""
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
