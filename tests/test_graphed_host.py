"""Host-side contract of graphed.GraphedTrainStep that needs no GPU: what it refuses, and that warm-up batches (and everything
beyond `max_graphs` shapes) are ordinary eager steps -- each batch stepped exactly once.  The capture / replay itself is covered on
the MI355X by tests/test_gpu_r3.py::test_graphed_train_step_equals_eager_steps."""
import os
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402,F401  (registers the complex_yolov4_pytorch_amd alias)
from complex_yolov4_pytorch_amd import ops  # noqa: E402
from complex_yolov4_pytorch_amd.graphed import GraphedTrainStep  # noqa: E402


class _Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(3))
        self.calls = []

    def forward(self, x, tg):
        self.calls.append((tuple(x.shape), int(tg.shape[0])))
        return (self.w * x.mean()).sum() + 0.0 * tg.sum(), None


class _Opt(torch.optim.SGD):
    capturable = True
    steps = 0

    def step(self, closure=None):
        type(self).steps += 1
        return super().step(closure)

    def refresh_groups(self):      # pragma: no cover (replay path only)
        raise AssertionError('no replay expected on a CPU box')

    note_replayed = refresh_groups


def test_refusals(monkeypatch):
    model = _Model()
    opt = _Opt(model.parameters(), lr=0.1)
    wrapped = torch.nn.Module()
    wrapped.module = model
    with pytest.raises(ops.CyoloError, match='single-process'):
        GraphedTrainStep(wrapped, opt)
    plain = torch.optim.SGD(model.parameters(), lr=0.1)
    with pytest.raises(ops.CyoloError, match='capturable'):
        GraphedTrainStep(model, plain)
    monkeypatch.setenv('CY_WGRAD_SIDE_STREAM', '0')
    with pytest.raises(ops.CyoloError, match='two-stream'):
        GraphedTrainStep(model, opt)


def test_warmup_batches_and_overflow_shapes_are_single_eager_steps(monkeypatch):
    monkeypatch.delenv('CY_WGRAD_SIDE_STREAM', raising=False)
    model = _Model()
    _Opt.steps = 0
    opt = _Opt(model.parameters(), lr=0.1)
    step = GraphedTrainStep(model, opt, warmup=2, max_graphs=0)      # no graph may ever be captured: every call is eager
    x = torch.ones(2, 3, 4, 4)
    w0 = model.w.detach().clone()
    for i, nt in enumerate((5, 5, 7, 5, 5)):
        loss = step(x, torch.zeros(nt, 8))
        assert loss.requires_grad and step.replays == 0 and not step._graphs
        assert _Opt.steps == i + 1 and len(model.calls) == i + 1          # one forward and one optimizer step per batch
    assert model.calls == [((2, 3, 4, 4), 5), ((2, 3, 4, 4), 5), ((2, 3, 4, 4), 7), ((2, 3, 4, 4), 5), ((2, 3, 4, 4), 5)]
    torch.testing.assert_close(model.w.detach(), w0 - 5 * 0.1 * torch.ones(3))
    assert step._seen == {((2, 3, 4, 4), 5, True): 4, ((2, 3, 4, 4), 7, True): 1}


def test_refuses_a_dynamic_loss_scale(monkeypatch):
    """the overflow scan of optim.DynamicLossScale is not part of the captured step -- refuse instead of replaying
    with a flag nobody refreshes (at construction, and when a scaler is attached afterwards)."""
    monkeypatch.delenv('CY_WGRAD_SIDE_STREAM', raising=False)
    model = _Model()
    opt = _Opt(model.parameters(), lr=0.1)
    opt.skip_flag = torch.zeros(1, dtype=torch.int32)
    with pytest.raises(ops.CyoloError, match='DynamicLossScale'):
        GraphedTrainStep(model, opt)
    opt.skip_flag = None
    step = GraphedTrainStep(model, opt, warmup=5)
    step(torch.ones(2, 3, 4, 4), torch.zeros(3, 8))
    opt.skip_flag = torch.zeros(1, dtype=torch.int32)
    with pytest.raises(ops.CyoloError, match='DynamicLossScale'):
        step(torch.ones(2, 3, 4, 4), torch.zeros(3, 8))


def test_wgrad_choice_validation_per_mode():
    """a persisted weight-gradient split is only usable when its slabs fit the layer's slab region (plain and
    64-tile: split <= cap), an atomic split is bounded by the 512-pixel minimum, deterministic engines take plain splits only."""
    from complex_yolov4_pytorch_amd.models.engine import Engine
    e = Engine.__new__(Engine)
    e.det = False
    cap, M = 12, 5776
    assert e._wgrad_choice_ok(12, cap, M) and e._wgrad_choice_ok(1, cap, M)
    assert not e._wgrad_choice_ok(13, cap, M) and not e._wgrad_choice_ok(512, cap, M) and not e._wgrad_choice_ok(0, cap, M)
    assert e._wgrad_choice_ok(1012, cap, M) and not e._wgrad_choice_ok(1013, cap, M) and not e._wgrad_choice_ok(1000, cap, M)
    assert e._wgrad_choice_ok(-12, cap, M) and not e._wgrad_choice_ok(-13, cap, M)         # ceil(5776 / 512) = 12
    e.det = True
    assert e._wgrad_choice_ok(12, cap, M) and not e._wgrad_choice_ok(1004, cap, M) and not e._wgrad_choice_ok(-4, cap, M)
