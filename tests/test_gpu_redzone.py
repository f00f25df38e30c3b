"""Red-zone runs of the whole hot path (VERDICT r4 next #1b): every device buffer an engine owns -- activations, pre-BN tensors,
gradients, packed weights, statistics / sum tables, split-K slabs, head workspaces, the flat parameter gradient -- is allocated
between two 64 KiB bands of 0xFF bytes (models/engine.py::Arena), the step runs at the BASELINE shapes through every kernel
family (forward, dgrad incl. stride 2 and the x_bias gather descriptors, wgrad incl. loader waves, BN passes, pack / fold,
pools, heads, NMS-free eval), and afterwards

* every band must be intact (an out-of-bounds WRITE of any kernel names the buffer it ran out of), and
* loss, outputs and the flat gradient must equal the unguarded run BIT FOR BIT in deterministic mode (0xFF... is a NaN in f16,
  bf16 and f32: an out-of-bounds READ that reaches an accumulator shows up as a differing or non-finite result).

Round 4's driver bench died of a GPU memory-access fault that no test could have caught: nothing ran against guarded buffers.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd.models import engine as engine_mod  # noqa: E402
from complex_yolov4_pytorch_amd.optim import FusedAdam  # noqa: E402
from tests.test_gpu_r2 import DEV, _model  # noqa: E402

GUARD = 64 << 10


def _violations(model):
    bad = []
    for key, e in model._engines.items():
        bad += [(key[:4],) + v for v in e.arena.violations()]
    if getattr(model, '_grad_arena', None) is not None:
        bad += [('model',) + v for v in model._grad_arena.violations()]
    return bad


def _train_steps(cfg, dtype, B, S, det, steps, seed=31, nt=6):
    model = _model(cfg, dtype, deterministic=det)
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    x, tg = syn.bev_images(B, S, seed=seed).to(DEV), syn.targets(B, nt, S, seed=seed, collide=True).to(DEV)
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss, out = model(x, tg)
        loss.backward()
        res = (loss.detach().clone(), out.detach().clone(), model.flat_grad.detach().clone())
        opt.step()
    torch.cuda.synchronize()
    bad = _violations(model)
    nblocks = sum(len(e.arena.blocks) for e in model._engines.values())
    replayed = sum(e.replayed for e in model._engines.values())
    model.release_engines()
    del opt, model
    torch.cuda.empty_cache()
    return res, bad, nblocks, replayed


TRAIN_CASES = [            # cfg, dtype, batch, size: BASELINE configs[1] / [4] / [2]'s per-GPU canvas, the parity mode, the other cfgs
    ('complex_yolov4.cfg', 'f16', 16, 608), ('complex_yolov4.cfg', 'bf16', 16, 608), ('complex_yolov4.cfg', 'f32', 4, 608),
    ('complex_yolov4.cfg', 'f16', 8, 1024), ('complex_yolov4.cfg', 'f16', 2, 1216), ('complex_yolov4.cfg', 'f16', 3, 416),
    ('complex_yolov4_tiny.cfg', 'f16', 2, 608), ('complex_yolov3.cfg', 'f16', 2, 608), ('complex_yolov3_tiny.cfg', 'f16', 2, 416)]


@pytest.mark.parametrize('cfg,dtype,B,S', TRAIN_CASES)
def test_train_step_between_red_zones_deterministic(monkeypatch, cfg, dtype, B, S):
    """Three deterministic train steps (eager + tuned, recorded, REPLAYED launch list) without and with red zones."""
    plain, _, _, _ = _train_steps(cfg, dtype, B, S, True, 3)
    monkeypatch.setattr(engine_mod.Engine, 'GUARD_BYTES', GUARD)
    guarded, bad, nblocks, replayed = _train_steps(cfg, dtype, B, S, True, 3)
    print('%s %s B%d %dx%d: %d guarded buffers, %d replayed passes, loss %.4f' % (cfg, dtype, B, S, S, nblocks, replayed, float(guarded[0])))
    assert nblocks > 20 and bad == [], bad
    for name, a, b in zip(('loss', 'outputs', 'flat gradient'), plain, guarded):
        assert torch.isfinite(b).all(), name
        assert torch.equal(a, b), '%s differs between the plain and the red-zoned run (an out-of-bounds read?)' % name


@pytest.mark.parametrize('dtype', ['f16', 'bf16'])
def test_benchmarked_default_mode_between_red_zones(monkeypatch, dtype):
    """The mode bench.py times (atomics, weight gradients on the side stream, per-layer tuned kernels, replayed launch lists):
    not bit-reproducible, so bands + finiteness + the loss against the deterministic mode's."""
    det, _, _, _ = _train_steps('complex_yolov4.cfg', dtype, 16, 608, True, 1)
    monkeypatch.setattr(engine_mod.Engine, 'GUARD_BYTES', GUARD)
    for steps in (1, 4):
        got, bad, nblocks, replayed = _train_steps('complex_yolov4.cfg', dtype, 16, 608, False, steps)
        assert bad == [], bad
        assert all(bool(torch.isfinite(t).all()) for t in got)
        if steps == 1:
            # (bf16: two runs of the SAME default-mode engine differ by up to 5 % in the loss on this random-init net -- 395.5 vs 415.9
            # in test_gpu_r6's concatenation test --, the fp32 atomics ordering the BatchNorm statistics differently every run and 8
            # mantissa bits amplifying it through 107 layers; f16 stays within 0.5 %.  A red zone that leaked would give NaN.)
            assert abs(float(got[0]) - float(det[0])) <= (8e-2 if dtype == 'bf16' else 2e-2) * abs(float(det[0]))
        else:
            assert replayed >= 2


@pytest.mark.parametrize('dtype,B,S', [('f16', 32, 608), ('f32', 4, 608), ('f16', 4, 1024)])
def test_eval_forward_between_red_zones(monkeypatch, golden, dtype, B, S):
    """BASELINE configs[3]'s forward (fused conv + BN + activation eval kernels, static_eval_weights, replayed) between red zones.
    BatchNorm running statistics as calibrated by the reference for the configs[3] golden (with the initial 0 / 1 statistics the
    random-init net overflows f16 after ~100 layers whatever the buffers look like)."""
    from tests.test_gpu_r4 import _eval_model
    g = golden('darknet_eval')

    def run():
        model = _eval_model(g, dtype)
        model.static_eval_weights = dtype != 'f32'
        x = syn.bev_images(B, S, seed=33).to(DEV)
        with torch.no_grad():
            outs = [model(x) for _ in range(3)]
        torch.cuda.synchronize()
        bad = _violations(model)
        model.release_engines()
        return outs, bad
    plain, _ = run()
    monkeypatch.setattr(engine_mod.Engine, 'GUARD_BYTES', GUARD)
    guarded, bad = run()
    assert bad == [], bad
    for a, b in zip(plain, guarded):
        assert torch.isfinite(b).all() and torch.equal(a, b)
    assert torch.equal(guarded[0], guarded[2])


def test_red_zones_do_catch_an_overrun(monkeypatch):
    """The harness itself: a kernel made to write one row past its view is reported with the buffer's name."""
    from complex_yolov4_pytorch_amd import ops
    monkeypatch.setattr(engine_mod.Engine, 'GUARD_BYTES', GUARD)
    arena = engine_mod.Arena(DEV, GUARD)
    a = arena.new('a', 64 * 8 * 8 * 32, torch.float16)
    b = arena.new('b', 64 * 8 * 8 * 32, torch.float16, zero=True)
    assert arena.violations() == []
    src = ops.View(a, 0, 64, 8, 8, 32, 32, ops.CY_F16)
    a.fill_(1.0)
    ops.slice_copy(ops.View(a, 0, 64, 8, 8, 32, 32, ops.CY_F16), ops.View(b, 0, 64, 8, 8, 32, 32, ops.CY_F16))
    assert arena.violations() == []
    over = ops.View(b, 32, 64, 8, 8, 32, 32, ops.CY_F16)      # starts one pixel row (64 bytes) in: the last row overruns by 64 bytes
    ops.slice_copy(src, over)
    torch.cuda.synchronize()
    bad = arena.violations()
    assert bad and bad[0][0] == 'b' and bad[0][1] == 'above' and bad[0][2] == 1, bad
