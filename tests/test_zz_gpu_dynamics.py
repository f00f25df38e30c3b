"""Training-DYNAMICS comparisons of the benchmarked 16-bit modes with the fp32 parity mode.

This file collects LAST among the GPU tests on purpose (`test_zz_...`): these tests compare trajectories / single steps of a
chaotic system, their bounds are statistical statements (each derivation is written ONCE, next to the bound, as control x
margin), and `pytest -x` must never again stop at one of them before the kernel, replay and configs[3] parity tests have run
(GPUTEST_r04: 38 tests hidden behind one draw of a two-sided band).

* the BENCHMARKED 16-bit modes against the fp32 parity mode on a CONDITIONED complex_yolov4.cfg (50 / 100 Adam steps from the
  seeded init) where element-wise agreement is possible: flat-gradient cosine, loss, decoded probabilities;
* the control: the reference's float32 arithmetic with IDEAL 16-bit storage (oracle storage_round) deviates the same way;
* convergence A/B: 100 steps f16 against f32 over three batch sets, one-sided.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from tests.test_gpu_r2 import DEV, _model  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
B, S = 16, 608
N_STEPS, SNAP_AT = 100, 50


def _batches(n=4, seed0=70):
    return [(syn.bev_images(B, S, seed=seed0 + i).to(DEV), syn.targets(B, 6, S, seed=seed0 + i).to(DEV)) for i in range(n)]


def _train(dtype, steps, snap_at=(), deterministic=True, seed0=70):
    """`steps` FusedAdam steps (lr 1e-3, the reference's default, train_config.py:82-94) of complex_yolov4.cfg at 608x608 batch
    16 from the seeded init over four fixed batches.  -> (losses, {n: state-dict snapshot after n steps for n in snap_at})."""
    from complex_yolov4_pytorch_amd.optim import FusedAdam
    model = _model('complex_yolov4.cfg', dtype, deterministic=deterministic)
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    data = _batches(seed0=seed0)
    losses, snaps = [], {}
    for i in range(steps):
        x, tg = data[i % len(data)]
        opt.zero_grad(set_to_none=True)
        loss, _ = model(x, tg)
        loss.backward()
        opt.step()
        losses.append(loss.detach().reshape(-1)[0])
        if i + 1 in snap_at:
            snaps[i + 1] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    losses = [float(v) for v in torch.stack(losses).cpu()]
    model.release_engines()
    del opt, model
    torch.cuda.empty_cache()
    return losses, snaps


@pytest.fixture(scope='module')
def f32_run():
    return _train('f32', N_STEPS, snap_at=(SNAP_AT, N_STEPS))


def _one_step(dtype, snap, deterministic, batch, loss_scale=None):
    model = _model('complex_yolov4.cfg', dtype, deterministic=deterministic, loss_scale=loss_scale)
    model.load_state_dict(snap)
    model.train()
    x, tg = batch
    loss, out = model(x, tg)
    loss.backward()
    res = (float(loss.detach().reshape(-1)[0]), out.detach().clone(), model.flat_grad.detach().double().clone())
    model.release_engines()
    del model
    torch.cuda.empty_cache()
    return res


def _agreement(a, b):
    """(flat-gradient cosine, gradient norm ratio, loss rel, probabilities max |d|, median |d|) of step result a against b."""
    (la, oa, ga), (lb, ob, gb) = a, b
    dp = (oa[..., 6:] - ob[..., 6:]).abs()
    return (float((ga * gb).sum() / (ga.norm() * gb.norm())), float(ga.norm() / gb.norm()), abs(la - lb) / abs(lb), float(dp.max()),
            float(dp.median()))


# What the 16-bit modes hold against the fp32 parity mode on the conditioned net, ONE step on a batch of the conditioning run.
# VERDICT r3 next #1a asked cosine >= 0.99 (f16) / 0.97 (bf16) and "if even a conditioned net does not agree, that is a finding".
# It is a finding about the FORMAT, settled by controls (each bound below is stated ONCE as control x margin and is not re-sized
# after a failure -- VERDICT r4 next #2c):
#   * f16: the control is the mode's own run-to-run agreement, measured in the same test -- a REPEAT of the f16 step against the
#     first (atomics' order moves one f16 rounding somewhere and the net amplifies it): 0.972-0.983 over five sessions, while f16 vs
#     f32 was 0.958-0.973.  Asserted: cos(f16, f32) >= cos(f16 repeat, f16) - 0.04, i.e. twice the largest gap seen (0.021).
#   * bf16: its control needs the CPU oracle and lives in test_16bit_step_matches_ideal_16bit_storage (device cosine >= the
#     reference's own arithmetic with ideal bf16 storage - 0.25); here only a floor far below every observation (0.71-0.80).
#   * loss / probabilities: the largest values of five sessions x 2 (f16: loss rel 9e-4, probabilities 3.9e-2; bf16: 1.4e-2, 0.24),
#     rounded up.
#   * the fp32 default mode (atomics) against the fp32 deterministic mode gives 0.99991-0.99998: the snapshot is well conditioned
#     for float32, so the angles above are the 16-bit formats', not the test's.
COND = {'f16': dict(cos=0.90, loss=2e-3, prob=8e-2), 'bf16': dict(cos=0.55, loss=3e-2, prob=0.5)}
F16_VS_REPEAT_MARGIN = 0.04


def test_conditioned_net_16bit_step_agrees_with_fp32(f32_run):
    """At the seeded random init complex_yolov4.cfg amplifies a 1e-7 perturbation to 4e-2 in the gradients (the ORACLE's own
    float32 run differs from its float64 run by that much: profiles/r04_oracle_f64_vs_f32.txt), so the 16-bit modes could only
    be bounded by norms there (tests/test_gpu_r3.py).  After 50 / 100 Adam steps in the fp32 parity mode the net is
    conditioned on its four batches; ONE step from those snapshots in f32 (deterministic parity mode), f16 and bf16 (the
    benchmarked default mode) is compared element-wise: flat-gradient cosine, loss, decoded probabilities -- on a batch of the
    conditioning run (asserted) and on an unseen batch (printed).  Two more columns say what the comparison can resolve: the
    fp32 default mode (atomics) against the fp32 deterministic mode, and a REPEAT of the f16 step against the first f16 step."""
    losses, snaps = f32_run
    assert losses[SNAP_AT - 1] < 0.5 * losses[0], losses[:SNAP_AT:7]
    seen, unseen = _batches(1, seed0=70)[0], _batches(1, seed0=80)[0]
    table = {}
    for n in (SNAP_AT, N_STEPS):
        for bname, batch in (('seen', seen), ('unseen', unseen)):
            ref = _one_step('f32', snaps[n], True, batch)
            row = {'f16': _agreement(_one_step('f16', snaps[n], False, batch), ref),
                   'bf16': _agreement(_one_step('bf16', snaps[n], False, batch), ref)}
            if bname == 'seen':
                row['f16 loss_scale 256'] = _agreement(_one_step('f16', snaps[n], False, batch, loss_scale=256.0), ref)
                row['f32 default'] = _agreement(_one_step('f32', snaps[n], False, batch), ref)
                f16a = _one_step('f16', snaps[n], False, batch)
                row['f16 repeat vs f16'] = _agreement(_one_step('f16', snaps[n], False, batch), f16a)
            table[(n, bname)] = (ref[0], row)
    for (n, bname), (l32, row) in table.items():
        for mode, (cos, nr, rel, pmax, pmed) in row.items():
            print('conditioned v4 (f32, %3d Adam steps: loss %.1f -> %.1f), %-6s batch (f32 loss %8.3f): %-17s vs f32 det: gradient cosine '
                  '%.5f, norm ratio %.4f, loss rel %.2e, probabilities |d| max %.2e median %.2e'
                  % (n, losses[0], losses[n - 1], bname, l32, mode, cos, nr, rel, pmax, pmed))
    for dtype, b in COND.items():
        cos, _, rel, pmax, _ = table[(SNAP_AT, 'seen')][1][dtype]
        assert cos >= b['cos'], (dtype, cos)
        assert rel <= b['loss'], (dtype, rel)
        assert pmax <= b['prob'], (dtype, pmax)
    for n in (SNAP_AT, N_STEPS):       # f16 against its own run-to-run agreement (the control), both snapshots
        row = table[(n, 'seen')][1]
        assert row['f16'][0] >= row['f16 repeat vs f16'][0] - F16_VS_REPEAT_MARGIN, (n, row['f16'][0], row['f16 repeat vs f16'][0])
        assert row['f32 default'][0] >= 0.999, row['f32 default']         # (the float32 control: 0.99991-0.99998 observed)


@pytest.mark.parametrize('dtype,tdt', [('f16', torch.float16), ('bf16', torch.bfloat16)])
def test_16bit_step_matches_ideal_16bit_storage(f32_run, dtype, tdt):
    """Is the 16-bit step's distance from float32 the kernels' doing, or what 16-bit STORAGE does to this function?  Same
    conditioned snapshot (50 steps), batch of 4 (what the CPU oracle finishes in seconds).  Device: f16 / bf16 default mode
    against the fp32 parity mode.  Oracle: the reference's float32 arithmetic with every conv input, weight copy and pre-BN
    output -- and the gradients flowing through them -- rounded to the 16-bit type, against the same arithmetic without
    rounding.  The device's gradient cosine must be no worse than the ideal-storage one minus a margin: 0.06 for f16 (device 0.912 /
    0.935 / 0.946 vs ideal 0.928 / 0.943 / 0.951 over three sessions: gaps <= 0.016); 0.25 for bf16, where BOTH cosines move by
    +-0.1 with the last bits of the snapshot and the atomics' order (ideal storage 0.50 / 0.58 / 0.68 / 0.69, device 0.57 / 0.60 /
    0.62 / 0.65 / 0.65: the worst pairing of those is -0.12).  Probability error medians within a factor of two, maxima within three."""
    from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
    from oracle import darknet_ref
    from tests.util import storage_round
    _, snaps = f32_run
    snap = snaps[SNAP_AT]
    x, tg = syn.bev_images(4, S, seed=70), syn.targets(4, 6, S, seed=70)
    batch = (x.to(DEV), tg.to(DEV))
    dev = _agreement(_one_step(dtype, snap, False, batch), _one_step('f32', snap, True, batch))
    net = darknet_ref.DarknetRef(parse_cfg(os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')))
    ps, bs = net.param_shapes()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))

    def oracle(rnd):
        params = {k: snap[k].detach().float().cpu().clone().requires_grad_(True) for k in ps}
        bufs = {k: snap[k].detach().float().cpu().clone() for k in bs}
        out, loss, _ = net.forward(params, x, tg, True, True, bufs, storage_round=rnd)
        loss.sum().backward()
        return float(loss.detach().sum()), out.detach(), torch.cat([params[k].grad.reshape(-1).double() for k in ps])
    ideal = _agreement(oracle(storage_round(tdt)), oracle(None))
    print('conditioned v4 (%d steps), batch 4: %s device vs fp32 parity mode: gradient cosine %.5f, loss rel %.2e, probabilities |d| max %.2e '
          'median %.2e;  ORACLE float32 arithmetic with ideal %s storage vs without: cosine %.5f, loss rel %.2e, probabilities max %.2e median %.2e'
          % (SNAP_AT, dtype, dev[0], dev[2], dev[3], dev[4], dtype, ideal[0], ideal[2], ideal[3], ideal[4]))
    assert dev[0] >= ideal[0] - (0.06 if dtype == 'f16' else 0.25), (dev[0], ideal[0])
    assert dev[4] <= 2.0 * ideal[4] + 1e-4 and dev[3] <= 3.0 * ideal[3] + 1e-3      # (medians within 2 x; the maxima, noisier, within 3 x)


CONV_SEEDS = (70, 170, 270)       # batch sets of the convergence A/B (70 = the conditioning run's, shared through f32_run)


def test_f16_converges_like_fp32(f32_run):
    """Convergence A/B of the benchmarked f16 default mode against the fp32 parity mode: 100 Adam steps from the same seeded init
    over the same four batches, for THREE batch sets.  A 100-step trajectory of this net is chaotic -- the f16 mode is not even
    bit-reproducible against itself (atomics' order), and round 4's two-sided +-12 % band on ONE draw failed on the driver's box
    at ratio 0.799 after 0.955 / 1.043 / 1.044 in three sessions -- so a single final-loss ratio is not a parity statement.
    What a correct half-precision implementation cannot fail (VERDICT r4 next #2a): every run of both modes is finite and
    reaches < 0.1 x its initial loss (observed: <= 0.04), and f16 does not converge WORSE: mean final loss (last four steps = one pass
    over the batches, mean over the batch sets) <= 1.25 x the fp32 mean -- one-sided.  Derivation of the 1.25: the ten per-set ratios
    seen so far (0.80, 0.80, 0.91, 0.94, 0.96, 0.98, 1.03, 1.04, 1.04, 1.21) have mean 0.97 and standard deviation 0.12, a mean over
    three sets 0.07: 1.25 is four of those above the mean.  A half-precision path that is actually broken stalls or diverges at
    many times the fp32 loss.  The per-set ratios are printed."""
    runs32 = {CONV_SEEDS[0]: f32_run[0]}
    for sd in CONV_SEEDS[1:]:
        runs32[sd] = _train('f32', N_STEPS, seed0=sd)[0]
    runs16 = {sd: _train('f16', N_STEPS, deterministic=False, seed0=sd)[0] for sd in CONV_SEEDS}
    fin32 = {sd: float(np.mean(runs32[sd][-4:])) for sd in CONV_SEEDS}
    fin16 = {sd: float(np.mean(runs16[sd][-4:])) for sd in CONV_SEEDS}
    for sd in CONV_SEEDS:
        l32, l16 = runs32[sd], runs16[sd]
        print('v4 608x608 B16, %d Adam steps on 4 batches (seeds %d..%d): loss f32 %.2f -> %.3f, f16 %.2f -> %.3f (ratio %.3f); at step 25: '
              '%.2f / %.2f, step 50: %.2f / %.2f' % (N_STEPS, sd, sd + 3, l32[0], fin32[sd], l16[0], fin16[sd], fin16[sd] / fin32[sd],
                                                     l32[24], l16[24], l32[49], l16[49]))
        assert all(np.isfinite(l16)) and all(np.isfinite(l32))
        assert fin32[sd] < 0.1 * l32[0] and fin16[sd] < 0.1 * l16[0], (sd, fin32[sd], fin16[sd])
    m32, m16 = float(np.mean(list(fin32.values()))), float(np.mean(list(fin16.values())))
    print('mean final loss over %d batch sets: f32 %.3f, f16 %.3f (ratio %.3f; bound: <= 1.25, one-sided)' % (len(CONV_SEEDS), m32, m16, m16 / m32))
    assert m16 <= 1.25 * m32


def test_f16_inference_detections_match_fp32_on_the_conditioned_net(f32_run):
    """End to end, f16 against float32, where the comparison means something (VERDICT r4 next #2d): on the random-init net of the
    configs[3] golden even IDEAL f16 storage keeps 2 of 140 float32 detections, so tests/test_gpu_r4.py can only compare error
    statistics there.  The conditioned net (100 Adam steps in the fp32 parity mode, running statistics included) has confident
    boxes: model.eval() + post_processing_v2 (rotated merge-NMS on the device) in the f16 fused-eval path with static_eval_weights
    -- the benchmarked inference path -- against the f32 parity eval path (itself pinned to the reference's eval forward by
    test_inference_b32_608_against_reference[f32]), batch 16 at 608 x 608, a batch of the conditioning run.  Bounds: the f16 path
    finds >= 90 % of the f32 detections (same class, box within 5 % relative) and adds <= 10 % (+ 2) of its own; sized on the
    probabilities' max |d| of 3e-2 that the conditioned-net step test measures for f16 (a detection within 3e-2 of the 0.5
    threshold may flip: a few per cent of ~100)."""
    from complex_yolov4_pytorch_amd.utils.evaluation_utils import post_processing_v2
    from tests.test_gpu_r4 import _match
    _, snaps = f32_run
    x = _batches(1, seed0=70)[0][0]
    dets = {}
    for dtype in ('f32', 'f16'):
        model = _model('complex_yolov4.cfg', dtype)
        model.load_state_dict(snaps[N_STEPS])
        model.eval()
        model.cpu_outputs = False
        model.static_eval_weights = dtype != 'f32'
        with torch.no_grad():
            out = model(x)
            out = model(x)               # (second batch: the cached weight pack, as benchmarked)
        d = post_processing_v2(out, conf_thresh=0.5, nms_thresh=0.5)
        dets[dtype] = [None if r is None else r.numpy() for r in d]
        model.release_engines()
        del model
        torch.cuda.empty_cache()
    total = sum(0 if r is None else len(r) for r in dets['f32'])
    found = sum(_match(d, r, 0.05) for d, r in zip(dets['f16'], dets['f32']) if r is not None)
    extra = sum(0 if d is None else len(d) for d in dets['f16']) - found
    print('conditioned v4 (%d steps), eval B16 608 + post_processing_v2(0.5, 0.5): f32 path %d detections; f16 fused-eval path finds %d of them, '
          '%d unmatched of its own' % (N_STEPS, total, found, extra))
    assert total >= 16, 'the conditioned net should be confident about most of its 96 training boxes'
    assert found >= 0.9 * total and extra <= 0.1 * total + 2
