"""End-to-end GPU parity: the drop-in Darknet on the HIP path vs the golden fixtures produced by the reference
(tests/golden/darknet.npz) and vs the oracle on the same seeded batch.

f32 parity mode is held to the north_star tolerances where the problem is well conditioned (loss 1e-4 relative,
probabilities/logits 1e-3 .. 2e-3 absolute after 21/110 float32 layers); f16 performance mode reports its own,
looser band explicitly (after 110 half-precision conv layers head logits cannot meet 1e-3)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet  # noqa: E402
from tests.golden.make_golden import METRIC_KEYS  # noqa: E402

CFG = os.path.join(os.path.dirname(__file__), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg')
DEV = 'cuda'


def _model(cfg, giou, dtype, **kw):
    torch.manual_seed(0)
    m = Darknet(os.path.join(CFG, cfg), use_giou_loss=giou, dtype=dtype, **kw)
    sd = m.state_dict()
    sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
    m.load_state_dict(sd)
    return m.to(DEV)


CASES = [('tiny', 'complex_yolov4_tiny.cfg', 2, 608), ('v4', 'complex_yolov4.cfg', 1, 416)]


@pytest.mark.parametrize('tag,cfg,B,S', CASES)
@pytest.mark.parametrize('mode', ['giou', 'mse'])
def test_train_step_f32_parity(golden, tag, cfg, B, S, mode):
    g = golden('darknet')
    # v4 at 416 / batch 1 is ill-conditioned enough (BN over 169 samples at stride 32) that the summation-order noise of
    # the default mode's fp32 atomics occasionally (about 1 run in 10) pushes one gradient norm past the band below: the
    # parity check of that case runs in the deterministic mode (fixed-order folds, bit-identical from run to run)
    model = _model(cfg, mode == 'giou', 'f32', deterministic=(tag == 'v4'))
    model.train()
    x, tg = syn.bev_images(B, S, seed=1).to(DEV), syn.targets(B, 6, S, seed=1).to(DEV)
    loss, out = model(x, tg)
    loss.sum().backward()
    key = '%s_%s_' % (tag, mode)
    assert out.is_cuda and list(out.shape) == list(g[key + 'out_shape'])
    assert loss.dim() == (1 if mode == 'giou' else 0)
    np.testing.assert_allclose(loss.detach().cpu().numpy().reshape(-1), g[key + 'loss'], rtol=1e-4)
    np.testing.assert_allclose(out[:, ::97].cpu().numpy(), g[key + 'out_rows'], rtol=2e-3, atol=2e-3)
    met = [[yl.metrics[k] for k in METRIC_KEYS] for yl in model.yolo_layers]
    np.testing.assert_allclose(met, g[key + 'metrics'], rtol=2e-3, atol=1e-5)
    assert all(yl.metrics_raw[19] == 0 for yl in model.yolo_layers)
    gn = np.asarray([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    np.testing.assert_allclose(gn, g[key + 'grad_norm'], rtol=5e-3 if tag == 'tiny' else 3e-2, atol=1e-6)
    gh = np.stack([p.grad.reshape(-1)[:8].cpu().numpy() for _, p in model.named_parameters()])
    ref_gh = g[key + 'grad_head']
    # v4 at random init / batch 1 is ill-conditioned, and the all-leaky tiny net is exposed to kink flips (a pre-activation
    # within round-off of 0; see tests/test_plan_sim.py and DESIGN.md section 4): most tensors tight, none wrong.
    # The tight element-wise gradient check is test_mini_cfg_all_block_types[f32].
    rel = (np.abs(gh - ref_gh) / (np.abs(ref_gh).max(1, keepdims=True) + 1e-12)).max(1)
    print('%s %s grad-head rel err: median %.2e max %.2e' % (tag, mode, float(np.median(rel)), float(rel.max())))
    assert np.median(rel) < (5e-3 if tag == 'tiny' else 0.1) and rel.max() < (0.05 if tag == 'tiny' else 0.5)
    sd = model.state_dict()
    bn = np.stack([sd[str(n)][:8].cpu().numpy() for n in g[key + 'bn_names']])
    # running_var comes from fp32 partial sums (E[x^2]-m^2 folded in double): 5e-4
    np.testing.assert_allclose(bn, g[key + 'bn_head'], rtol=5e-4, atol=1e-5)
    if mode == 'giou':
        model.eval()
        with torch.no_grad():
            o = model(x)
        assert not o.is_cuda                                   # reference returns a CPU tensor (darknet2pytorch.py:228)
        np.testing.assert_allclose(o[:, ::97].numpy(), g['%s_eval_rows' % tag], rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize('tag,cfg,B,S', CASES)
def test_train_step_f16_band(golden, monkeypatch, tag, cfg, B, S):
    """Performance mode (fp16 storage, fp32 accumulate): stated band, not the fp32 tolerance.  The kernels are the library's
    shape-only defaults (no per-layer timing): these small shapes are not in the persisted tune table, a timed choice differs from
    run to run, and on the batch-1 random-init net a different summation order moves the loss by ~1 % (ten runs on the MI355X,
    round 6: 0.1-1.2 %) -- the band is about fp16 storage, not about which tile the tuner happened to crown."""
    monkeypatch.setenv('CY_CONV_AUTOTUNE', '0')
    monkeypatch.setenv('CY_WGRAD_AUTOTUNE', '0')
    g = golden('darknet')
    model = _model(cfg, True, 'f16')
    model.train()
    x, tg = syn.bev_images(B, S, seed=1).to(DEV), syn.targets(B, 6, S, seed=1).to(DEV)
    loss, out = model(x, tg)
    loss.sum().backward()
    key = '%s_giou_' % tag
    ref_loss = float(g[key + 'loss'][0])
    rel = abs(float(loss.detach()) - ref_loss) / ref_loss
    print('f16 loss %.5f vs fp32 reference %.5f (rel %.2e)' % (float(loss), ref_loss, rel))
    assert rel < 3e-2
    got, ref = out[:, ::97].cpu().numpy(), g[key + 'out_rows']
    dprob = np.abs(got[..., 6:] - ref[..., 6:])
    print('f16 probabilities: median |d| %.3e, max |d| %.3e' % (float(np.median(dprob)), float(dprob.max())))
    if tag == 'tiny':
        # 21 layers: probabilities within 3e-2, boxes within 5% / 0.5 px
        assert dprob.max() < 3e-2
        np.testing.assert_allclose(got[..., :4], ref[..., :4], rtol=5e-2, atol=0.5)
    else:
        # 110 fp16 layers on a random-init net with batch-1 BatchNorm (ill-conditioned, see test_plan_sim.py): stated band
        assert np.median(dprob) < 3e-2 and dprob.max() < 0.35
    gn = np.asarray([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    assert np.all(np.isfinite(gn))
    ref_gn = g[key + 'grad_norm']
    # gradient norms track the fp32 reference (loose: fp16 rounding on an ill-conditioned random-init net)
    ratio = gn / np.maximum(ref_gn, 1e-12)
    assert 0.8 < np.median(ratio) < 1.25


def test_train_step_matches_oracle_with_collisions():
    """Fresh batch with two targets sharing a cell in every head (App. A #7), fp32, vs the oracle end to end."""
    from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
    from oracle import darknet_ref
    cfg = os.path.join(CFG, 'complex_yolov4_tiny.cfg')
    model = _model('complex_yolov4_tiny.cfg', True, 'f32')
    model.train()
    x, tg = syn.bev_images(2, 320, seed=9), syn.targets(2, 4, 320, seed=9, collide=True)
    loss, out = model(x.to(DEV), tg.to(DEV))
    loss.backward()
    net = darknet_ref.DarknetRef(parse_cfg(cfg))
    ps, bs = net.param_shapes()
    params = {k: v.requires_grad_(True) for k, v in syn.fill_state_dict(ps).items()}
    o_ref, l_ref, _ = net.forward(params, x, tg, True, True, syn.fill_state_dict(bs))
    l_ref.sum().backward()
    from tests.util import grad_rel_errors
    np.testing.assert_allclose(float(loss.detach()), float(l_ref.detach()), rtol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), o_ref.detach().numpy(), rtol=2e-3, atol=2e-3)
    errs = grad_rel_errors([(n, p.grad.cpu()) for n, p in model.named_parameters()], {k: v.grad for k, v in params.items()})
    print('grad rel err: median %.2e max %.2e' % (float(np.median(list(errs.values()))), max(errs.values())))
    # A leaky-ReLU pre-activation within fp32 round-off of 0 flips slope between two evaluation orders; on this batch one
    # such element carries a large head gradient and moves every upstream gradient by ~5% (reproduced with the CPU
    # simulator, so it is a property of the problem, not of the kernels).  Tight gradient parity: test_mini_cfg_*.
    assert max(errs.values()) < 1.0


@pytest.mark.parametrize('dtype', ['f32', 'f16'])
def test_mini_cfg_all_block_types(dtype):
    """All-Mish mini cfg (tests/util.py): grouped route, copied cat member, alias route, fused shortcut, SPP max-pools,
    upsample, two heads.  Smooth activations -> every parameter gradient is compared tightly in fp32 mode."""
    from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
    from oracle import darknet_ref
    from tests.util import grad_rel_errors, mini_cfg_path
    cfg = mini_cfg_path()
    torch.manual_seed(0)
    model = Darknet(cfg, use_giou_loss=True, dtype=dtype)
    sd = model.state_dict()
    sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
    model.load_state_dict(sd)
    model.to(DEV).train()
    x, tg = syn.bev_images(2, 64, seed=4, sparsity=0.5), syn.targets(2, 3, 64, seed=4, collide=True)
    loss, out = model(x.to(DEV), tg.to(DEV))
    loss.backward()
    net = darknet_ref.DarknetRef(parse_cfg(cfg))
    ps, bs = net.param_shapes()
    params = {k: v.requires_grad_(True) for k, v in syn.fill_state_dict(ps).items()}
    o_ref, l_ref, _ = net.forward(params, x, tg, True, True, syn.fill_state_dict(bs))
    l_ref.sum().backward()
    errs = grad_rel_errors([(n, p.grad.cpu()) for n, p in model.named_parameters()], {k: v.grad for k, v in params.items()})
    rel_loss = abs(float(loss.detach()) - float(l_ref.detach())) / float(l_ref.detach())
    print('%s: loss rel %.2e, grad rel err median %.2e max %.2e' % (dtype, rel_loss, float(np.median(list(errs.values()))), max(errs.values())))
    if dtype == 'f32':
        assert rel_loss < 1e-4
        np.testing.assert_allclose(out.cpu().numpy(), o_ref.detach().numpy(), rtol=1e-3, atol=1e-3)
        assert max(errs.values()) < 2e-3, sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    else:
        assert rel_loss < 1e-2
        assert np.median(list(errs.values())) < 0.15 and max(errs.values()) < 0.6


def test_inference_and_nms_pipeline():
    """evaluate.py path: model.eval()(imgs) -> post_processing_v2 (reference evaluate.py:44-45)."""
    from complex_yolov4_pytorch_amd.utils.evaluation_utils import post_processing_v2
    model = _model('complex_yolov4_tiny.cfg', True, 'f16')
    model.eval()
    with torch.no_grad():
        out = model(syn.bev_images(2, 608, seed=3).to(DEV))
    assert tuple(out.shape) == (2, 5415, 10) and not out.is_cuda
    dets = post_processing_v2(out, conf_thresh=0.5, nms_thresh=0.5)
    assert len(dets) == 2
    for d in dets:
        assert d is None or (d.shape[1] == 9 and not d.is_cuda)


def test_dynamic_loss_scale_backs_off_and_trains():
    """fp16 mini net with DynamicLossScale started at an absurd scale: the overflowing steps are skipped on the device
    (parameters untouched), the scale backs off until gradients are finite, then steps go through; the gradient of a
    clean scaled step equals the unscaled one (the scale is divided out in the fp32 reductions)."""
    from complex_yolov4_pytorch_amd.optim import DynamicLossScale, FusedAdam
    from tests.util import mini_cfg_path
    torch.manual_seed(0)
    model = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f16')
    sd = model.state_dict()
    sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
    model.load_state_dict(sd)
    model.to(DEV).train()
    x, tg = syn.bev_images(2, 64, seed=4, sparsity=0.5).to(DEV), syn.targets(2, 3, 64, seed=4).to(DEV)
    loss, _ = model(x, tg)
    loss.backward()
    g_ref = model.flat_grad.clone()
    opt = FusedAdam(model.parameters(), lr=1e-4)
    scaler = DynamicLossScale(model, opt, init_scale=2.0 ** 40, backoff_factor=0.5 ** 8, growth_interval=1000)
    p0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    scales, moved = [], []
    for step in range(10):
        opt.zero_grad(set_to_none=True)
        loss, _ = model(x, tg)
        loss.backward()
        scaler.check()
        opt.step()
        scaler.update()
        scales.append(scaler.scale)
        moved.append(not torch.equal(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), p0))
    assert scaler.skipped >= 1 and scales[-1] < 2.0 ** 40          # it overflowed and backed off ...
    assert not moved[0] and moved[-1]                                # ... the first step was skipped, later ones were taken
    assert all(torch.isfinite(p).all() for p in model.parameters())
    # a clean step at the final scale reproduces the unscaled gradient (fp16 rounding of the scaled activations aside)
    model2 = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f16', loss_scale=256.0)
    model2.load_state_dict(sd)
    model2.to(DEV).train()
    loss2, _ = model2(x, tg)
    loss2.backward()
    rel = float((model2.flat_grad - g_ref).norm() / g_ref.norm())
    assert rel < 0.2, rel
