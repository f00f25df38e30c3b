"""GPU parity on BASELINE.json's own shapes ("close the parity envelope"): complex_yolov4.cfg at 608x608 batch 16
and 1024x1024 in the fp32 parity mode against the oracle, layer shapes of the 608 / 304 / 152 grids at operator level, the
bf16 storage mode's band next to fp16's, and run-to-run determinism of the deterministic mode."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.ops as ops  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet  # noqa: E402
from complex_yolov4_pytorch_amd.ops import CY_BF16, CY_F16, CY_F32, View  # noqa: E402

CFG = os.path.join(os.path.dirname(__file__), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg')
DEV = 'cuda'


def _model(cfg, dtype, **kw):
    torch.manual_seed(0)
    m = Darknet(os.path.join(CFG, cfg), use_giou_loss=True, dtype=dtype, **kw)
    sd = m.state_dict()
    sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
    m.load_state_dict(sd)
    return m.to(DEV)


# bounds on the element-wise gradient-head error of the fp32 parity mode (measured: see DESIGN.md section 4)
# (round 3, MI355X: median 4.1-4.2e-2, 90th percentile 7.2-7.5e-2, max 0.15-0.17 on all three shapes -- the fp32 summation
# order of 110 layers against the reference's; the per-tensor NORMS of the same gradients agree to 1.7 %)
ELEMWISE_MEDIAN = {'b16_608': 8e-2, 'b2_1024': 8e-2, 'b2_1216m': 8e-2}
ELEMWISE_P90 = {'b16_608': 0.15, 'b2_1024': 0.15, 'b2_1216m': 0.15}
MOSAIC_SEEDS = (3, 11)        # tests/golden/make_golden_big.py


def mosaic_batch(g, key, seed0=60, B=2, half=608, nt=6):
    """The b2_1216m case's input, assembled ON THE DEVICE (data_process.transformation.make_mosaic) from the same seeded
    tiles and ``random`` seeds the reference's KittiDataset.load_mosaic got in make_golden_big.py; checked against the
    golden's canvas sample and target rows (bit-identical) before it feeds the train step."""
    import random
    from complex_yolov4_pytorch_amd.data_process import transformation as T
    canvases, rows = [], []
    for b in range(B):
        tiles = [syn.bev_images(1, half, seed=seed0 + 10 * b + k)[0].to(DEV) for k in range(4)]
        tts = [syn.targets(1, nt, half, seed=seed0 + 10 * b + k) for k in range(4)]
        random.seed(MOSAIC_SEEDS[b])
        c, t = T.make_mosaic(tiles, tts, half, random_padding=True)
        t[:, 0] = b
        canvases.append(c); rows.append(t)
    x, tg = torch.stack(canvases), torch.cat(rows, 0)
    np.testing.assert_array_equal(x[:, :, ::97, ::89].cpu().numpy(), g[key + 'canvas_rows'])
    np.testing.assert_array_equal(tg.cpu().numpy(), g[key + 'targets'])
    return x, tg


def grad_head_errors(model, ref_heads):
    """Element-wise agreement of the first 8 entries of every parameter gradient with the reference's (``grad_head`` of the
    goldens): per tensor max |g - ref| over max |ref| -> array over the 327 tensors."""
    gh = np.stack([p.grad.reshape(-1)[:8].float().cpu().numpy() for _, p in model.named_parameters()])
    return (np.abs(gh - ref_heads) / (np.abs(ref_heads).max(1, keepdims=True) + 1e-12)).max(1)


@pytest.mark.parametrize('tag,B,S', [('b16_608', 16, 608), ('b2_1024', 2, 1024), ('b2_1216m', 2, 1216)])
def test_v4_fp32_parity_at_baseline_shapes(golden, tag, B, S):
    """BASELINE configs[1] (608x608, batch 16), configs[4]'s resolution (1024x1024) and configs[2]'s input (1216x1216 mosaic
    canvases of four 608x608 maps, built by the device mosaic kernels), fp32 parity mode, one train step against THE
    REFERENCE's own result on the same seeded batch (tests/golden/darknet_big.npz, make_golden_big.py): loss 1e-4 relative,
    probabilities 1e-3, the 18 metrics per head, every parameter-gradient tensor by norm AND element-wise on the golden's
    gradient heads, BatchNorm running statistics."""
    from tests.golden.make_golden import METRIC_KEYS
    g = golden('darknet_big')
    key = tag + '_'
    model = _model('complex_yolov4.cfg', 'f32', deterministic=True)
    model.train()
    if tag.endswith('m'):
        x, tg = mosaic_batch(g, key)
    else:
        x, tg = syn.bev_images(B, S, seed=21), syn.targets(B, 6, S, seed=21)
    loss, out = model(x.to(DEV), tg.to(DEV))
    loss.backward()
    assert list(out.shape) == list(g[key + 'out_shape'])
    l_ref = float(g[key + 'loss'][0])
    rel = abs(float(loss.detach()) - l_ref) / abs(l_ref)
    got, ref = out[:, ::97].cpu().numpy(), g[key + 'out_rows']
    dprob = float(np.abs(got[..., 6:] - ref[..., 6:]).max())
    dim = float(np.abs(got[..., 4:6] - ref[..., 4:6]).max())
    dbox = float((np.abs(got[..., :4] - ref[..., :4]) / (np.abs(ref[..., :4]) + 1.0)).max())
    gn = np.asarray([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    rn = g[key + 'grad_norm']
    ok = rn > 1e-12
    ratio = gn[ok] / rn[ok]
    print('v4 %dx%d B%d f32 vs reference: loss rel %.2e, probabilities |d| %.2e, im/re |d| %.2e, boxes rel %.2e, '
          'grad-norm ratio median %.5f min %.4f max %.4f' % (S, S, B, rel, dprob, dim, dbox, float(np.median(ratio)),
                                                            float(ratio.min()), float(ratio.max())))
    assert rel < 1e-4
    assert dprob < 1e-3                                    # objectness / class probabilities ("conf/class logits within 1e-3")
    assert dim < 5e-3 and dbox < 5e-3                      # regression outputs after 110 fp32 layers (exp / scale amplified)
    met = [[yl.metrics[k] for k in METRIC_KEYS] for yl in model.yolo_layers]
    np.testing.assert_allclose(met, g[key + 'metrics'], rtol=2e-3, atol=1e-5)
    assert abs(float(np.median(ratio)) - 1.0) < 5e-3 and 0.95 < ratio.min() and ratio.max() < 1.05
    # element-wise (norms only).  Batch-16 BatchNorm conditions the net far better than the batch-1 case
    # of test_gpu_model.py; what remains is the leaky-ReLU kink sensitivity of DESIGN.md section 4 on a few tensors
    err = grad_head_errors(model, g[key + 'grad_head'])
    print('   gradient heads (8 elements x 327 tensors) rel err: median %.2e, 90th pct %.2e, max %.2e'
          % (float(np.median(err)), float(np.percentile(err, 90)), float(err.max())))
    assert np.median(err) < ELEMWISE_MEDIAN[tag] and np.percentile(err, 90) < ELEMWISE_P90[tag]
    sd = model.state_dict()
    bn = np.stack([sd[str(n)][:8].cpu().numpy() for n in g[key + 'bn_names']])
    np.testing.assert_allclose(bn, g[key + 'bn_head'], rtol=5e-4, atol=1e-5)


@pytest.mark.parametrize('tag,cfg,B,S', [('tiny', 'complex_yolov4_tiny.cfg', 2, 608), ('v4', 'complex_yolov4.cfg', 1, 416)])
def test_train_step_bf16_band(golden, tag, cfg, B, S):
    """bf16 storage mode (north_star "MFMA fp16/bf16"): same fp32 accumulation, 8 mantissa bits instead of 11, fp32's
    exponent range -- no loss scaling anywhere.  Its band against the reference goldens is stated next to fp16's
    (tests/test_gpu_model.py::test_train_step_f16_band)."""
    g = golden('darknet')
    key = '%s_giou_' % tag
    ref_loss = float(g[key + 'loss'][0])
    stats = {}
    for dt in ('bf16', 'f16'):
        model = _model(cfg, dt)
        assert model.loss_scale == 1.0
        model.train()
        x, tg = syn.bev_images(B, S, seed=1).to(DEV), syn.targets(B, 6, S, seed=1).to(DEV)
        loss, out = model(x, tg)
        loss.sum().backward()
        rel = abs(float(loss.detach()) - ref_loss) / ref_loss
        dprob = np.abs(out[:, ::97].cpu().numpy()[..., 6:] - g[key + 'out_rows'][..., 6:])
        gn = np.asarray([float(p.grad.double().norm()) for _, p in model.named_parameters()])
        assert np.all(np.isfinite(gn))
        ratio = gn / np.maximum(g[key + 'grad_norm'], 1e-12)
        stats[dt] = (rel, float(np.median(dprob)), float(dprob.max()), float(np.median(ratio)))
        print('%s %s: loss rel %.2e, prob median |d| %.2e max %.2e, grad-norm ratio median %.3f' % ((tag, dt) + stats[dt]))
    rel, med, mx, rat = stats['bf16']
    if tag == 'tiny':
        assert rel < 2e-2 and mx < 0.1
    else:
        # 110 bf16 layers (8 mantissa bits), random init, batch-1 BatchNorm: measured 0.08-0.12 from run to run (the per-layer
        # kernel choice and the atomics' order move it); fp16's band on the same case is 1.5e-2 (11 bits)
        assert rel < 0.25 and med < 0.15 and mx < 0.7
    assert 0.7 < rat < 1.4


def test_bf16_trains_without_loss_scaling():
    """100 Adam steps on one synthetic batch in bf16 at loss scale 1: the loss falls like the fp16 run's."""
    from complex_yolov4_pytorch_amd.optim import FusedAdam
    from tests.util import mini_cfg_path
    torch.manual_seed(0)
    model = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='bf16')
    sd = model.state_dict()
    sd.update({k: syn.fill_tensor(k, tuple(v.shape)) for k, v in sd.items() if v.dtype.is_floating_point})
    model.load_state_dict(sd)
    model.to(DEV).train()
    x, tg = syn.bev_images(4, 64, seed=4, sparsity=0.5).to(DEV), syn.targets(4, 3, 64, seed=4).to(DEV)
    opt = FusedAdam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(60):
        opt.zero_grad(set_to_none=True)
        loss, _ = model(x, tg)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert np.all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


@pytest.mark.parametrize('dtype', ['f16', 'bf16', 'f32'])
def test_deterministic_mode_repeats_are_bit_identical(dtype):
    """the default mode's fp32 atomics into shared bins (BatchNorm statistics) reorder sums from run to
    run; the reference's CPU path is deterministic.  With deterministic=True two repeats of the v4 train step give
    bit-identical loss, outputs and flat gradient (batch 16 at 608x608 in the 16-bit modes; batch 4 in fp32 to bound time)."""
    B = 4 if dtype == 'f32' else 16
    model = _model('complex_yolov4.cfg', dtype, deterministic=True)
    model.train()
    x, tg = syn.bev_images(B, 608, seed=5).to(DEV), syn.targets(B, 6, 608, seed=5).to(DEV)
    runs = []
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        loss, out = model(x, tg)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), out.clone(), model.flat_grad.clone()))
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0])
        assert torch.equal(r[1], runs[0][1])
        assert torch.equal(r[2], runs[0][2]), float((r[2] - runs[0][2]).abs().max())
    assert float(runs[0][2].abs().max()) > 0


BIG_CONV = [
    # N, Cin, H, Cout, ks, stride -- the 608 / 304 / 152 grids of complex_yolov4.cfg (layer 0, 1, 2, 5 and a 152-grid 3x3)
    (2, 3, 608, 32, 3, 1),
    (2, 32, 608, 64, 3, 2),
    (2, 64, 304, 64, 1, 1),
    (2, 64, 304, 32, 1, 1),
    (2, 64, 152, 64, 3, 1),
]


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16, CY_F32])
@pytest.mark.parametrize('case', BIG_CONV)
def test_conv_large_grids(dt, case):
    """Forward + BN statistics, dgrad and wgrad at the big-grid layer shapes (op tests stopped at 64x64)."""
    N, Ci, H, Co, ks, st = case
    pad = (ks - 1) // 2
    ch = ops.chunk(dt)
    tol = dict(rtol=1.6e-2, atol=1.6e-2) if dt == CY_BF16 else (dict(rtol=2e-3, atol=2e-3) if dt == CY_F16 else dict(rtol=1e-4, atol=2e-5))
    rnd = (lambda t: t.bfloat16().float()) if dt == CY_BF16 else ((lambda t: t.half().float()) if dt == CY_F16 else (lambda t: t))
    g = torch.Generator().manual_seed(31)
    x = rnd(torch.randn(N, Ci, H, H, generator=g))
    w = rnd(torch.randn(Co, Ci, ks, ks, generator=g) / math.sqrt(Ci * ks * ks))
    ref = F.conv2d(x.double(), w.double(), None, st, pad)
    OH = ref.shape[2]
    cpad = (Ci + ch - 1) // ch * ch
    xv = View.from_nchw(x.to(DEV), dt, cpad=cpad)
    wf, wd = ops.pack_weights(w.to(DEV), Co, cpad, dt)
    out = View.alloc(N, OH, OH, Co, dt, zero=True)
    rows = ops.conv_stats_rows(N * OH * OH, Co)
    stats = torch.zeros(rows, 2, Co, device=DEV)
    ops.conv_igemm(xv, wf, Co, out, ks, st, pad, flags=ops.CONV_STATS, stats=stats)
    torch.testing.assert_close(out.to_nchw().cpu(), ref.float(), **tol)
    s = stats.sum(0).cpu().double()
    torch.testing.assert_close(s[0], ref.sum((0, 2, 3)), rtol=1e-3, atol=N * OH * OH * 1e-5)
    torch.testing.assert_close(s[1], (ref ** 2).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    dy = rnd(torch.randn(N, Co, OH, OH, generator=g))
    dyv = View.from_nchw(dy.to(DEV), dt)
    if Ci % ch == 0:
        gref = torch.nn.grad.conv2d_input((N, Ci, H, H), w.double(), dy.double(), st, pad).float()
        dx = View.alloc(N, H, H, Ci, dt, zero=True)
        ops.conv_igemm(dyv, wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED)
        torch.testing.assert_close(dx.to_nchw().cpu(), gref, **tol)
    wref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, ks, ks), dy.double(), st, pad).float()
    M = N * OH * OH
    split = ops.wgrad_split(M, Co, cpad, ks)
    part = torch.empty(split * Co * ks * ks * cpad, device=DEV)
    ops.conv_wgrad(dyv, xv, ks, st, pad, part, split)
    gw = torch.zeros(Co, Ci, ks, ks, device=DEV)
    ops.wgrad_reduce(part, split, Co, cpad, ks, Co, Ci, 1.0, False, gw)
    # sums of ~M products of unit-variance numbers: compare relative to the tensor's scale
    scale = float(wref.abs().max())
    assert float((gw.cpu() - wref).abs().max()) <= (3e-2 if dt == CY_BF16 else 4e-3 if dt == CY_F16 else 2e-4) * scale


def _sha(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32).tobytes()).hexdigest()


def test_device_augmentation_matches_reference(golden):
    """SURVEY section 8f row 1, VERDICT r1 missing #2: Horizontal_Flip / Cutout / Compose and the mosaic of
    KittiDataset.load_mosaic on the device against the reference's own outputs (tests/golden/aug.npz, produced by
    make_golden_aug.py from the unmodified reference with the same host RNG seeds): pixels and target rows bit-identical."""
    import random
    from complex_yolov4_pytorch_amd.data_process import transformation as T
    g = golden('aug')
    img = syn.bev_images(1, 608, seed=51)[0].to(DEV)
    tg = syn.targets(1, 6, 608, seed=51)
    np.random.seed(1)
    fi, ft = T.Horizontal_Flip(p=1.0)(img.clone(), tg.clone())
    assert fi.is_cuda and _sha(fi) == str(g['flip_sha'])
    np.testing.assert_array_equal(ft.cpu().numpy(), g['flip_targets'])
    np.random.seed(int(g['cut_seed'][0]))
    ci, ct = T.Cutout(n_holes=3, ratio=0.25, fill_value=0.5, p=1.0)(img.clone(), tg.clone())
    assert _sha(ci) == str(g['cut_sha'])
    np.testing.assert_array_equal(ct.cpu().numpy(), g['cut_targets'])
    assert 0 < ct.shape[0] < tg.shape[0]                          # the case drops a target
    comp = T.Compose([T.Horizontal_Flip(p=1.0), T.Cutout(n_holes=12, ratio=0.1, fill_value=0.0, p=1.0)], p=1.0)
    np.random.seed(77)
    pi, pt = comp(img.clone(), tg.clone())
    assert _sha(pi) == str(g['comp_sha'])
    np.testing.assert_array_equal(pt.cpu().numpy(), g['comp_targets'])
    np.random.seed(5)                                              # p gates: nothing happens, inputs returned untouched
    tgd = tg.to(DEV)
    ni, nt = T.Horizontal_Flip(p=0.0)(img, tgd)
    assert ni is img and nt is tgd
    _, nh = T.Cutout(n_holes=2, ratio=0.1, p=0.0)(img, tg)         # host targets come back on the device, gate fired or not
    assert nh.is_cuda and torch.equal(nh.cpu(), tg)
    tiles = [syn.bev_images(1, 608, seed=60 + k)[0].to(DEV) for k in range(4)]
    tts = [syn.targets(1, 6, 608, seed=60 + k) for k in range(4)]
    for tag, rp in (('mosaic_fixed', False), ('mosaic_rand_a', True), ('mosaic_rand_b', True)):
        random.seed(int(g[tag + '_seed'][0]))
        canvas, targets = T.make_mosaic(tiles, [t.clone() for t in tts], 608, random_padding=rp)
        assert tuple(canvas.shape) == (3, 1216, 1216) and _sha(canvas) == str(g[tag + '_sha']), tag
        np.testing.assert_array_equal(canvas[:, ::97, ::89].cpu().numpy(), g[tag + '_rows'])
        np.testing.assert_array_equal(targets.cpu().numpy(), g[tag + '_targets'])
    # the mosaic canvas feeds the train step (BASELINE configs[2]: "mosaic aug on" = a 1216 x 1216 input)
    model = _model('complex_yolov4_tiny.cfg', 'f16')
    model.train()
    t = targets.clone(); t[:, 0] = 0
    loss, out = model(canvas[None], t.to(DEV))
    loss.backward()
    assert np.isfinite(float(loss.detach())) and out.shape[1] == 3 * (76 * 76 + 38 * 38)


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16, CY_F32])
@pytest.mark.parametrize('C,M,act', [(32, 5000, 'mish'), (128, 3000, 'leaky'), (256, 777, 'mish'), (1024, 361, 'linear')])
def test_fused_bn_finalisers_match_the_two_kernel_path(dt, C, M, act):
    """cy_bn_act_fwd_fused / cy_bn_act_bwd_apply_fused (statistics fold in the consumer's prologue, alternating tables)
    against cy_bn_finalize + cy_bn_act_fwd and cy_bn_bwd_finalize + cy_bn_act_bwd_apply on the same tables."""
    g = torch.Generator().manual_seed(C + M)
    tdt = ops.torch_dtype(dt)
    x = View(torch.randn(M * C, generator=g).to(DEV).to(tdt), 0, 1, 1, M, C, C, dt)
    res = View(torch.randn(M * C, generator=g).to(DEV).to(tdt), 0, 1, 1, M, C, C, dt)
    rows = ops.conv_stats_rows(M, C)
    xf = x.buf.float().view(M, C)
    bins = torch.zeros(rows, 2, C, device=DEV)
    for b in range(rows):                                     # what the conv epilogue leaves: per-bin partial sums
        part = xf[b::rows]
        bins[b, 0], bins[b, 1] = part.sum(0), (part * part).sum(0)
    gamma, beta = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV)
    rm0, rv0 = torch.randn(C, generator=g).to(DEV), torch.rand(C, generator=g).to(DEV) + 0.5
    a = ops.ACT[act]
    # two-kernel path
    rm1, rv1, nbt1 = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device=DEV)
    vec1 = torch.empty(4, C, device=DEV)
    t1 = bins.clone()
    ops.bn_finalize(t1, rows, C, M, gamma, beta, rm1, rv1, nbt1, 0.1, 1e-5, vec1[0], vec1[1], vec1[2], vec1[3])
    y1 = View.alloc(1, 1, M, C, dt)
    ops.bn_act_fwd(x, y1, res, vec1[2], vec1[3], a)
    assert float(t1.abs().max()) == 0.0
    # fused
    rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), torch.zeros(1, dtype=torch.int64, device=DEV)
    vec2 = torch.empty(4, C, device=DEV)
    t2, other = bins.clone(), torch.ones(rows * 2 * C + 100, device=DEV)
    y2 = View.alloc(1, 1, M, C, dt)
    ops.bn_act_fwd_fused(x, y2, res, t2, rows, gamma, beta, rm2, rv2, nbt2, 0.1, 1e-5, vec2, other, a)
    assert float(other.abs().max()) == 0.0 and torch.equal(t2, bins)          # the OTHER table is zeroed, the read one kept
    torch.testing.assert_close(vec2, vec1, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(rm2, rm1, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(rv2, rv1, rtol=1e-6, atol=1e-7)
    assert int(nbt2) == int(nbt1) == 1
    tol = dict(rtol=1.6e-2, atol=1.6e-2) if dt == CY_BF16 else (dict(rtol=2e-3, atol=2e-3) if dt == CY_F16 else dict(rtol=1e-5, atol=1e-5))
    torch.testing.assert_close(y2.buf.float(), y1.buf.float(), **tol)
    # backward
    dy = View(torch.randn(M * C, generator=g).to(DEV).to(tdt), 0, 1, 1, M, C, C, dt)
    prow = ops.bn_bwd_rows(M, C, dt)
    part = torch.zeros(prow, 2, C, device=DEV)
    ops.bn_act_bwd_reduce(x, dy, vec1[0], vec1[1], vec1[2], vec1[3], a, part, prow)
    p1 = part.clone()
    dgs, dbs = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    gg1, gb1 = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    ops.bn_bwd_finalize(p1, prow, C, dgs, dbs, gg1, gb1, 0.5)
    dx1, rg1 = View.alloc(1, 1, M, C, dt), View.alloc(1, 1, M, C, dt, zero=True)
    ops.bn_act_bwd_apply(x, dy, dx1, rg1, True, vec1[0], vec1[1], vec1[2], vec1[3], dgs, dbs, a)
    gg2, gb2 = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    other = torch.ones(prow * 2 * C, device=DEV)
    dx2, rg2 = View.alloc(1, 1, M, C, dt), View.alloc(1, 1, M, C, dt, zero=True)
    ops.bn_act_bwd_apply_fused(x, dy, dx2, rg2, True, vec1[0], vec1[1], vec1[2], vec1[3], part, prow, gg2, gb2, 0.5, other, a)
    assert float(other.abs().max()) == 0.0
    torch.testing.assert_close(gg2, gg1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gb2, gb1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx2.buf.float(), dx1.buf.float(), **tol)
    assert torch.equal(rg2.buf, rg1.buf)


# (Cin of the conv = channels of the incoming gradient dy, Cout = channels of the produced gradient, ks, H, N)
DGRAD_SUMS = [(64, 128, 3, 38, 4), (128, 64, 1, 76, 2), (256, 256, 3, 19, 16), (512, 1024, 1, 19, 3), (64, 64, 3, 152, 1)]


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16])
@pytest.mark.parametrize('accum', [False, True])
@pytest.mark.parametrize('act', ['mish', 'leaky', 'linear'])
@pytest.mark.parametrize('case', DGRAD_SUMS)
def test_dgrad_bn_sums_epilogue_matches_separate_reduce(dt, accum, act, case):
    """cy_conv_dgrad_bn_sums = cy_conv_igemm (dgrad, with / without fan-in accumulation) followed by cy_bn_act_bwd_reduce
    over (raw, stored gradient): same gradient tensor bit for bit, same (d beta, d gamma) sums up to fp32 summation order."""
    Cdy, Cg, ks, H, N = case
    pad = (ks - 1) // 2
    g = torch.Generator().manual_seed(Cdy * 3 + Cg + H)
    tdt = ops.torch_dtype(dt)
    dy = View.alloc(N, H, H, Cdy, dt); dy.buf.copy_(torch.randn(dy.buf.numel(), generator=g).to(tdt))
    raw = View.alloc(N, H, H, Cg, dt); raw.buf.copy_(torch.randn(raw.buf.numel(), generator=g).to(tdt))
    w = torch.randn(Cdy, Cg, ks, ks, generator=g).to(DEV) * (1.0 / (ks * ks * Cdy) ** 0.5)
    _, wd = ops.pack_weights(w, Cdy, Cg, dt)
    vec = torch.stack([torch.randn(Cg, generator=g) * 0.1, torch.rand(Cg, generator=g) + 0.5,
                       torch.rand(Cg, generator=g) + 0.5, torch.randn(Cg, generator=g) * 0.2]).to(DEV)
    a = ops.ACT[act]
    base = torch.randn(raw.buf.numel(), generator=g).to(DEV).to(tdt)
    flags = ops.CONV_TRANSPOSED | (ops.CONV_ACCUM if accum else 0)
    M = N * H * H
    rows = ops.conv_stats_rows(M, Cg)
    ref_sums = None
    for hint in (2, 3, 4, 5, 6, 7, 8, 9):
        if hint in (3, 8) and Cg <= 64:
            continue
        # separate: the same kernel / tile without the sums, then the reduce pass
        g1 = View.alloc(N, H, H, Cg, dt); g1.buf.copy_(base)
        ops.conv_igemm(dy, wd, Cg, g1, ks, 1, pad, flags=flags, tile=hint)
        if ref_sums is None:
            part = torch.zeros(ops.bn_bwd_rows(M, Cg, dt), 2, Cg, device=DEV)
            ops.bn_act_bwd_reduce(raw, g1, vec[0], vec[1], vec[2], vec[3], a, part)
            ref_sums = part.double().sum(0)
        g2 = View.alloc(N, H, H, Cg, dt); g2.buf.copy_(base)
        tbl = torch.zeros(rows, 2, Cg, device=DEV)
        ops.conv_dgrad_bn_sums(dy, wd, Cg, g2, ks, 1, pad, raw, vec[0], vec[1], vec[2], vec[3], a, tbl, flags=flags, tile=hint)
        if not accum:
            assert torch.equal(g2.buf, g1.buf), hint     # (accumulating launches take the direct-store epilogue without the sums)
        else:
            tol = 2e-2 if dt == CY_BF16 else 3e-3
            torch.testing.assert_close(g2.buf.float(), g1.buf.float(), rtol=tol, atol=tol)
            part = torch.zeros(ops.bn_bwd_rows(M, Cg, dt), 2, Cg, device=DEV)
            ops.bn_act_bwd_reduce(raw, g2, vec[0], vec[1], vec[2], vec[3], a, part)
            ref_sums = part.double().sum(0)
        got = tbl.double().sum(0)
        scale = ref_sums.abs().max(1, keepdim=True).values + 1e-6
        assert float(((got - ref_sums).abs() / scale).max()) < 2e-5, (hint, float(((got - ref_sums).abs() / scale).max()))


def test_dgrad_bn_sums_rejects_what_the_pipelined_kernel_cannot_run():
    dy, raw, gv = View.alloc(1, 8, 8, 32, CY_F16), View.alloc(1, 8, 8, 64, CY_F16), View.alloc(1, 8, 8, 64, CY_F16)
    wd = torch.zeros(64, 32, dtype=torch.float16, device=DEV)
    v = torch.ones(64, device=DEV)
    tbl = torch.zeros(16 * 2 * 64, device=DEV)
    with pytest.raises(ops.CyoloError):      # 32 gradient channels: not a multiple of 64
        ops.conv_dgrad_bn_sums(dy, wd, 64, gv, 1, 1, 0, raw, v, v, v, v, ops.ACT['mish'], tbl, flags=ops.CONV_TRANSPOSED)
    dy32 = View.alloc(1, 8, 8, 64, CY_F32)
    with pytest.raises(ops.CyoloError):      # fp32 parity mode keeps the separate pass
        ops.conv_dgrad_bn_sums(dy32, wd, 64, View.alloc(1, 8, 8, 64, CY_F32), 1, 1, 0, View.alloc(1, 8, 8, 64, CY_F32), v, v, v, v,
                               ops.ACT['mish'], tbl, flags=ops.CONV_TRANSPOSED)


@pytest.mark.parametrize('dtype', ['f16', 'bf16'])
def test_model_backward_with_dgrad_sums_matches_separate_reduce(monkeypatch, dtype):
    """complex_yolov4.cfg train step with every marked layer taking its BN-backward sums from the dgrad epilogue
    (CY_DGRAD_BN_SUMS=2) against the same step with the separate reduce pass (=0): same loss, parameter gradients equal up
    to the run-to-run noise band of the default (atomics) mode."""
    x, tg = syn.bev_images(4, 416, seed=3), syn.targets(4, 6, 416, seed=3)
    grads = {}
    for mode in ('0', '2', '0'):
        monkeypatch.setenv('CY_DGRAD_BN_SUMS', mode)
        m = _model('complex_yolov4.cfg', dtype)
        m.train()
        loss, _ = m(x.to(DEV), tg.to(DEV))
        loss.sum().backward()
        eng = next(iter(m._engines.values()))
        assert (len(eng._sums_fused) >= 70) if mode == '2' else not eng._sums_fused
        grads.setdefault(mode, []).append(torch.cat([p.grad.reshape(-1).float() for p in m.parameters()]).cpu())
        assert bool(torch.isfinite(grads[mode][-1]).all())
    a0, a1 = grads['0']
    b = grads['2'][0]
    noise = float((a0 - a1).norm() / a0.norm())
    diff = float((b - a0).norm() / a0.norm())
    assert diff < max(3 * noise, 2e-2), (diff, noise)


def test_two_stage_deterministic_folds():
    """cy_fold_rows (+ finaliser on its output) = finaliser on the full table; cy_bias_grad_det = cy_bias_grad; both are
    bit-identical from call to call."""
    g = torch.Generator().manual_seed(9)
    rows, C, M = 5000, 64, 5000 * 64
    table = torch.randn(rows, 2, C, generator=g).to(DEV)
    table[:, 1] = table[:, 1].abs() * 64 + 70.0 * 64        # second moments: large enough for a positive variance
    gamma, beta = torch.rand(C, generator=g).to(DEV) + 0.5, torch.randn(C, generator=g).to(DEV)
    outs = []
    for two_stage in (False, True, True):
        t = table.clone()
        vec = torch.empty(4, C, device=DEV)
        tbl, r = t, rows
        if two_stage:
            tmp = torch.zeros(256 * 2 * C, device=DEV)
            r = ops.fold_rows(t, rows, 2 * C, tmp)
            assert r == ops.fold_rows_out(rows) <= 128 and float(t.abs().max()) == 0.0      # the rows read are zeroed
            tbl = tmp
        ops.bn_finalize(tbl, r, C, M, gamma, beta, None, None, None, 0.1, 1e-5, vec[0], vec[1], vec[2], vec[3])
        outs.append(vec)
    torch.testing.assert_close(outs[1], outs[0], rtol=1e-5, atol=1e-6)
    assert torch.equal(outs[1], outs[2])
    d = torch.randn(76 * 76 * 4 * 30, generator=g).to(DEV)
    Mh = 76 * 76 * 4
    ref = torch.ones(30, device=DEV)
    ops.bias_grad(d, Mh, 30, 0.5, ref)
    got = []
    for _ in range(2):
        gb = torch.ones(30, device=DEV)
        ops.bias_grad_det(d, Mh, 30, 0.5, gb, torch.empty(256 * 32, device=DEV))
        got.append(gb)
    # cy_bias_grad adds its per-block partial sums with fp32 atomics in whatever order the blocks retire: against the fixed-order
    # result the 23 104-term column sums (|sum| up to ~175) differ by up to ~2e-4 absolute (emulated over 300 random orders of the
    # partials) -- the round-2 bound of 1e-5 + 1e-5 |x| was exceeded by about one order in fifty
    torch.testing.assert_close(got[0], ref, rtol=1e-5, atol=2e-3)
    assert torch.equal(got[0], got[1])


# (N, Cin, H, W, Cout, stride): the three shapes conv_direct.hip serves, at sizes that leave partial tiles on every edge
DIRECT = [(2, 3, 70, 90, 32, 1), (3, 32, 45, 77, 64, 1), (2, 32, 61, 83, 64, 2), (1, 32, 64, 64, 64, 2), (1, 3, 608, 608, 32, 1)]


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16])
@pytest.mark.parametrize('case', DIRECT)
def test_direct_small_cin_conv(dt, case):
    """conv_direct.hip (3 -> 32, 32 -> 64 s1 / s2 forward) against float64 torch and against the 4-wave implicit-GEMM kernel
    (CY_CONV_TILE(1)): training launch (raw output + BN statistics) and eval launch (affine + activation + shortcut)."""
    N, Ci, H, W, Co, st = case
    tol = dict(rtol=1.6e-2, atol=1.6e-2) if dt == CY_BF16 else dict(rtol=2e-3, atol=2e-3)
    rnd = (lambda t: t.bfloat16().float()) if dt == CY_BF16 else (lambda t: t.half().float())
    g = torch.Generator().manual_seed(Ci + H + W)
    x = rnd(torch.randn(N, Ci, H, W, generator=g))
    w = rnd(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(Ci * 9))
    ref = F.conv2d(x.double(), w.double(), None, st, 1)
    OH, OW = ref.shape[2], ref.shape[3]
    cpad = (Ci + 7) // 8 * 8
    xv = View.from_nchw(x.to(DEV), dt, cpad=cpad)
    wf, _ = ops.pack_weights(w.to(DEV), Co, cpad, dt)
    rows = ops.conv_stats_rows(N * OH * OW, Co)
    outs, sts = [], []
    for hint in (0, 1):                                   # 0: library default (the direct kernel), 1: the 4-wave kernel
        out = View.alloc(N, OH, OW, Co, dt, zero=True)
        stats = torch.zeros(rows, 2, Co, device=DEV)
        n0 = ops.direct_launches()
        ops.conv_igemm(xv, wf, Co, out, 3, st, 1, flags=ops.CONV_STATS, stats=stats, tile=hint)
        assert ops.direct_launches() - n0 == (1 if hint == 0 else 0)
        outs.append(out.to_nchw().cpu())
        sts.append(stats.sum(0).cpu().double())
    torch.testing.assert_close(outs[0], ref.float(), **tol)
    assert torch.equal(outs[0], outs[1]) or float((outs[0] - outs[1]).abs().max()) <= tol['atol']
    torch.testing.assert_close(sts[0][0], ref.sum((0, 2, 3)), rtol=1e-3, atol=N * OH * OW * 1e-5)
    torch.testing.assert_close(sts[0][1], (ref ** 2).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    # eval epilogue: act(conv * scale + shift) + res
    scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2
    res = rnd(torch.randn(N, Co, OH, OW, generator=g))
    resv = View.from_nchw(res.to(DEV), dt)
    for act in ('mish', 'leaky'):
        z = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        want = (z * torch.tanh(F.softplus(z)) if act == 'mish' else F.leaky_relu(z, 0.1)) + res.double()
        got = []
        for hint in (0, 1):
            out = View.alloc(N, OH, OW, Co, dt, zero=True)
            ops.conv_bn_act_eval(xv, wf, Co, out, 3, st, 1, scale.to(DEV), shift.to(DEV), ops.ACT[act], resv, tile=hint)
            got.append(out.to_nchw().cpu())
        etol = dict(rtol=2.5e-2, atol=2.5e-2) if dt == CY_BF16 else dict(rtol=4e-3, atol=4e-3)
        torch.testing.assert_close(got[0], want.float(), **etol)
        torch.testing.assert_close(got[0], got[1], **etol)


# (Cin, Cout) instantiations of the 1x1 streaming kernel; M is ragged (not a multiple of the 128 / 256-pixel tile)
PW = [(64, 64), (128, 64), (64, 128), (64, 32), (32, 64)]


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16])
@pytest.mark.parametrize('ci,co', PW)
def test_direct_1x1_stream_conv(dt, ci, co):
    """conv_direct.hip's 1x1 kernel (CY_CONV_TILE(10)) against float64 torch and the 4-wave kernel: forward with BN statistics,
    eval epilogue with shortcut, and the 1x1 dgrad (store and fan-in accumulate), on channel-slice views (ld > C)."""
    N, H, W = 3, 37, 53
    tol = dict(rtol=1.6e-2, atol=1.6e-2) if dt == CY_BF16 else dict(rtol=2e-3, atol=2e-3)
    rnd = (lambda t: t.bfloat16().float()) if dt == CY_BF16 else (lambda t: t.half().float())
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = rnd(torch.randn(N, ci, H, W, generator=g))
    w = rnd(torch.randn(co, ci, 1, 1, generator=g) / math.sqrt(ci))
    ref = F.conv2d(x.double(), w.double())
    xv = View.from_nchw(x.to(DEV), dt, ld=ci + 64)            # a slice of a wider storage
    wf, wd = ops.pack_weights(w.to(DEV), co, ci, dt)
    M = N * H * W
    rows = ops.conv_stats_rows(M, co)
    res = {}
    for hint in (10, 1):
        out = View.from_nchw(torch.zeros(N, co, H, W, device=DEV), dt, ld=co + 32)
        stats = torch.zeros(rows, 2, co, device=DEV)
        n0 = ops.direct_launches()
        ops.conv_igemm(xv, wf, co, out, 1, 1, 0, flags=ops.CONV_STATS, stats=stats, tile=hint)
        assert ops.direct_launches() - n0 == (1 if hint == 10 else 0)
        res[hint] = (out.to_nchw().cpu(), stats.sum(0).cpu().double())
    torch.testing.assert_close(res[10][0], ref.float(), **tol)
    torch.testing.assert_close(res[10][0], res[1][0], **tol)
    torch.testing.assert_close(res[10][1][0], ref.sum((0, 2, 3)), rtol=1e-3, atol=M * 1e-5)
    torch.testing.assert_close(res[10][1][1], (ref ** 2).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    # eval epilogue
    scale, shift = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.2
    r = rnd(torch.randn(N, co, H, W, generator=g))
    rv = View.from_nchw(r.to(DEV), dt)
    z = ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    want = z * torch.tanh(F.softplus(z)) + r.double()
    out = View.alloc(N, H, W, co, dt, zero=True)
    ops.conv_bn_act_eval(xv, wf, co, out, 1, 1, 0, scale.to(DEV), shift.to(DEV), ops.ACT['mish'], rv, tile=10)
    etol = dict(rtol=2.5e-2, atol=2.5e-2) if dt == CY_BF16 else dict(rtol=4e-3, atol=4e-3)
    torch.testing.assert_close(out.to_nchw().cpu(), want.float(), **etol)
    # dgrad of the same layer: dX[p][ci] = sum_co dY[p][co] W[co][ci]; its (Cin, Cout) roles are swapped
    if (co, ci) in PW:
        dy = rnd(torch.randn(N, co, H, W, generator=g))
        dyv = View.from_nchw(dy.to(DEV), dt)
        gref = torch.nn.grad.conv2d_input((N, ci, H, W), w.double(), dy.double())
        base = rnd(torch.randn(N, ci, H, W, generator=g))
        for acc in (False, True):
            dx = View.from_nchw(base.to(DEV), dt)
            n0 = ops.direct_launches()
            ops.conv_igemm(dyv, wd, ci, dx, 1, 1, 0, flags=ops.CONV_TRANSPOSED | (ops.CONV_ACCUM if acc else 0), tile=10)
            assert ops.direct_launches() - n0 == 1
            wantg = gref + (base.double() if acc else 0)
            torch.testing.assert_close(dx.to_nchw().cpu(), wantg.float(), **etol)


def test_direct_1x1_default_threshold():
    """Without a hint the 1x1 stream kernel takes launches of >= 256 k pixels only."""
    for H, expect in ((152, 1), (76, 0)):
        x = View.alloc(16, H, H, 64, CY_F16, zero=True)
        out = View.alloc(16, H, H, 64, CY_F16)
        wf = torch.zeros(64, 64, dtype=torch.float16, device=DEV)
        stats = torch.zeros(ops.conv_stats_rows(16 * H * H, 64) * 2 * 64, device=DEV)
        n0 = ops.direct_launches()
        ops.conv_igemm(x, wf, 64, out, 1, 1, 0, flags=ops.CONV_STATS, stats=stats)
        assert ops.direct_launches() - n0 == expect
