"""Round-3 GPU evidence (VERDICT r2 "next round" #1 and #6): the BENCHMARKED modes on the BENCHMARKED shape against the
reference's golden step (f16 / bf16, complex_yolov4.cfg, 608x608, batch 16), RcclDataParallel over ``nccl`` on hardware,
and deterministic mode across fresh processes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from tests.test_gpu_r2 import DEV, _model, grad_head_errors  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))

# Stated bands of the 16-bit storage modes on BASELINE configs[1]'s own shape, against the reference's fp32 CPU step
# (tests/golden/darknet_big.npz).  Measured values are printed by the test and quoted in DESIGN.md section 4.
# (No element-wise gradient bound HERE: at this random init the net amplifies a 1e-7 perturbation to 4e-2 -- the oracle's own
# float32 and float64 runs differ by that, profiles/r04_oracle_f64_vs_f32.txt -- so 16-bit gradients are uncorrelated with the
# reference's element by element; round 3 "bounded" them at 300 %, which bounded nothing.  The element-wise evidence for the
# 16-bit modes is on a CONDITIONED net, with controls: tests/test_zz_gpu_dynamics.py::test_conditioned_net_16bit_step_agrees_with_fp32 and
# ::test_16bit_step_matches_ideal_16bit_storage.)
BANDS = {
    #        loss rel, prob median, prob max, grad-norm ratio median window
    'f16': dict(loss=1e-2, pmed=4e-2, pmax=0.5, gn=(0.9, 1.1)),
    'bf16': dict(loss=6e-2, pmed=0.15, pmax=0.9, gn=(0.8, 1.25)),
}


@pytest.mark.parametrize('dtype', ['f16', 'bf16'])
def test_v4_16bit_band_at_benchmark_shape(golden, dtype):
    """What bench.py times (default mode, f16 / bf16, v4 at 608x608 batch 16) next to what the reference computes for the
    same seeded batch: loss, decoded probabilities, im/re, parameter-gradient norms and gradient heads element-wise."""
    g = golden('darknet_big')
    key = 'b16_608_'
    model = _model('complex_yolov4.cfg', dtype)
    model.train()
    x, tg = syn.bev_images(16, 608, seed=21), syn.targets(16, 6, 608, seed=21)
    loss, out = model(x.to(DEV), tg.to(DEV))
    loss.backward()
    l_ref = float(g[key + 'loss'][0])
    rel = abs(float(loss.detach()) - l_ref) / abs(l_ref)
    got, ref = out[:, ::97].cpu().numpy(), g[key + 'out_rows']
    dprob = np.abs(got[..., 6:] - ref[..., 6:])
    dim = float(np.abs(got[..., 4:6] - ref[..., 4:6]).max())
    gn = np.asarray([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    assert np.all(np.isfinite(gn))
    rn = g[key + 'grad_norm']
    ok = rn > 1e-12
    ratio = gn[ok] / rn[ok]
    err = grad_head_errors(model, g[key + 'grad_head'])
    print('v4 608x608 B16 %s vs reference fp32: loss rel %.2e, probabilities |d| median %.2e max %.2e, im/re |d| max %.2e, '
          'grad-norm ratio median %.4f min %.3f max %.3f, gradient heads rel err median %.2e 90th pct %.2e'
          % (dtype, rel, float(np.median(dprob)), float(dprob.max()), dim, float(np.median(ratio)), float(ratio.min()),
             float(ratio.max()), float(np.median(err)), float(np.percentile(err, 90))))
    b = BANDS[dtype]
    assert rel < b['loss']
    assert np.median(dprob) < b['pmed'] and dprob.max() < b['pmax']
    assert b['gn'][0] < np.median(ratio) < b['gn'][1]


def _worker(job, tmp_path, name, *args, timeout=600, env_extra=None):
    out = os.path.join(str(tmp_path), name + '.json')
    env = dict(os.environ)
    env.pop('CY_TUNE_RECORD', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, '-m', 'tests.gpu_workers', job, out] + [str(a) for a in args], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with open(out) as f:
        return json.load(f)


@pytest.mark.parametrize('dtype', ['f16', 'f32'])
def test_deterministic_across_processes(tmp_path, dtype):
    """deterministic=True pins every kernel / tile / split-K choice to the persisted table or the shape-only heuristics (no
    timing), so two FRESH processes -- two data-parallel ranks, a resumed run -- produce bit-identical loss, outputs and
    flat gradient (round 2: only repeats inside one process did)."""
    B = 16 if dtype == 'f16' else 2
    a = _worker('det_hash', tmp_path, 'a', dtype, B, 608)
    b = _worker('det_hash', tmp_path, 'b', dtype, B, 608)
    assert a['grad_absmax'] > 0 and np.isfinite(a['loss_value'])
    assert a['fwd_tiles'] == b['fwd_tiles'] and a['wsplit'] == b['wsplit']
    for k in ('loss', 'outputs', 'grad'):
        assert a[k] == b[k], (k, a['loss_value'], b['loss_value'])


def test_rccl_data_parallel_on_hardware(tmp_path):
    """RcclDataParallel over the ``nccl`` (= RCCL) backend on the MI355X, one rank (CY_DDP_FORCE=1): the flat gradient after
    a plain step and after a 2-micro-step ``no_sync()`` accumulation is BIT-equal to the unwrapped model's (deterministic
    mode; SUM over one rank of gradient / 1), every bucket's all-reduce is issued on the wrapper's side stream (not the
    compute stream), tail first, >= 2 buckets, covering the buffer exactly once per optimizer step."""
    r = _worker('rccl', tmp_path, 'rccl')
    assert r['active'] and r['world'] == 1 and r['backend'] == 'nccl'
    assert r['grad_absmax'] > 0
    assert r['step1_equal'], r['step1_maxdiff']
    assert r['step2_equal'], r['step2_maxdiff']
    assert r['loss1'][0] == r['loss1'][1] and r['loss2'][0] == r['loss2'][1]
    for calls in (r['calls_step1'], r['calls_step2']):      # the no_sync() micro-step issues none: one set per step
        assert len(calls) >= 2, calls
        assert all(c['on_side'] and not c['on_default'] for c in calls), calls
        assert sum(c['numel'] for c in calls) == r['total']
    assert r['form'] == 'reduced'


def test_fused_adam_skipped_step_keeps_bias_correction():
    """the step after a device-skipped step must run with bias correction t, not t + 1.  With a skip flag
    attached the step count lives on the device (cy_adam_multi_dev); five steps, the 2nd and 3rd carrying a non-finite
    gradient, against torch.optim.Adam taking only the three finite ones."""
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(5000, generator=g)
    grads = [torch.randn(5000, generator=g) for _ in range(5)]
    grads[1][17] = float('inf')
    grads[2][4000] = float('nan')
    pr = p0.clone().to(DEV).requires_grad_(True)
    ref = torch.optim.Adam([pr], lr=1e-2, weight_decay=1e-3)
    pf = p0.clone().to(DEV).requires_grad_(True)
    opt = FusedAdam([pf], lr=1e-2, weight_decay=1e-3)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    opt.skip_flag = flag
    pf.grad = torch.zeros_like(pf)
    for gr in grads:
        pf.grad.copy_(gr.to(DEV))
        ops.grad_nonfinite(pf.grad, flag)
        opt.step()
        if bool(torch.isfinite(gr).all()):
            pr.grad = gr.to(DEV)
            ref.step()
    torch.testing.assert_close(pf.detach(), pr.detach(), rtol=2e-5, atol=1e-6)
    assert int(opt._step_dev[opt._ping]) == 3          # applied steps only


@pytest.mark.parametrize('dt', ['f16', 'bf16', 'f32'])
@pytest.mark.parametrize('case', [(2, 64, 38, 128, 3, 1), (2, 128, 38, 64, 1, 1), (2, 64, 76, 128, 3, 2), (1, 32, 40, 30, 1, 1)])
def test_wgrad_atomic_single_slab_matches_split_slabs(dt, case):
    """cy_conv_wgrad's atomic mode (every pixel split ADDS into one resident slab; the fold reads that slab and leaves it
    zeroed) against the split-slab mode and float64 torch: same gradient up to fp32 summation order."""
    import math
    import torch.nn.functional as F  # noqa: F401
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.ops import View
    code = ops.dtype_code(dt)
    N, Ci, H, Co, ks, st = case
    pad = (ks - 1) // 2
    rnd = (lambda t: t.bfloat16().float()) if dt == 'bf16' else ((lambda t: t.half().float()) if dt == 'f16' else (lambda t: t))
    g = torch.Generator().manual_seed(77)
    x = rnd(torch.randn(N, Ci, H, H, generator=g))
    OH = (H + 2 * pad - ks) // st + 1
    dy = rnd(torch.randn(N, Co, OH, OH, generator=g))
    wref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, ks, ks), dy.double(), st, pad).float()
    xv, dyv = View.from_nchw(x.to(DEV), code), View.from_nchw(dy.to(DEV), code, cpad=(Co + 31) // 32 * 32)
    cop = dyv.C
    slab = cop * ks * ks * Ci
    grads = {}
    for atomic, split in ((False, 5), (True, 5), (True, 23)):
        part = torch.zeros((1 if atomic else split) * slab, device=DEV)
        ops.conv_wgrad(dyv, xv, ks, st, pad, part, split, atomic=atomic)
        gw = torch.zeros(Co, Ci, ks, ks, device=DEV)
        desc, blocks = ops.make_reduce_table([(part, gw, 1 if atomic else split, cop, Ci, ks, Co, Ci, 1 if atomic else 0)], DEV)
        ops.wgrad_reduce_multi(desc, blocks, 1.0, False)
        torch.cuda.synchronize()
        if atomic:
            assert float(part.abs().max()) == 0.0          # the fold left the resident slab zeroed for the next step
        grads[(atomic, split)] = gw.cpu()
    scale = float(wref.abs().max())
    tol = (3e-2 if dt == 'bf16' else 4e-3 if dt == 'f16' else 2e-4) * scale
    for k, gw in grads.items():
        assert float((gw - wref).abs().max()) <= tol, k
    for k in ((True, 5), (True, 23)):
        assert float((grads[k] - grads[(False, 5)]).abs().max()) <= 2e-5 * scale * math.sqrt(N * OH * OH / 64), k


@pytest.mark.parametrize('dt', ['f16', 'bf16'])
@pytest.mark.parametrize('case', [(2, 96, 19, 320, 3, 1, 3), (1, 128, 38, 256, 3, 2, 2), (3, 256, 19, 512, 1, 1, 1), (2, 64, 24, 288, 3, 1, 4)])
def test_wgrad_eight_wave_tile_matches_torch(tmp_path, dt, case):
    """The 256 x 128 weight-gradient tile (eight compute + four loader waves, three-stage ring with counted vmcnt; taken for layers
    with >= 256 output channels) against float64 torch: whole and partial channel tiles (320 = 256 + 64, 288), partial column tiles
    (9 * 96 = 864), stride 2, 1x1, splits that cut the pixel range unevenly -- and against the 128 x 128 loader-wave kernel, which
    must give the same slabs up to fp32 summation order."""
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.ops import View
    code = ops.dtype_code(dt)
    N, Ci, H, Co, ks, st, split = case
    pad = (ks - 1) // 2
    rnd = (lambda t: t.bfloat16().float()) if dt == 'bf16' else (lambda t: t.half().float())
    g = torch.Generator().manual_seed(5)
    x = rnd(torch.randn(N, Ci, H, H, generator=g))
    OH = (H + 2 * pad - ks) // st + 1
    dy = rnd(torch.randn(N, Co, OH, OH, generator=g))
    wref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, ks, ks), dy.double(), st, pad).float()
    xv, dyv = View.from_nchw(x.to(DEV), code), View.from_nchw(dy.to(DEV), code)
    part = torch.full((split, dyv.C, ks * ks * Ci), float('nan'), device=DEV)
    ops.conv_wgrad(dyv, xv, ks, st, pad, part, split)
    gw = torch.zeros(Co, Ci, ks, ks, device=DEV)
    ops.wgrad_reduce(part, split, dyv.C, Ci, ks, Co, Ci, 1.0, False, gw)
    scale = float(wref.abs().max())
    assert float((gw.cpu() - wref).abs().max()) <= (3e-2 if dt == 'bf16' else 4e-3) * scale
    # the same call in a process where the library never takes the eight-wave tile (CY_WGRAD_LOADERS is read once per process)
    out = _worker('wgrad_case', tmp_path, 'w', dt, *case, env_extra={'CY_WGRAD_LOADERS': '2'})
    other = torch.tensor(out['grad']).reshape(Co, Ci, ks, ks)
    assert float((gw.cpu() - other).abs().max()) <= 2e-5 * scale * (N * OH * OH / 64) ** 0.5


# (dY channels = conv Cout, dX channels = conv Cin, dX height, dX width, batch): the direct kernel's shape (64 -> 32) with whole
# and partial tiles (8 x 64 dX pixels), and a shape only the implicit-GEMM kernels take (one merged launch)
S2_DGRAD = [(64, 32, 80, 128, 2), (64, 32, 20, 96, 3), (64, 32, 304, 304, 1), (128, 64, 40, 48, 2)]


@pytest.mark.parametrize('dt', ['f16', 'bf16'])
@pytest.mark.parametrize('case', S2_DGRAD)
def test_stride2_dgrad_one_launch_and_direct_kernel(dt, case):
    """3x3 / stride-2 input gradient: the four parity classes as one launch of the implicit-GEMM kernels (hint 1: 4-wave,
    6: pipelined) and, for 32 <- 64 channels, the direct kernel (hint 10) -- against float64 torch; gradient fan-in; and the
    BatchNorm-backward sums of the producer layer in the epilogue (cy_conv_dgrad_bn_sums with stride 2) against the separate
    reduce pass over the stored gradient."""
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.ops import View
    code = ops.dtype_code(dt)
    Cdy, Cg, H, W, N = case
    rnd = (lambda t: t.bfloat16().float()) if dt == 'bf16' else (lambda t: t.half().float())
    g = torch.Generator().manual_seed(Cdy + H)
    dy = rnd(torch.randn(N, Cdy, H // 2, W // 2, generator=g))
    w = rnd(torch.randn(Cdy, Cg, 3, 3, generator=g) / (9 * Cdy) ** 0.5)
    gref = torch.nn.grad.conv2d_input((N, Cg, H, W), w.double(), dy.double(), 2, 1).float()
    dyv = View.from_nchw(dy.to(DEV), code)
    _, wd = ops.pack_weights(w.to(DEV), Cdy, Cg, code)
    tol = dict(rtol=1.6e-2, atol=1.6e-2) if dt == 'bf16' else dict(rtol=2e-3, atol=2e-3)
    hints = [1, 6] + ([10] if (Cdy, Cg) == (64, 32) else [])
    n0 = ops.direct_launches()
    outs = {}
    for h in hints:
        dx = View.alloc(N, H, W, Cg, code, ld=Cg + 8, zero=True)
        ops.conv_igemm(dyv, wd, Cg, dx, 3, 2, 1, flags=ops.CONV_TRANSPOSED, tile=h)
        torch.testing.assert_close(dx.to_nchw().cpu(), gref, **tol)
        assert float(dx.buf.view(-1, Cg + 8)[:, Cg:].abs().max()) == 0.0          # nothing written beside the view
        ops.conv_igemm(dyv, wd, Cg, dx, 3, 2, 1, flags=ops.CONV_TRANSPOSED | ops.CONV_ACCUM, tile=h)
        torch.testing.assert_close(dx.to_nchw().cpu(), 2 * gref, rtol=2 * tol['rtol'], atol=2 * tol['atol'])
        outs[h] = dx
    if 10 in hints:
        assert ops.direct_launches() == n0 + 2
    # BN-backward sums of the layer whose output gradient this is
    raw = View.alloc(N, H, W, Cg, code)
    raw.buf.copy_(torch.randn(raw.buf.numel(), generator=g).to(raw.buf.dtype))
    vec = torch.stack([torch.randn(Cg, generator=g) * 0.1, torch.rand(Cg, generator=g) + 0.5,
                       torch.rand(Cg, generator=g) + 0.5, torch.randn(Cg, generator=g) * 0.2]).to(DEV)
    M = N * H * W
    for act in ('mish', 'leaky'):
        a = ops.ACT[act]
        for h in [x for x in hints if x != 1]:
            g1 = View.alloc(N, H, W, Cg, code, zero=True)
            ops.conv_igemm(dyv, wd, Cg, g1, 3, 2, 1, flags=ops.CONV_TRANSPOSED, tile=h)
            part = torch.zeros(ops.bn_bwd_rows(M, Cg, code), 2, Cg, device=DEV)
            ops.bn_act_bwd_reduce(raw, g1, vec[0], vec[1], vec[2], vec[3], a, part)
            ref_sums = part.double().sum(0)
            g2 = View.alloc(N, H, W, Cg, code, zero=True)
            tbl = torch.zeros(ops.conv_stats_rows(M, Cg), 2, Cg, device=DEV)
            ops.conv_dgrad_bn_sums(dyv, wd, Cg, g2, 3, 2, 1, raw, vec[0], vec[1], vec[2], vec[3], a, tbl,
                                   flags=ops.CONV_TRANSPOSED, tile=h)
            assert torch.equal(g2.buf, g1.buf), h
            got = tbl.double().sum(0)
            scale = ref_sums.abs().max(1, keepdim=True).values + 1e-6
            assert float(((got - ref_sums).abs() / scale).max()) < 2e-5, (act, h)


@pytest.mark.parametrize('dt', ['f16', 'bf16'])
@pytest.mark.parametrize('accum', [False, True])
@pytest.mark.parametrize('case', [(64, 64, 304, 1), (64, 128, 152, 2), (128, 64, 152, 1), (32, 64, 200, 1), (64, 32, 150, 2)])
def test_direct1x1_dgrad_bn_sums_and_prefetched_fan_in(dt, accum, case):
    """The 1x1 streaming kernel as an input-gradient launch (hint 10): gradient fan-in with the stored gradient prefetched, and
    the BatchNorm-backward sums of the producer layer in its epilogue, against the 4-wave kernel followed by the reduce pass."""
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.ops import View
    code = ops.dtype_code(dt)
    Cdy, Cg, H, N = case
    g = torch.Generator().manual_seed(Cdy * 5 + Cg + H)
    tdt = ops.torch_dtype(code)
    dy = View.alloc(N, H, H, Cdy, code); dy.buf.copy_(torch.randn(dy.buf.numel(), generator=g).to(tdt))
    raw = View.alloc(N, H, H, Cg, code); raw.buf.copy_(torch.randn(raw.buf.numel(), generator=g).to(tdt))
    w = torch.randn(Cdy, Cg, 1, 1, generator=g).to(DEV) * (1.0 / Cdy ** 0.5)
    _, wd = ops.pack_weights(w, Cdy, Cg, code)
    vec = torch.stack([torch.randn(Cg, generator=g) * 0.1, torch.rand(Cg, generator=g) + 0.5,
                       torch.rand(Cg, generator=g) + 0.5, torch.randn(Cg, generator=g) * 0.2]).to(DEV)
    base = torch.randn(raw.buf.numel(), generator=g).to(DEV).to(tdt)
    flags = ops.CONV_TRANSPOSED | (ops.CONV_ACCUM if accum else 0)
    M = N * H * H
    for act in ('mish', 'leaky', 'linear'):
        a = ops.ACT[act]
        g1 = View.alloc(N, H, H, Cg, code); g1.buf.copy_(base)
        ops.conv_igemm(dy, wd, Cg, g1, 1, 1, 0, flags=flags, tile=1)
        g2 = View.alloc(N, H, H, Cg, code); g2.buf.copy_(base)
        n0 = ops.direct_launches()
        tbl = torch.zeros(ops.conv_stats_rows(M, Cg), 2, Cg, device=DEV)
        ops.conv_dgrad_bn_sums(dy, wd, Cg, g2, 1, 1, 0, raw, vec[0], vec[1], vec[2], vec[3], a, tbl, flags=flags, tile=10)
        assert ops.direct_launches() == n0 + 1
        tol = 2e-2 if dt == 'bf16' else 3e-3
        torch.testing.assert_close(g2.buf.float(), g1.buf.float(), rtol=tol, atol=tol)
        part = torch.zeros(ops.bn_bwd_rows(M, Cg, code), 2, Cg, device=DEV)
        ops.bn_act_bwd_reduce(raw, g2, vec[0], vec[1], vec[2], vec[3], a, part)      # over the gradient the fused launch stored
        ref_sums, got = part.double().sum(0), tbl.double().sum(0)
        scale = ref_sums.abs().max(1, keepdim=True).values + 1e-6
        assert float(((got - ref_sums).abs() / scale).max()) < 2e-5, act
        g3 = View.alloc(N, H, H, Cg, code); g3.buf.copy_(base)
        ops.conv_igemm(dy, wd, Cg, g3, 1, 1, 0, flags=flags, tile=10)                # the same kernel without the sums
        assert torch.equal(g3.buf, g2.buf)


@pytest.mark.parametrize('warmup', [0, 1])
def test_graphed_train_step_equals_eager_steps(warmup):
    """graphed.GraphedTrainStep (forward + loss + backward + Adam as ONE captured hipGraph, replayed) against the same steps
    issued eagerly, deterministic mode: parameters, BatchNorm running statistics and losses after three different batches are
    bit-identical; a learning-rate change between replays is honoured without re-capture; a new target count captures a
    second graph (warmup=0) or runs its first batch eagerly (warmup=1, the default: warm-up batches are ordinary steps, no
    batch is ever stepped twice)."""
    import copy
    from complex_yolov4_pytorch_amd.graphed import GraphedTrainStep
    from complex_yolov4_pytorch_amd.optim import FusedAdam
    batches = [(syn.bev_images(2, 416, seed=40 + i).to(DEV), syn.targets(2, 6, 416, seed=40 + i).to(DEV)) for i in range(3)]
    extra = (syn.bev_images(2, 416, seed=50).to(DEV), syn.targets(2, 5, 416, seed=50).to(DEV))       # 10 target rows instead of 12
    runs = []
    for graphed in (False, True):
        model = _model('complex_yolov4.cfg', 'f16', deterministic=True)
        model.train()
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-4, capturable=True)

        def eager(x, tg, model=model, opt=opt):
            opt.zero_grad(set_to_none=True)
            loss, _ = model(x, tg)
            loss.backward()
            opt.step()
            return loss
        eager(*batches[0])                                        # both arms: one eager step first (tuning, state, workspaces)
        step = GraphedTrainStep(model, opt, warmup=warmup) if graphed else eager
        losses = []
        for i, (x, tg) in enumerate(batches + [extra]):
            if i == 2:
                for g in opt.param_groups:
                    g['lr'] = 3e-4                                 # an LR schedule moving on between steps
            losses.append(float(step(x, tg).detach()))
        torch.cuda.synchronize()
        if graphed:
            assert (step.replays, len(step._graphs)) == ((4, 2) if warmup == 0 else (2, 1))
        runs.append((losses, copy.deepcopy({k: v.detach().clone() for k, v in model.state_dict().items()})))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for k, v in runs[0][1].items():
        assert torch.equal(v, runs[1][1][k]), k


@pytest.mark.parametrize('giou', [True, False])
def test_batched_heads_equal_per_head_calls(giou):
    """cy_yolo_loss_multi (decode + loss of the three heads, head = blockIdx.y) against cy_yolo_decode + cy_yolo_loss per head on
    the same logits and targets (collisions included): outputs, the 20 metrics and d(logits) of every head bit-identical."""
    import complex_yolov4_pytorch_amd.ops as ops
    B, A, C, S = 4, 3, 3, 608
    anchors = [[(11, 14, -3.14, 1), (11, 14, 0, 1), (23, 51, 0.3, 0.9)], [(23, 51, -3.14, 1), (23, 51, 0, 1), (24, 60, 1, 0)],
               [(27, 63, 0, 1), (29, 74, 0.2, 0.95), (29, 74, -1, 0.1)]]
    Gs = [76, 38, 19]
    g = torch.Generator().manual_seed(9)
    logits = [(torch.randn(B * G * G * A * (7 + C), generator=g) * 0.6).to(DEV) for G in Gs]
    tg = syn.targets(B, 7, S, seed=9, collide=True).to(DEV)
    bad = tg[3:4].clone(); bad[0, 0] = -1                       # a rejected row (sample index out of range)
    tg = torch.cat([tg, bad], 0).contiguous()
    nT = tg.shape[0]
    rows_total = A * sum(G * G for G in Gs)
    offs = [0, A * Gs[0] ** 2, A * (Gs[0] ** 2 + Gs[1] ** 2)]
    # per head
    out1 = torch.zeros(B, rows_total, 7 + C, device=DEV)
    met1 = [torch.zeros(20, device=DEV) for _ in Gs]
    dl1 = [torch.empty_like(l) for l in logits]
    for h, G in enumerate(Gs):
        ops.yolo_decode(logits[h], B, G, A, C, anchors[h], S, out1, rows_total, offs[h])
        ws = torch.empty(ops.yolo_loss_workspace(B, G, A, C, nT), dtype=torch.uint8, device=DEV)
        ops.yolo_loss(logits[h], B, G, A, C, tg, anchors[h], S, 0.7, giou, ws, met1[h], dl1[h])
    # batched
    out2 = torch.zeros(B, rows_total, 7 + C, device=DEV)
    met2 = [torch.zeros(20, device=DEV) for _ in Gs]
    dl2 = [torch.empty_like(l) for l in logits]
    table = ops.make_head_table([(logits[h], dl2[h], met2[h], anchors[h], Gs[h], offs[h]) for h in range(3)])
    ws = torch.empty(ops.yolo_loss_multi_workspace(Gs, B, A, C, nT + 5), dtype=torch.uint8, device=DEV)
    for _ in range(2):                                          # twice: the workspace is re-zeroed by the call itself
        ops.yolo_loss_multi(table, 3, B, A, C, tg, S, 0.7, giou, ws, out2, rows_total)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)
    for h in range(3):
        assert torch.equal(met1[h], met2[h]), (h, met1[h], met2[h])
        assert torch.equal(dl1[h], dl2[h]), h
    assert float(met1[0][19]) == 1.0                            # the rejected row is counted, not assigned
    # ... and with the row count ON THE DEVICE (cy_yolo_loss_multi_n, round 5): launches sized for a 64-row bucket, the buffer's
    # rows beyond the batch's count hold GARBAGE (stale rows of an earlier batch, a NaN row, an out-of-range sample index) that
    # must never be looked at: everything bit-identical again -- for this batch, for a shorter one and for an empty one
    for cap, n_live in ((64, nT), (128, nT), (64, 5), (64, 0)):
        buf = torch.full((cap + 8, 8), float('nan'), device=DEV)
        buf[:nT] = tg
        buf[nT:nT + 6] = tg[:6] * 0.5 + 0.1                      # plausible-looking stale rows
        buf[nT + 6, 0] = 99.0
        nt_dev = torch.tensor([n_live], dtype=torch.int32, device=DEV)
        out3 = torch.zeros_like(out2)
        met3 = [torch.zeros(20, device=DEV) for _ in Gs]
        dl3 = [torch.empty_like(l) for l in logits]
        table3 = ops.make_head_table([(logits[h], dl3[h], met3[h], anchors[h], Gs[h], offs[h]) for h in range(3)])
        ws3 = torch.empty(ops.yolo_loss_multi_workspace(Gs, B, A, C, cap), dtype=torch.uint8, device=DEV)
        ops.yolo_loss_multi(table3, 3, B, A, C, buf, S, 0.7, giou, ws3, out3, rows_total, cap=cap, nt_dev=nt_dev)
        # the reference for a shorter batch: the exact-count entry point on the first n_live rows
        out4 = torch.zeros_like(out2)
        met4 = [torch.zeros(20, device=DEV) for _ in Gs]
        dl4 = [torch.empty_like(l) for l in logits]
        table4 = ops.make_head_table([(logits[h], dl4[h], met4[h], anchors[h], Gs[h], offs[h]) for h in range(3)])
        ops.yolo_loss_multi(table4, 3, B, A, C, tg[:n_live].contiguous() if n_live else None, S, 0.7, giou, ws, out4, rows_total)
        torch.cuda.synchronize()
        same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))      # noqa: E731
        assert torch.equal(out3, out4), (cap, n_live)
        for h in range(3):
            # (an EMPTY batch has 0 / 0 means -- NaN metrics, as in the reference's empty-target step -- in both forms alike)
            assert same(met3[h], met4[h]), (cap, n_live, h, met3[h], met4[h])
            assert same(dl3[h], dl4[h]), (cap, n_live, h)
            if n_live:
                assert torch.isfinite(dl3[h]).all() and torch.isfinite(met3[h]).all()
