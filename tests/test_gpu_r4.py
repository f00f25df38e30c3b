"""Round-4 GPU evidence (VERDICT r3 "next round" #1, #2, #7 and ADVICE r3):

* (the training-dynamics comparisons of the 16-bit modes -- conditioned net, ideal-storage control, convergence A/B -- live in
  tests/test_zz_gpu_dynamics.py, which collects LAST: a chaotic-trajectory test must never again stop `-x` before the kernel,
  replay and configs[3] tests below, as it did in GPUTEST_r04)
* BASELINE configs[3] (batch 32, 608x608 inference + rotated NMS) against THE REFERENCE's eval forward + post_processing_v2
  (tests/golden/darknet_eval.npz), f32 eval path and the f16 fused-eval path with static_eval_weights;
* a canary for the shipped stream configuration of the heads;
* a captured step survives a head-workspace growth; FusedAdam(capturable) checkpoints carry the true step count;
* bench.py --gpus 1 through the data-parallel wrapper equals the plain line.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from tests.test_gpu_r2 import DEV, _model  # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
B, S = 16, 608


# ---- BASELINE configs[3]: inference batch 32 at 608x608 + rotated NMS against the reference ---------------------------------
def _eval_model(g, dtype):
    model = _model('complex_yolov4.cfg', dtype)
    sd = model.state_dict()
    off = 0
    for name, n in zip(g['bn_names'], g['bn_sizes']):
        sd[str(name)] = torch.from_numpy(g['bn_values'][off:off + int(n)].copy())
        off += int(n)
    model.load_state_dict(sd)
    model.eval()
    model.cpu_outputs = False
    return model


def _match(det, ref, tol):
    """Greedy one-to-one matching of two detection lists of one image (rows x, y, w, l, im, re, obj, cls score, cls): a pair
    matches when the classes are equal and the boxes agree within tol.  -> number of matched pairs."""
    if det is None or len(det) == 0 or len(ref) == 0:
        return 0
    used, n = np.zeros(len(det), dtype=bool), 0
    for r in ref:
        d = (np.abs(det[:, :6] - r[:6]) / (np.abs(r[:6]) + 1.0)).max(1)
        d[used | (det[:, 8] != r[8])] = np.inf
        j = int(np.argmin(d))
        if d[j] <= tol:
            used[j] = True
            n += 1
    return n


EVAL = {  # decoded rows vs the reference: probabilities (abs), im / re (abs), boxes (relative to the coordinate, + 0.02 px); `borderline`:
    # how far from the confidence threshold a row may be whose side of it differs; `found` / `tol`: share of the reference's
    # detections found end to end with the same class and a box within tol (relative)
    'f32': dict(prob=1e-3, imre=4e-3, box=3e-3, borderline=2e-3, found=0.99, tol=3e-3),
    # f16: the fused eval kernels.  The bounds are NOT kernel tolerances: the reference's own float32 arithmetic with ideal f16
    # storage (oracle storage_round) moves these outputs by median 1.7e-2 / max 0.28 on this random-init (calibrated) net;
    # the test requires the device's error statistics to equal that (factor 1.5), see below
    'f16': dict(prob=0.45, imre=2.5, box=None, borderline=None, found=None, tol=None),
}


@pytest.mark.parametrize('dtype', ['f32', 'f16'])
def test_inference_b32_608_against_reference(golden, dtype):
    """The reference's model.eval()(imgs) + post_processing_v2 on the seeded batch of 32 (evaluate.py:32-45), BatchNorm
    running statistics calibrated by the reference (make_golden_eval.py).  f32: the parity eval path; f16: the benchmarked
    fused conv+BN+act eval kernels with static_eval_weights.  (1) decoded rows against the golden sample and the golden's
    candidate rows; (2) the rows passing the confidence threshold: any disagreement with the reference must be a row within
    the stated band of the threshold; (3) rotated merge-NMS on the device over the REFERENCE's candidate rows: per-image counts
    and classes exact, boxes 1e-4; (4) end to end, the device's own outputs through the device NMS: the reference's
    detections found (same class, box within tol) and no more spurious ones than borderline rows allow."""
    from complex_yolov4_pytorch_amd.utils.evaluation_utils import post_processing_v2
    g = golden('darknet_eval')
    band = EVAL[dtype]
    model = _eval_model(g, dtype)
    if dtype != 'f32':
        model.static_eval_weights = True
    x = syn.bev_images(32, 608, seed=33).to(DEV)
    with torch.no_grad():
        out = model(x)
        if dtype != 'f32':
            out2 = model(x)                               # second batch on the cached weight pack: the benchmarked state
            assert torch.equal(out, out2)
    assert tuple(out.shape) == tuple(g['out_shape'])
    got, ref = out[:, ::97].cpu().numpy(), g['out_rows']
    dprob, dimre, dbox = np.abs(got[..., 6:] - ref[..., 6:]), np.abs(got[..., 4:6] - ref[..., 4:6]), np.abs(got[..., :4] - ref[..., :4])
    flat = out.reshape(-1, out.shape[-1])
    cand_idx = torch.from_numpy(g['cand_idx']).to(DEV)
    dcand = (flat[cand_idx].cpu().numpy() - g['cand_rows'])
    print('%s eval B32 608 vs reference: probabilities |d| median %.2e max %.2e, im/re max %.2e, boxes max %.2e px; candidate rows: '
          'probabilities max %.2e boxes max %.2e px' % (dtype, float(np.median(dprob)), float(dprob.max()), float(dimre.max()), float(dbox.max()),
                                                        float(np.abs(dcand[:, 6:]).max()), float(np.abs(dcand[:, :4]).max())))
    assert dprob.max() <= band['prob'] and dimre.max() <= band['imre']
    assert np.abs(dcand[:, 6:]).max() <= band['prob']
    if dtype == 'f32':
        assert np.all(dbox <= band['box'] * np.abs(ref[..., :4]) + 2e-2)
        assert np.all(np.abs(dcand[:, :4]) <= band['box'] * np.abs(g['cand_rows'][:, :4]) + 2e-2)
    else:
        # what ideal f16 storage does to the REFERENCE's arithmetic on the first two images (eval-mode BatchNorm is per sample)
        from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
        from oracle import darknet_ref
        from tests.util import storage_round
        net = darknet_ref.DarknetRef(parse_cfg(os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')))
        ps, bs = net.param_shapes()
        params = syn.fill_state_dict(ps)
        bufs, off = {}, 0
        for name, n in zip(g['bn_names'], g['bn_sizes']):
            bufs[str(name)] = torch.from_numpy(g['bn_values'][off:off + int(n)].copy())
            off += int(n)
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        with torch.no_grad():
            o32, _, _ = net.forward(params, x[:2].cpu(), None, True, False, bufs)
            o16, _, _ = net.forward(params, x[:2].cpu(), None, True, False, bufs, storage_round=storage_round(torch.float16))
        ideal = (o16[..., 6:] - o32[..., 6:]).abs().numpy()
        mine = (out[:2, :, 6:].cpu() - o32[..., 6:]).abs().numpy()
        stats = lambda a: (float(np.median(a)), float(np.percentile(a, 99)), float(a.max()))
        print('  images 0-1, probabilities: device f16 vs reference arithmetic |d| median %.2e p99 %.2e max %.2e;  ideal f16 storage in the '
              'reference arithmetic: median %.2e p99 %.2e max %.2e' % (stats(mine) + stats(ideal)))
        for a, b in zip(stats(mine)[:2], stats(ideal)[:2]):
            assert a <= 1.5 * b, (stats(mine), stats(ideal))
        # ... which is why the candidate set of a confidence threshold in the tail is not comparable in f16 on this net: ideal
        # storage changes it as much as the device path does
        thr0 = float(g['conf_thresh'][0])
        c32 = o32[..., 6] >= thr0
        d_ideal = int(((o16[..., 6] >= thr0) ^ c32).sum())
        d_mine = int(((out[:2, :, 6].cpu() >= thr0) ^ c32).sum())
        print('  images 0-1: rows on the other side of the confidence threshold than in float32: device f16 %d, ideal f16 storage %d (of %d candidates)'
              % (d_mine, d_ideal, int(c32.sum())))
        assert d_mine <= 1.5 * d_ideal + 8          # (measured: 81 vs 68 of 140)
        # END TO END on those two images (VERDICT r4 next #2d): the float32 oracle's detections (its outputs through the device
        # NMS) are the reference; how many of them survive ideal f16 storage is the control, and the f16 device path must find
        # as many, with no more spurious ones -- bound = control x margin, stated below
        nms_t = float(g['nms_thresh'][0])

        def e2e(rows):
            dets = post_processing_v2(rows.to(DEV).float().contiguous(), conf_thresh=thr0, nms_thresh=nms_t)
            return [None if d is None else d.numpy() for d in dets]
        ref32, ideal16, dev16 = e2e(o32), e2e(o16), e2e(out[:2])
        tot = sum(0 if r is None else len(r) for r in ref32)
        f_i = sum(_match(d, r, 0.05) for d, r in zip(ideal16, ref32) if r is not None)
        f_d = sum(_match(d, r, 0.05) for d, r in zip(dev16, ref32) if r is not None)
        x_i = sum(0 if d is None else len(d) for d in ideal16) - f_i
        x_d = sum(0 if d is None else len(d) for d in dev16) - f_d
        print('  images 0-1 end to end vs the float32 oracle\'s %d detections: found %d (device f16) / %d (ideal f16 storage); unmatched '
              '%d (device) / %d (ideal)' % (tot, f_d, f_i, x_d, x_i))
        # (printed, not asserted: on this random-init net even IDEAL f16 storage keeps 2 of the float32 oracle's 140 detections --
        # measured round 5 -- so no end-to-end statement about f16 can be made here.  The end-to-end f16-vs-f32 detection check that
        # CAN fail runs on the conditioned net: tests/test_zz_gpu_dynamics.py::test_f16_inference_detections_match_fp32_on_the_conditioned_net)
    # (2) threshold crossings
    thr = float(g['conf_thresh'][0])
    mine = set(torch.nonzero(flat[:, 6] >= thr).reshape(-1).cpu().tolist())
    theirs = set(g['cand_idx'].tolist())
    diff = mine ^ theirs
    near = dict(zip(g['near_idx'].tolist(), g['near_obj'].tolist()))
    worst = max([abs(near[i] - thr) if i in near else 1.0 for i in diff], default=0.0)
    print('  rows >= %.6f: device %d, reference %d, symmetric difference %d (farthest from the threshold: %.2e)' % (thr, len(mine), len(theirs), len(diff), worst))
    if band['borderline'] is not None:
        assert worst <= band['borderline']
    # (3) the device NMS on the reference's own candidate rows
    N, W = out.shape[1], out.shape[2]
    sparse = torch.zeros(32 * N, W, device=DEV)
    sparse[cand_idx] = torch.from_numpy(g['cand_rows']).to(DEV)
    dets = post_processing_v2(sparse.view(32, N, W), conf_thresh=thr, nms_thresh=float(g['nms_thresh'][0]))
    counts = np.asarray([0 if d is None else d.shape[0] for d in dets])
    np.testing.assert_array_equal(counts, g['det_count'])
    allrows = np.concatenate([d.numpy() for d in dets if d is not None], 0)
    np.testing.assert_array_equal(allrows[:, 8], g['det'][:, 8])
    np.testing.assert_allclose(allrows[:, :8], g['det'][:, :8], rtol=1e-4, atol=1e-4)
    # (4) end to end
    dets = post_processing_v2(out, conf_thresh=thr, nms_thresh=float(g['nms_thresh'][0]))
    found = total = extra = same_count = 0
    off = 0
    for b in range(32):
        n = int(g['det_count'][b])
        ref_b = g['det'][off:off + n]
        off += n
        d = None if dets[b] is None else dets[b].numpy()
        m = _match(d, ref_b, band['tol'] if band['tol'] is not None else 0.05)
        found += m
        total += n
        extra += (0 if d is None else len(d)) - m
        same_count += int((0 if d is None else len(d)) == n)
    print('  end to end: %d of %d reference detections found (class equal, box within %.0e relative), %d unmatched device detections, %d of 32 '
          'images with the same count' % (found, total, band['tol'] if band['tol'] is not None else 0.05, extra, same_count))
    if band['found'] is not None:
        assert found >= band['found'] * total
        assert extra <= (1 - band['found']) * total + len(diff)


# ---- canary: the shipped stream configuration of the heads ---------------------------------------------------------------------
def test_head_canary_default_step_is_bit_reproducible():
    """the GIoU kernels return wrong IoUs (lanes 48-63) when they run beside two of our conv instantiations
    on another stream (profiles/r03_head_race.txt; mechanism unknown), so the heads run on the trunk's stream.  This guards
    that SHIPPED configuration: 300 repeats of the deterministic v4 step at the benchmarked shape, every repeat's loss, outputs
    and flat gradient bit-identical to the first -- with the weight-gradient side stream on, as benchmarked.  A change that lets
    a head kernel overlap a conv again shows up here as a differing repeat within a few hundred steps (96 of 3999 in the probe)."""
    assert os.environ.get('CY_HEADS_SIDE', '0') != '1'
    model = _model('complex_yolov4.cfg', 'f16', deterministic=True)
    model.train()
    x, tg = syn.bev_images(B, S, seed=21).to(DEV), syn.targets(B, 6, S, seed=21, collide=True).to(DEV)
    eng = None
    ref, bad = None, torch.zeros(3, device=DEV)
    for i in range(300):
        model.zero_grad(set_to_none=True)
        loss, out = model(x, tg)
        loss.backward()
        cur = (loss.detach().clone(), out.detach(), model.flat_grad)
        if ref is None:
            ref = tuple(t.clone() for t in cur)
            eng = next(iter(model._engines.values()))
            assert eng.side is not None and not eng._heads_on_side
        else:
            bad += torch.stack([(c != r).any().float() for c, r in zip(cur, ref)])
    bad = bad.cpu().tolist()
    assert bad == [0.0, 0.0, 0.0], 'repeats differing in (loss, outputs, gradient): %s of 299' % bad
    assert float(ref[2].abs().max()) > 0 and np.isfinite(float(ref[0]))


# ---- ADVICE r3 --------------------------------------------------------------------------------------------------------------
def test_graph_replay_survives_head_workspace_growth_and_checkpoints_the_step_count():
    """A graph captured while the batched-heads workspace was sized for <= 64 target rows keeps that pointer in its kernel
    arguments; a later eager batch with more rows re-allocates the workspace.  The old one must stay alive (replays of the
    first graph write there).  Also: after replays the optimizer's checkpoint carries the true step count, and a resumed
    optimizer continues with the same bias correction."""
    import copy
    from complex_yolov4_pytorch_amd.graphed import GraphedTrainStep
    from complex_yolov4_pytorch_amd.optim import FusedAdam
    small = [(syn.bev_images(2, 416, seed=90 + i).to(DEV), syn.targets(2, 6, 416, seed=90 + i).to(DEV)) for i in range(4)]      # 12 rows
    big = (syn.bev_images(2, 416, seed=95).to(DEV), syn.targets(2, 40, 416, seed=95).to(DEV))                                  # 80 rows > 64
    runs = []
    for graphed in (False, True):
        model = _model('complex_yolov4.cfg', 'f16', deterministic=True)
        model.train()
        opt = FusedAdam(model.parameters(), lr=1e-3, capturable=True)

        def eager(x, tg, model=model, opt=opt):
            opt.zero_grad(set_to_none=True)
            loss, _ = model(x, tg)
            loss.backward()
            opt.step()
            return loss
        step = GraphedTrainStep(model, opt, warmup=1) if graphed else eager
        losses = [float(step(*small[0]).detach()), float(step(*small[1]).detach())]       # eager warm-up, then capture + replay
        eng = next(iter(model._engines.values()))
        ws_before = eng._head_table[1].data_ptr()
        losses.append(float(step(*big).detach()))                                          # grows the workspace (eager: new shape)
        assert eng._head_table[1].data_ptr() != ws_before and len(eng._retired_ws) == 1
        junk = [torch.full((eng._retired_ws[0][1].numel() // 4,), float('nan'), device=DEV) for _ in range(8)]   # would land on freed memory
        losses += [float(step(*small[2]).detach()), float(step(*small[3]).detach())]       # replays of the FIRST graph
        del junk
        torch.cuda.synchronize()
        if graphed:
            assert step.replays == 3 and len(step._graphs) == 1
        sd = opt.state_dict()
        steps = {int(v['step']) for v in sd['state'].values()}
        assert steps == {5}, steps                       # 5 optimizer steps happened, however they were issued
        assert opt._steps == 5 and int(opt._counter) == 5
        resumed = FusedAdam(model.parameters(), lr=1e-3, capturable=True)
        resumed.load_state_dict(copy.deepcopy(sd))
        assert resumed._steps == 5
        runs.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for k, v in runs[0][1].items():
        assert torch.equal(v, runs[1][1][k]), k


# ---- two-phase conv + BatchNorm (batch statistics) + activation -----------------------------------------------------------------
FUSED_CASES = [  # N, Cin, H, Cout, ks, stride, residual, act, tile hint
    (2, 64, 19, 128, 3, 1, False, 'mish', 2), (2, 64, 19, 128, 3, 1, True, 'leaky', 7), (1, 128, 38, 64, 1, 1, False, 'linear', 4),
    (2, 64, 38, 128, 3, 2, False, 'mish', 3), (16, 256, 38, 256, 3, 1, True, 'mish', 9), (16, 512, 19, 1024, 1, 1, False, 'leaky', 3),
    (4, 128, 76, 128, 3, 1, False, 'mish', 5), (16, 128, 76, 120, 1, 1, False, 'mish', 5), (4, 64, 38, 144, 1, 1, True, 'mish', 2),
]


@pytest.mark.parametrize('dt', ['f16', 'bf16'])
@pytest.mark.parametrize('case', FUSED_CASES)
def test_conv_bn_act_train_two_phase_equals_the_two_launch_path(dt, case):
    """cy_conv_bn_act_train (conv -> grid ticket -> BatchNorm with batch statistics + activation (+ shortcut) from the
    accumulators, VERDICT r3 missing #1) against cy_conv_igemm(CY_CONV_STATS) + cy_bn_act_fwd_fused on the same tile: the pre-BN
    tensor bit-identical; where every statistics bin receives one add (<= 16 pixel tiles) everything else bit-identical too,
    otherwise at the fp32 atomics' summation order; (mean, invstd, scale, shift), running statistics, num_batches_tracked, the
    zeroed other table, the ticket back at zero; a float64 torch reference on top."""
    import torch.nn.functional as F
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.ops import View
    N, Ci, H, Co, ks, st, with_res, actname, hint = case
    code, act = ops.dtype_code(dt), ops.ACT[actname]
    rnd = (lambda t: t.bfloat16().float()) if dt == 'bf16' else (lambda t: t.half().float())
    g = torch.Generator().manual_seed(77)
    pad = (ks - 1) // 2
    OH = (H + 2 * pad - ks) // st + 1
    x = rnd(torch.randn(N, Ci, H, H, generator=g))
    w = rnd(torch.randn(Co, Ci, ks, ks, generator=g) * (2.0 / (Ci * ks * ks)) ** 0.5)
    res = rnd(torch.randn(N, Co, OH, OH, generator=g)) if with_res else None
    gamma, beta = (1 + 0.1 * torch.randn(Co, generator=g)).to(DEV), (0.1 * torch.randn(Co, generator=g)).to(DEV)
    cop = (Co + 31) // 32 * 32
    xv = View.from_nchw(x.to(DEV), code)
    wf, _ = ops.pack_weights(w.to(DEV), cop, Ci, code)
    resv = View.from_nchw(res.to(DEV), code, ld=Co + 8) if with_res else None
    M = N * OH * OH
    rows = ops.conv_stats_rows(M, Co)

    def run(fused):
        raw = View.alloc(N, OH, OH, Co, code, ld=Co + 16 if Co % 16 == 0 else Co)
        out = View.alloc(N, OH, OH, Co, code, ld=Co + 24 if Co % 8 == 0 else Co)
        raw.buf.fill_(7.0); out.buf.fill_(7.0)
        bins, other = torch.zeros(rows * 2 * Co, device=DEV), torch.full((rows * 2 * Co,), 3.0, device=DEV)
        vec = torch.zeros(4, Co, device=DEV)
        rm, rv, nbt = torch.full((Co,), 0.25, device=DEV), torch.full((Co,), 2.0, device=DEV), torch.full((1,), 5, dtype=torch.int64, device=DEV)
        ticket = torch.zeros(4, dtype=torch.int32, device=DEV)
        if fused:
            took = ops.conv_bn_act_train(xv, wf, cop, raw, out, resv, ks, st, pad, bins, gamma, beta, rm, rv, nbt, 0.1, 1e-5, vec, other, act,
                                         ticket, tile=hint)
            if not took:
                return None
        else:
            ops.conv_igemm(xv, wf, cop, raw, ks, st, pad, flags=ops.CONV_STATS, stats=bins, tile=hint)
            ops.bn_act_fwd_fused(raw, out, resv, bins, rows, gamma, beta, rm, rv, nbt, 0.1, 1e-5, vec, other, act)
        torch.cuda.synchronize()
        return dict(raw=raw.to_nchw(), out=out.to_nchw(), vec=vec, rm=rm, rv=rv, nbt=int(nbt), other=other, ticket=ticket.cpu().tolist(),
                    rawpad=raw.buf.view(-1, raw.ld)[:, Co:], outpad=out.buf.view(-1, out.ld)[:, Co:])
    a, b = run(True), run(False)
    if N * OH * OH * ((Co + 127) // 128) > 256 * 384:            # (more than one round whatever the tile: must be refused)
        assert a is None
        return
    assert a is not None, 'the kernel refused a single-round launch'
    assert a['ticket'] == [0, 0, 0, 0] and a['nbt'] == b['nbt'] == 6
    assert float(a['other'].abs().max()) == 0.0 and float(b['other'].abs().max()) == 0.0
    assert torch.equal(a['raw'], b['raw'])
    for pad_t in (a['rawpad'], a['outpad']):                                       # channel padding untouched
        assert pad_t.numel() == 0 or float((pad_t.float() - 7.0).abs().max()) == 0.0
    cap = {2: 128, 7: 128, 3: 192, 8: 192, 4: 256, 9: 256, 5: 384}[hint]
    single_add = (M + cap - 1) // cap <= 16
    # float64 reference of the block (reference darknet2pytorch.py:247-278 in train mode)
    y = F.conv2d(x.double(), w.double(), None, st, pad)
    mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=False)
    z = (y - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * gamma.cpu().double().view(1, -1, 1, 1) + beta.cpu().double().view(1, -1, 1, 1)
    ref = {'mish': z * torch.tanh(F.softplus(z)), 'leaky': F.leaky_relu(z, 0.1), 'linear': z}[actname]
    if with_res:
        ref = ref + res.double()
    rtol = dict(rtol=3e-2, atol=3e-2) if dt == 'bf16' else dict(rtol=4e-3, atol=4e-3)
    for name, r in (('two-phase', a), ('two launches', b)):
        torch.testing.assert_close(r['out'].cpu().double(), ref, msg=lambda m, name=name: name + ' vs float64: ' + m, **rtol)
        torch.testing.assert_close(r['vec'][0].cpu().double(), mean, rtol=1e-4, atol=1e-5)
    unb = var * M / (M - 1)
    torch.testing.assert_close(a['rv'].cpu().double(), 0.9 * 2.0 + 0.1 * unb, rtol=1e-4, atol=1e-5)
    # two kernels, two compilations of the same fp32 formulas (fused multiply-adds contracted differently): last-bit differences
    # in (scale, shift) and one storage-type ulp in the output where every bin received a single add; the fp32 atomics'
    # summation order on top where a bin received several
    t_vec = dict(rtol=2e-6, atol=2e-7) if single_add else dict(rtol=2e-5, atol=2e-6)
    for k in ('vec', 'rm', 'rv'):
        torch.testing.assert_close(a[k], b[k], **t_vec)
    ulp = 2.0 ** -7 if dt == 'bf16' else 2.0 ** -10
    tol = dict(rtol=1.01 * ulp, atol=1e-3 * ulp) if single_add else dict(rtol=3 * ulp, atol=3 * ulp)
    torch.testing.assert_close(a['out'], b['out'], **tol)
    assert float((a['out'] != b['out']).float().mean()) < (0.02 if single_add else 0.2)


def test_conv_bn_act_train_refuses_a_grid_of_more_than_one_round():
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.ops import View
    code = ops.dtype_code('f16')
    N, C, H = 16, 64, 152                     # 369,664 pixels: 963 tiles of 384
    xv = View.alloc(N, H, H, C, code, zero=True)
    wf = torch.zeros(64, 9 * C, dtype=torch.float16, device=DEV)
    raw, out = View.alloc(N, H, H, 64, code), View.alloc(N, H, H, 64, code)
    bins, other = torch.zeros(16 * 2 * 64, device=DEV), torch.zeros(16 * 2 * 64, device=DEV)
    ones = torch.ones(64, device=DEV)
    ticket = torch.zeros(4, dtype=torch.int32, device=DEV)
    for hint in (2, 4, 5, 9):
        assert ops.conv_bn_act_train(xv, wf, 64, raw, out, None, 3, 1, 1, bins, ones, ones, None, None, None, 0.1, 1e-5, torch.zeros(4, 64, device=DEV),
                                     other, 2, ticket, tile=hint) is False
    torch.cuda.synchronize()
    assert ticket.cpu().tolist() == [0, 0, 0, 0] and float(bins.abs().max()) == 0.0          # nothing was launched


def test_model_forward_with_two_phase_convs_matches_the_separate_passes(monkeypatch):
    """The v4 train step with every eligible layer on the two-phase launch (CY_CONV_BN_FUSED=2) against the same step with none
    (=0), default mode f16, batch 16 at 608x608: loss, outputs and the flat gradient within the run-to-run band of the default
    mode (the fp32 atomics of the statistics bins), BatchNorm running statistics at 1e-5; and how many layers qualified."""
    res = {}
    x, tg = syn.bev_images(16, 608, seed=21).to(DEV), syn.targets(16, 6, 608, seed=21).to(DEV)
    for mode in ('2', '0', '0b'):
        monkeypatch.setenv('CY_CONV_BN_FUSED', mode[0])
        model = _model('complex_yolov4.cfg', 'f16')
        model.train()
        for _ in range(2):                                      # (the second step runs from the recorded launch list)
            model.zero_grad(set_to_none=True)
            loss, out = model(x, tg)
            loss.backward()
        torch.cuda.synchronize()
        eng = next(iter(model._engines.values()))
        eng.check_grid_waits()
        res[mode] = (float(loss.detach()), out.detach().clone(), model.flat_grad.detach().double().clone(),
                     {k: v.detach().clone() for k, v in model.state_dict().items() if 'running' in k}, len(eng._fwd_fused))
        model.release_engines()
        del model
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
    band = cos(res['0'][2], res['0b'][2])                        # two runs of the SAME configuration
    got = cos(res['2'][2], res['0'][2])
    print('two-phase conv+BN+act on %d of 107 BN layers: loss %.5f vs %.5f, gradient cosine vs separate passes %.5f (two separate-pass runs: %.5f), '
          'outputs max |d| %.2e (%.2e)' % (res['2'][4], res['2'][0], res['0'][0], got, band, float((res['2'][1] - res['0'][1]).abs().max()),
                                          float((res['0b'][1] - res['0'][1]).abs().max())))
    assert res['2'][4] >= 30 and res['0'][4] == 0
    # (two runs of the separate-pass configuration differ by 0.2-0.4 % in the loss at this random init; 5e-3 was met by luck)
    assert abs(res['2'][0] - res['0'][0]) <= 5e-3 * abs(res['0'][0]) + 2.0 * abs(res['0b'][0] - res['0'][0])
    assert got >= band - 0.15                                    # (both are draws of the same chaotic quantity at this random init)
    # running statistics: against the run-to-run spread of the separate-pass configuration itself (deep layers see inputs that
    # move with the atomics' order at this random init)
    ratios = []
    for k, v in res['0'][3].items():
        spread = float((res['0b'][3][k] - v).abs().max())
        d = float((res['2'][3][k] - v).abs().max())
        assert d <= 10 * spread + 1e-3 * float(v.abs().max()) + 1e-6, (k, d, spread)
        ratios.append(d / (spread + 1e-12 + 1e-7 * float(v.abs().max())))
    assert np.median(ratios) <= 3.0, np.median(ratios)


# ---- recorded launch lists (cy_run_plan) ----------------------------------------------------------------------------------------
def test_replayed_launch_lists_equal_eager_steps(monkeypatch):
    """the passes of a step re-issued from C (ops.start_recording -> cy_run_plan) against the same steps
    issued call by call from Python, deterministic mode: losses, parameters and BatchNorm running statistics bit-identical over
    nine Adam steps with three target-row counts in TWO 64-row buckets (the batched heads read the live count on the device,
    cy_yolo_loss_multi_n: a new bucket records a new list, every count of a bucket seen before replays it -- also after the head
    workspace has grown in between) and a learning-rate change."""
    from complex_yolov4_pytorch_amd.optim import FusedAdam
    S = 416
    mk = lambda seed, per: (syn.bev_images(2, S, seed=seed).to(DEV), syn.targets(2, per, S, seed=seed).to(DEV))
    seq = [mk(100, 6), mk(101, 6), mk(102, 6), mk(103, 5), mk(104, 6), mk(105, 40), mk(106, 6), mk(107, 5), mk(108, 40)]
    runs = {}
    for mode in ('replay', 'eager'):
        monkeypatch.setenv('CY_PLAN_REPLAY', '1' if mode == 'replay' else '0')
        model = _model('complex_yolov4.cfg', 'f16', deterministic=True)
        model.train()
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        losses = []
        for i, (x, tg) in enumerate(seq):
            if i == 4:
                for g in opt.param_groups:
                    g['lr'] = 3e-4
            opt.zero_grad(set_to_none=True)
            loss, out = model(x, tg)
            loss.backward()
            opt.step()
            losses.append((float(loss.detach()), float(out.abs().sum())))
        torch.cuda.synchronize()
        eng = next(iter(model._engines.values()))
        runs[mode] = (losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, eng.replayed,
                      (len(eng._fwd_progs), len(eng._bwd_progs)))
        model.release_engines()
        del opt, model
    assert runs['eager'][2] == 0 and runs['eager'][3] == (0, 0)
    # step 1 tunes, step 2 records (12 rows: bucket 64), 3 replays, 4 REPLAYS with 10 rows, 5 replays, 6 records (80 rows: bucket 128),
    # 7-9 replay
    assert runs['replay'][3] == (2, 2), runs['replay'][3]
    assert runs['replay'][2] == 2 * 6, runs['replay'][2]
    assert runs['replay'][0] == runs['eager'][0], (runs['replay'][0], runs['eager'][0])
    for k, v in runs['eager'][1].items():
        assert torch.equal(v, runs['replay'][1][k]), k


def test_replayed_inference_equals_eager(monkeypatch):
    """The eval engine with static_eval_weights: the forward after the weight pack is recorded, the following ones replay it;
    outputs bit-identical to the eager engine's, and a parameter change (mark_weights_dirty) re-packs instead of replaying a
    stale list."""
    outs = {}
    x = syn.bev_images(4, 608, seed=9).to(DEV)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'darknet_eval.npz'), allow_pickle=False)     # (calibrated running statistics: finite outputs)
    for mode in ('replay', 'eager'):
        monkeypatch.setenv('CY_PLAN_REPLAY', '1' if mode == 'replay' else '0')
        model = _eval_model(g, 'f16')
        model.static_eval_weights = True
        with torch.no_grad():
            a = [model(x).clone() for _ in range(4)]
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.mul_(1.01)
            model.mark_weights_dirty()
            b = [model(x).clone() for _ in range(3)]
        eng = next(iter(model._engines.values()))
        outs[mode] = (a, b, eng.replayed)
        model.release_engines()
        del model
    assert outs['eager'][2] == 0 and outs['replay'][2] >= 3
    assert bool(torch.isfinite(outs['eager'][0][0]).all())
    for i in range(4):
        assert torch.equal(outs['replay'][0][i], outs['eager'][0][0])
    for i in range(3):
        assert torch.equal(outs['replay'][1][i], outs['eager'][1][0])
    assert not torch.equal(outs['eager'][0][0], outs['eager'][1][0])


def test_replay_under_the_data_parallel_wrapper(tmp_path):
    from tests.test_gpu_r3 import _worker
    r = _worker('replay_ddp', tmp_path, 'replay_ddp', timeout=900)
    a, b = r['replay'], r['eager']
    assert b['replayed'] == 0 and a['replayed'] >= 8, (a['replayed'], a['programs'])
    assert a['losses'] == b['losses'] and a['params'] == b['params'] and a['bn'] == b['bn']
    assert a['per_step'] == b['per_step'] and all(len(c) >= 2 for c in a['per_step'])      # bucketed all-reduces every step


def test_two_data_parallel_ranks_with_real_kernels(tmp_path):
    """No 8-GPU node was ever available to the builder, and the nccl test above has ONE rank.  This one has TWO: both on this box's
    single GPU, exchanging over gloo (CUDA tensors staged through the host) -- real kernels, real hooks, a real second rank.
    After the first backward each rank's gradient must be the mean of the two ranks' LOCAL gradients (computed by plain models
    before the process group exists): the backward kernels emit gradient / 2, the SUM all-reduce adds them -- exact halvings and
    one rounding per element either way, so the comparison is bit for bit.  After three Adam steps both ranks hold identical
    parameters, and the later steps replayed their recorded launch lists around the all-reduce hooks."""
    import socket
    s_ = socket.socket(); s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]; s_.close()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'CY_TUNE_RECORD', 'CY_DDP_FORCE')}
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('GLOO_SOCKET_IFNAME', 'lo')          # (the box's hostname may not resolve: both ranks are local anyway)
    outs = [os.path.join(str(tmp_path), 'rank%d.json' % r) for r in (0, 1)]
    procs = [subprocess.Popen([sys.executable, '-m', 'tests.gpu_workers', 'two_ranks_one_gpu', outs[r], str(r)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), logs[0][-3000:] + logs[1][-3000:]
    res = [json.load(open(o)) for o in outs]
    if any('skip' in r for r in res):
        pytest.skip('gloo without CUDA-tensor support here: %s' % [r.get('skip') for r in res])
    t = [torch.load(o + '.pt') for o in outs]
    want = (t[0]['g_local'] + t[1]['g_local']) / 2
    for r in (0, 1):
        assert torch.equal(t[r]['g_ddp'], want), (r, float((t[r]['g_ddp'] - want).abs().max()), float(want.abs().max()))
    assert not torch.equal(t[0]['g_local'], t[1]['g_local'])          # (the ranks really had different batches)
    assert res[0]['params'] == res[1]['params']
    assert all(np.isfinite(res[0]['losses'])) and res[0]['losses'] != res[1]['losses']
    assert res[0]['replayed'] >= 2, res[0]
    print('two ranks on one GPU over gloo: gradient = mean of the local gradients bit for bit (max |g| %.3e), parameters equal after 3 steps, '
          '%d of %d passes replayed' % (float(want.abs().max()), res[0]['replayed'], res[0]['passes']))


# ---- bench.py --gpus N ---------------------------------------------------------------------------------------------------------
def _bench(extra_env, *args):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'CY_TUNE_RECORD')}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '12', '--warmup', '4', '--no-extra', '--no-cpu-baseline',
                        '--no-roofline'] + list(args), capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1])


def test_bench_through_the_data_parallel_path_equals_the_plain_line():
    """`python bench.py --gpus 1` plain, and with the RCCL wrapper forced on (CY_DDP_FORCE=1: bucketed all-reduces over nccl on
    the side stream, world size 1): same metric and configuration, throughput within 5 % (the wrapper's cost was 0.8 % in round
    3), the exposed part of the all-reduce reported."""
    plain = _bench({}, '--gpus', '1')
    ddp = _bench({'CY_DDP_FORCE': '1'}, '--gpus', '1')
    print('bench --gpus 1: plain %.1f images/s (%.3f ms), through RcclDataParallel %.1f images/s (%.3f ms), all-reduce exposed %.3f ms/step'
          % (plain['value'], plain['ms_per_step'], ddp['value'], ddp['ms_per_step'], ddp['allreduce_exposed_ms_per_step']))
    assert plain['metric'] == ddp['metric'] and plain['n_gpus'] == ddp['n_gpus'] == 1
    assert plain['config']['global_batch'] == ddp['config']['global_batch'] == 16
    assert 'allreduce_exposed_ms_per_step' not in plain and ddp['allreduce_exposed_ms_per_step'] >= 0.0
    assert abs(ddp['value'] / plain['value'] - 1.0) <= 0.05
    assert plain['step']['step_gflop'] > 6000 and 0.05 < plain['step']['step_frac'] < 0.6


# ---- MaxPoolDark (complex_yolov3_tiny.cfg) --------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', ['f16', 'f32'])
@pytest.mark.parametrize('k,stride,H', [(2, 1, 19), (2, 1, 8), (3, 2, 19), (4, 2, 9)])
def test_maxpooldark_kernels(dt, k, stride, H):
    """cy_maxpool_fwd / _bwd with pool_geometry's (output extent, leading pad) against the reference's MaxPoolDark
    (darknet2pytorch.py:30-59: replicate padding, then an unpadded pool) in float64: values exact, gradients at rounding."""
    import complex_yolov4_pytorch_amd.ops as ops
    from complex_yolov4_pytorch_amd.models.graph import pool_geometry
    from complex_yolov4_pytorch_amd.ops import View
    from tests.test_round4_cpu import _ref_maxpooldark
    code = ops.dtype_code(dt)
    N, C, W = 2, 32, H + 3
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, H, W, generator=g)
    x = x.half().float() if dt == 'f16' else x
    xr = x.double().requires_grad_(True)
    y = _ref_maxpooldark(xr, k, stride)
    (OH, pad), (OW, pad_w) = pool_geometry(H, k, stride), pool_geometry(W, k, stride)
    assert (OH, OW) == tuple(y.shape[2:]) and pad == pad_w
    dy = torch.randn(y.shape, generator=g)
    dy = dy.half().float() if dt == 'f16' else dy
    y.backward(dy.double())
    xv = View.from_nchw(x.to(DEV), code)
    yv = View.alloc(N, OH, OW, C, code, ld=C + 16)
    am = torch.zeros(ops.maxpool_argmax_bytes(N, H, OH, OW, C), dtype=torch.uint8, device=DEV)
    scratch = torch.empty(N * H * max(W, OW) * C, device=DEV)
    ops.maxpool_fwd(xv, yv, k, stride, pad, am, scratch)
    torch.testing.assert_close(yv.to_nchw().cpu(), y.detach().float(), rtol=0, atol=0)
    dyv = View.from_nchw(dy.to(DEV), code)
    dxv = View.from_nchw(torch.ones(N, C, H, W).to(DEV), code)
    ops.maxpool_bwd(dyv, am, dxv, k, stride, pad, True, scratch)
    tol = dict(rtol=2e-3, atol=2e-3) if dt == 'f16' else dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dxv.to_nchw().cpu(), 1 + xr.grad.float(), **tol)


@pytest.mark.parametrize('cfgname', ['complex_yolov3_tiny.cfg', 'complex_yolov3.cfg'])
def test_v3_tiny_train_step_on_device(cfgname):
    """The reference's other two cfgs: complex_yolov3_tiny.cfg (MaxPoolDark between its last two backbone convs; refused by
    round 3's plan; a 16-channel first layer) and complex_yolov3.cfg (Darknet-53 + FPN, leaky throughout) -- one fp32 parity-mode
    train step against the oracle: loss, outputs, parameter gradients (leaky kinks: most tensors tight, none wrong)."""
    from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
    from oracle import darknet_ref
    from tests.util import grad_rel_errors
    cfg = os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', cfgname)
    model = _model(cfgname, 'f32', deterministic=True)
    model.train()
    x, tg = syn.bev_images(2, 224, seed=7, sparsity=0.5), syn.targets(2, 4, 224, seed=7)
    loss, out = model(x.to(DEV), tg.to(DEV))
    loss.backward()
    net = darknet_ref.DarknetRef(parse_cfg(cfg))
    ps, bs = net.param_shapes()
    params = {k: v.requires_grad_(True) for k, v in syn.fill_state_dict(ps).items()}
    o_ref, l_ref, _ = net.forward(params, x, tg, True, True, syn.fill_state_dict(bs))
    l_ref.sum().backward()
    np.testing.assert_allclose(float(loss.detach()), float(l_ref.detach().sum()), rtol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), o_ref.detach().numpy(), rtol=2e-3, atol=2e-3)
    errs = grad_rel_errors([(n, p.grad.cpu()) for n, p in model.named_parameters()], {k: v.grad for k, v in params.items()})
    v = np.asarray(list(errs.values()))
    print('%s fp32 step vs oracle: loss %.5f / %.5f, gradient rel err median %.2e 90th pct %.2e max %.2e'
          % (cfgname, float(loss.detach()), float(l_ref.detach().sum()), float(np.median(v)), float(np.percentile(v, 90)), float(v.max())))
    if 'tiny' in cfgname:
        assert v.max() < 5e-2 and np.median(v) < 2e-3
    else:       # 75 leaky layers at batch 2: a pre-activation within round-off of zero flips its slope in one of the two evaluation orders (tests/util.py)
        assert v.max() < 1.0 and np.median(v) < 2e-2 and (v < 5e-2).mean() > 0.7
