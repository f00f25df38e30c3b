"""GPU parity tests, one C-ABI operator at a time: HIP kernel vs a plain PyTorch-CPU float32 (or float64)
statement of the same op / the oracle.  Tolerances: f32 parity mode 1e-4 relative (exact-f32 MFMA, only
summation order differs); f16 mode is compared against the same op computed in float64 on f16-rounded
inputs, so the only error left is the f16 rounding of the stored result (2^-11 relative)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.ops as ops  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd.ops import CY_BF16, CY_F16, CY_F32, View  # noqa: E402

DEV = 'cuda'


def _tol(dt):
    if dt == CY_BF16:       # bf16 rounding of the stored result: 2^-8 relative
        return dict(rtol=1.6e-2, atol=1.6e-2)
    return dict(rtol=2e-3, atol=2e-3) if dt == CY_F16 else dict(rtol=1e-4, atol=1e-5)


def _round(x, dt):
    if dt == CY_BF16:
        return x.bfloat16().float()
    return x.half().float() if dt == CY_F16 else x


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_tr16_lane_mapping():
    """gfx950 ds_read_b64_tr_b16: lane q of a 16-lane group receives column q of the 4x16 block whose rows
    are supplied by lanes (q>>2) -- the mapping conv_wgrad.hip relies on."""
    from tests import probes
    got = probes.probe_tr16().cpu().numpy().astype(np.int64) & 0xffff
    exp = np.zeros((64, 4), dtype=np.int64)
    for lane in range(64):
        q, g = lane & 15, lane >> 4
        for e in range(4):
            exp[lane, e] = (g * 4 + e) * 16 + q
    np.testing.assert_array_equal(got, exp)


CONV_CASES = [
    # N, Cin, H, W, Cout, ks, stride
    (2, 64, 19, 19, 128, 3, 1),
    (2, 128, 38, 38, 64, 1, 1),
    (1, 32, 64, 64, 64, 3, 2),
    (2, 256, 19, 19, 512, 1, 1),
    (3, 64, 21, 17, 32, 3, 1),
    (2, 32, 21, 17, 64, 3, 2),
]


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16, CY_F32])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_forward_and_stats(dt, case):
    N, Ci, H, W, Co, ks, st = case
    pad = (ks - 1) // 2
    x = _round(_rand(N, Ci, H, W, seed=1), dt)
    w = _round(_rand(Co, Ci, ks, ks, seed=2, scale=1 / math.sqrt(Ci * ks * ks)), dt)
    ref = F.conv2d(x.double(), w.double(), None, st, pad).float()
    xv = View.from_nchw(x.to(DEV), dt, ld=Ci + 2 * ops.chunk(dt)).channels(0, Ci)
    wf, _ = ops.pack_weights(w.to(DEV), Co, Ci, dt)
    OH, OW = ref.shape[2], ref.shape[3]
    out = View.alloc(N, OH, OW, Co, dt, ld=Co + 32, zero=True)
    rows = ops.conv_stats_rows(N * OH * OW, Co)
    stats = torch.zeros(rows + ops.bn_scratch_rows(), 2, Co, device=DEV)
    ops.conv_igemm(xv, wf, Co, out, ks, st, pad, flags=ops.CONV_STATS, stats=stats)
    got = out.to_nchw().cpu()
    torch.testing.assert_close(got, ref, **_tol(dt))
    s = stats[:rows].sum(0).cpu()
    torch.testing.assert_close(s[0], ref.double().sum((0, 2, 3)).float(), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(s[1], (ref.double() ** 2).sum((0, 2, 3)).float(), rtol=1e-3, atol=1e-2)


PIPE_CASES = [
    # N, Cin, H, W, Cout, ks, stride, (capacity, bn, pixels of the tile used) -- None = the library's own policy
    (2, 64, 19, 19, 128, 3, 1, (256, 128, 256)),
    (3, 64, 21, 17, 64, 3, 1, (256, 64, 200)),      # tiles straddle image boundaries, partly used capacity, Cout = 64
    (2, 128, 13, 29, 160, 3, 1, (128, 128, 128)),   # two channel tiles, the second one ragged
    (1, 192, 38, 38, 64, 3, 1, (128, 64, 97)),
    (2, 64, 38, 38, 128, 3, 1, (192, 128, 181)),    # 2 x 4 wave layout
    (2, 128, 38, 38, 256, 3, 1, (384, 128, 361)),   # 2-stage ring
    (1, 64, 40, 24, 64, 3, 1, (384, 64, 384)),
    (5, 64, 76, 76, 128, 3, 1, None),               # v4's stride-8 shape
    (2, 64, 7, 5, 64, 3, 1, (256, 64, 256)),        # one partial tile
    (2, 256, 19, 19, 512, 1, 1, (128, 128, 128)),   # 1 x 1
    (2, 128, 38, 38, 64, 1, 1, (256, 64, 256)),
    (1, 64, 64, 64, 128, 3, 2, (256, 128, 256)),    # stride 2: forward and the four dgrad parity classes
    (2, 64, 21, 17, 64, 3, 2, (128, 64, 128)),
]


@pytest.fixture
def pipe_policy():
    yield
    ops.conv_pipe_config(mode=1)


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16])
@pytest.mark.parametrize('variant', [0, 1, 2, 3])
@pytest.mark.parametrize('case', PIPE_CASES)
def test_conv_pipe_forward_dgrad(dt, variant, case, pipe_policy):
    """conv_pipe.hip (8 waves, 3-stage counted-vmcnt ring, 32x32x16 MFMA, LDS-transposed stores) against torch conv2d in
    float64: forward + BN statistics, the eval-mode epilogue with shortcut, dgrad and dgrad-accumulate, for every tile
    capacity / wave layout / ring depth; the 4-wave kernels run the same call to show the two paths agree."""
    N, Ci, H, W, Co, ks, st, tile = case
    if variant and tile is None:
        pytest.skip('variants are swept on forced tiles')
    if variant == 3 and tile[0] == 384:
        pytest.skip('loader / compute split needs the 3-stage ring (capacity <= 256)')
    pad = (ks - 1) // 2
    cfg = dict(mode=2, variant=variant)
    if tile:
        cfg.update(cap=tile[0], bn=tile[1], bm_eff=tile[2])
    ops.conv_pipe_config(**cfg)
    x = _round(_rand(N, Ci, H, W, seed=11), dt)
    w = _round(_rand(Co, Ci, ks, ks, seed=12, scale=1 / math.sqrt(Ci * ks * ks)), dt)
    ref = F.conv2d(x.double(), w.double(), None, st, pad).float()
    OH, OW = ref.shape[2], ref.shape[3]
    xv = View.from_nchw(x.to(DEV), dt, ld=Ci + 2 * ops.chunk(dt)).channels(0, Ci)
    wf, wd = ops.pack_weights(w.to(DEV), Co, Ci, dt)
    out = View.alloc(N, OH, OW, Co, dt, ld=Co + 32, zero=True)
    rows = ops.conv_stats_rows(N * OH * OW, Co)
    stats = torch.zeros(rows, 2, Co, device=DEV)
    n0 = ops.pipe_launches()
    ops.conv_igemm(xv, wf, Co, out, ks, st, pad, flags=ops.CONV_STATS, stats=stats)
    assert ops.pipe_launches() == n0 + 1
    torch.testing.assert_close(out.to_nchw().cpu(), ref, **_tol(dt))
    assert float(out.buf.view(-1, Co + 32)[:, Co:].abs().max()) == 0.0          # nothing written beside the view
    s = stats.sum(0).cpu()
    torch.testing.assert_close(s[0], ref.double().sum((0, 2, 3)).float(), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(s[1], (ref.double() ** 2).sum((0, 2, 3)).float(), rtol=1e-3, atol=1e-2)
    # the 4-wave kernels on the same call
    ops.conv_pipe_config(mode=0)
    out2 = View.alloc(N, OH, OW, Co, dt, zero=True)
    ops.conv_igemm(xv, wf, Co, out2, ks, st, pad)
    assert ops.pipe_launches() == n0 + 1
    torch.testing.assert_close(out2.to_nchw().cpu(), out.to_nchw().cpu(), **_tol(dt))
    ops.conv_pipe_config(**cfg)
    # eval-mode epilogue: BN affine + Mish + shortcut
    sc, sh = (_rand(Co, seed=14).abs() + 0.5).to(DEV), _rand(Co, seed=15).to(DEV)
    res = _round(_rand(N, Co, OH, OW, seed=16), dt)
    resv = View.from_nchw(res.to(DEV), dt, ld=Co + 8).channels(0, Co)
    out3 = View.alloc(N, OH, OW, Co, dt, zero=True)
    ops.conv_bn_act_eval(xv, wf, Co, out3, ks, st, pad, sc, sh, ops.ACT['mish'], resv)
    assert ops.pipe_launches() == n0 + 2
    z = ref.double() * sc.cpu().double().view(1, -1, 1, 1) + sh.cpu().double().view(1, -1, 1, 1)
    want = (z * torch.tanh(F.softplus(z)) + res.double()).float()
    tol = _tol(dt)
    torch.testing.assert_close(out3.to_nchw().cpu(), want, rtol=2 * tol['rtol'], atol=2 * tol['atol'])
    # dgrad (mirrored taps over the [Cin][tap, Cout] pack; stride 2 = the four parity classes in ONE launch when H and W are
    # even, else four launches) + accumulate
    if Co % 64 == 0:
        dy = _round(_rand(N, Co, OH, OW, seed=13), dt)
        wq = _round(_rand(Co, Ci, ks, ks, seed=12, scale=1 / math.sqrt(Co * ks * ks)), dt)
        _, wd = ops.pack_weights(wq.to(DEV), Co, Ci, dt)
        gref = torch.nn.grad.conv2d_input((N, Ci, H, W), wq.double(), dy.double(), st, pad).float()
        dx = View.alloc(N, H, W, Ci, dt, ld=Ci + 16, zero=True)
        n1 = ops.pipe_launches()
        ops.conv_igemm(View.from_nchw(dy.to(DEV), dt), wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED)
        assert ops.pipe_launches() == n1 + (4 if (st == 2 and (H % 2 or W % 2)) else 1)
        torch.testing.assert_close(dx.to_nchw().cpu(), gref, **tol)
        ops.conv_igemm(View.from_nchw(dy.to(DEV), dt), wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED | ops.CONV_ACCUM)
        torch.testing.assert_close(dx.to_nchw().cpu(), 2 * gref, rtol=2 * tol['rtol'], atol=2 * tol['atol'])


@pytest.mark.parametrize('dt', [CY_F16, CY_F32])
def test_conv_first_layer_padded_input(dt):
    """Layer 0: NCHW fp32 image -> NHWC with 3 channels padded to one 16-byte chunk; K = 9*chunk has a tail."""
    cp = ops.chunk(dt)
    x = syn.bev_images(2, 40, seed=3, sparsity=0.5)
    w = _round(_rand(32, 3, 3, 3, seed=4, scale=0.3), dt)
    ref = F.conv2d(_round(x, dt).double(), w.double(), None, 1, 1).float()
    xv = ops.nchw_to_nhwc(x.to(DEV), cp, dt)
    wf, _ = ops.pack_weights(w.to(DEV), 32, cp, dt, want_dgrad=False)
    out = View.alloc(2, 40, 40, 32, dt)
    ops.conv_igemm(xv, wf, 32, out, 3, 1, 1)
    torch.testing.assert_close(out.to_nchw().cpu(), ref, **_tol(dt))


@pytest.mark.parametrize('dt', [CY_F16, CY_F32])
def test_conv_head_bias_f32_out(dt):
    x = _round(_rand(2, 256, 19, 19, seed=5), dt)
    w = _round(_rand(30, 256, 1, 1, seed=6, scale=1 / 16), dt)
    b = _rand(30, seed=7)
    ref = F.conv2d(x.double(), w.double(), b.double()).float()
    xv = View.from_nchw(x.to(DEV), dt)
    wf, _ = ops.pack_weights(w.to(DEV), 32, 256, dt, want_dgrad=False)
    out = View.alloc(2, 19, 19, 30, CY_F32)
    ops.conv_igemm(xv, wf, 32, out, 1, 1, 0, flags=ops.CONV_BIAS_F32OUT, bias=b.to(DEV))
    torch.testing.assert_close(out.to_nchw().cpu(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dt', [CY_F16, CY_F32])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_dgrad(dt, case):
    N, Ci, H, W, Co, ks, st = case
    pad = (ks - 1) // 2
    w = _round(_rand(Co, Ci, ks, ks, seed=2, scale=1 / math.sqrt(Co * ks * ks)), dt)
    OH, OW = (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1
    dy = _round(_rand(N, Co, OH, OW, seed=8), dt)
    ref = torch.nn.grad.conv2d_input((N, Ci, H, W), w.double(), dy.double(), st, pad).float()
    dyv = View.from_nchw(dy.to(DEV), dt)
    _, wd = ops.pack_weights(w.to(DEV), Co, Ci, dt)
    dx = View.alloc(N, H, W, Ci, dt, ld=Ci + 16, zero=True)
    ops.conv_igemm(dyv, wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED)
    torch.testing.assert_close(dx.to_nchw().cpu(), ref, **_tol(dt))
    # gradient fan-in: a second dgrad accumulates
    ops.conv_igemm(dyv, wd, Ci, dx, ks, st, pad, flags=ops.CONV_TRANSPOSED | ops.CONV_ACCUM)
    tol = _tol(dt)
    torch.testing.assert_close(dx.to_nchw().cpu(), 2 * ref, rtol=2 * tol['rtol'], atol=2 * tol['atol'])


@pytest.mark.parametrize('mode', ['f16_tr', 'f16_scalar', 'f32'])
@pytest.mark.parametrize('case', CONV_CASES + [(2, 8, 24, 24, 32, 3, 1)])
def test_conv_wgrad(mode, case):
    dt = CY_F32 if mode == 'f32' else CY_F16
    N, Ci, H, W, Co, ks, st = case
    if Ci % ops.chunk(dt):
        pytest.skip('channel count below one chunk')
    pad = (ks - 1) // 2
    x = _round(_rand(N, Ci, H, W, seed=1), dt)
    OH, OW = (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1
    dy = _round(_rand(N, Co, OH, OW, seed=8, scale=0.5), dt)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, ks, ks), dy.double(), st, pad).float()
    xv, dyv = View.from_nchw(x.to(DEV), dt), View.from_nchw(dy.to(DEV), dt)
    split = max(2, min(7, ops.wgrad_split(N * OH * OW, Co, Ci, ks)))
    part = torch.full((split, Co, ks * ks * Ci), float('nan'), device=DEV)
    ops.conv_wgrad(dyv, xv, ks, st, pad, part, split, use_tr=0 if mode == 'f16_scalar' else 1)
    grad = torch.ones(Co, Ci, ks, ks, device=DEV)
    ops.wgrad_reduce(part, split, Co, Ci, ks, Co, Ci, 0.5, True, grad)
    torch.testing.assert_close(grad.cpu(), 1 + 0.5 * ref, rtol=1e-3, atol=1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize('dt', [CY_F16, CY_F32])
@pytest.mark.parametrize('act', ['mish', 'leaky', 'linear'])
@pytest.mark.parametrize('with_res', [False, True])
def test_bn_act_forward_backward(dt, act, with_res):
    N, C, H, W = 2, 64, 13, 11
    M = N * H * W
    x = _round(_rand(N, C, H, W, seed=11, scale=2.0) + 0.3, dt)
    res = _round(_rand(N, C, H, W, seed=12), dt)
    dy = _round(_rand(N, C, H, W, seed=13), dt)
    gamma, beta = 1 + 0.1 * _rand(C, seed=14), 0.1 * _rand(C, seed=15)
    rm, rv = torch.zeros(C), torch.ones(C)
    # reference on CPU (float64)
    xr = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rr = res.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    z = F.batch_norm(xr, rm_ref, rv_ref, g64, b64, True, 0.1, 1e-5)
    a = z * torch.tanh(F.softplus(z)) if act == 'mish' else (F.leaky_relu(z, 0.1) if act == 'leaky' else z)
    y = a + rr if with_res else a
    y.backward(dy.double())
    # device: statistics from partial sums (as the conv epilogue would emit them)
    xv = View.from_nchw(x.to(DEV), dt)
    stats = torch.stack((x.double().sum((0, 2, 3)), (x.double() ** 2).sum((0, 2, 3)))).float().view(1, 2, C).to(DEV)
    stats = torch.cat((stats, torch.zeros(63, 2, C, device=DEV)))
    dev = lambda t: t.clone().to(DEV)
    mean, invstd, scale, shift = (torch.empty(C, device=DEV) for _ in range(4))
    d_rm, d_rv = dev(rm), dev(rv)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    d_g, d_b = dev(gamma), dev(beta)
    ops.bn_finalize(stats, 64, C, M, d_g, d_b, d_rm, d_rv, nbt, 0.1, 1e-5, mean, invstd, scale, shift)
    assert float(stats.abs().max()) == 0.0          # the finaliser leaves the table zeroed
    torch.testing.assert_close(d_rm.cpu(), rm_ref.float(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d_rv.cpu(), rv_ref.float(), rtol=1e-5, atol=1e-6)
    assert int(nbt.item()) == 1
    yv = View.alloc(N, H, W, C, dt)
    resv = View.from_nchw(res.to(DEV), dt) if with_res else None
    ops.bn_act_fwd(xv, yv, resv, scale, shift, ops.ACT[act])
    torch.testing.assert_close(yv.to_nchw().cpu(), y.detach().float(), **_tol(dt))
    # backward
    dyv = View.from_nchw(dy.to(DEV), dt)
    rows = ops.bn_bwd_rows(M, C, dt)
    part = torch.zeros(rows + ops.bn_scratch_rows(), 2, C, device=DEV)
    ops.bn_act_bwd_reduce(xv, dyv, mean, invstd, scale, shift, ops.ACT[act], part)
    dgs, dbs = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    gg, gb = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    ops.bn_bwd_finalize(part, rows, C, dgs, dbs, gg, gb, 2.0)
    torch.testing.assert_close(gg.cpu(), 1 + 2 * g64.grad.float(), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(gb.cpu(), 1 + 2 * b64.grad.float(), rtol=2e-3, atol=2e-3)
    dxv = View.alloc(N, H, W, C, dt)
    rg = View.from_nchw(torch.ones(N, C, H, W).to(DEV), dt) if with_res else None
    ops.bn_act_bwd_apply(xv, dyv, dxv, rg, True, mean, invstd, scale, shift, dgs, dbs, ops.ACT[act])
    tol = dict(rtol=5e-3, atol=5e-3) if dt == CY_F16 else dict(rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(dxv.to_nchw().cpu(), xr.grad.float(), **tol)
    if with_res:
        torch.testing.assert_close(rg.to_nchw().cpu(), 1 + rr.grad.float(), **_tol(dt))


@pytest.mark.parametrize('dt', [CY_F16, CY_F32])
@pytest.mark.parametrize('k,stride', [(5, 1), (9, 1), (13, 1), (2, 2)])
def test_maxpool(dt, k, stride):
    N, C, H, W = 2, 32, 19, 19
    if stride == 2:
        H = W = 20
    x = _round(_rand(N, C, H, W, seed=21), dt)
    pad = k // 2 if stride == 1 else 0
    xr = x.double().requires_grad_(True)
    y = F.max_pool2d(xr, k, stride, pad)
    dy = _round(_rand(*y.shape, seed=22), dt)
    y.backward(dy.double())
    xv = View.from_nchw(x.to(DEV), dt)
    yv = View.alloc(N, y.shape[2], y.shape[3], C, dt, ld=C + 16)
    am = torch.zeros(ops.maxpool_argmax_bytes(N, H, y.shape[2], y.shape[3], C), dtype=torch.uint8, device=DEV)
    scratch = torch.empty(N * H * W * C, device=DEV)
    ops.maxpool_fwd(xv, yv, k, stride, pad, am, scratch)
    torch.testing.assert_close(yv.to_nchw().cpu(), y.detach().float(), rtol=0, atol=0)
    dyv = View.from_nchw(dy.to(DEV), dt)
    dxv = View.from_nchw(torch.ones(N, C, H, W).to(DEV), dt)
    ops.maxpool_bwd(dyv, am, dxv, k, stride, pad, True, scratch)
    torch.testing.assert_close(dxv.to_nchw().cpu(), 1 + xr.grad.float(), **_tol(dt))


@pytest.mark.parametrize('dt', [CY_F16, CY_F32])
def test_upsample_and_slices(dt):
    N, C, H, W = 2, 32, 7, 5
    x = _round(_rand(N, C, H, W, seed=31), dt)
    xv = View.from_nchw(x.to(DEV), dt)
    cat = View.alloc(N, 2 * H, 2 * W, C + 16, dt, zero=True)
    ops.upsample_fwd(xv, cat.channels(16, C), 2)
    ref = x.repeat_interleave(2, 2).repeat_interleave(2, 3)
    got = cat.to_nchw().cpu()
    torch.testing.assert_close(got[:, 16:], ref, rtol=0, atol=0)
    assert float(got[:, :16].abs().max()) == 0
    dy = _round(_rand(N, C, 2 * H, 2 * W, seed=32), dt)
    dyv = View.from_nchw(dy.to(DEV), dt)
    dxv = View.from_nchw(torch.ones(N, C, H, W).to(DEV), dt)
    ops.upsample_bwd(dyv, dxv, 2, True)
    ref_dx = 1 + dy.double().view(N, C, H, 2, W, 2).sum((3, 5)).float()
    torch.testing.assert_close(dxv.to_nchw().cpu(), ref_dx, **_tol(dt))
    # slice copy / accumulate / add
    a, b = _round(_rand(N, C, H, W, seed=33), dt), _round(_rand(N, C, H, W, seed=34), dt)
    av, bv = View.from_nchw(a.to(DEV), dt), View.from_nchw(b.to(DEV), dt)
    yv = View.alloc(N, H, W, C, dt, ld=C + 8, zero=True)
    ops.slice_copy(av, yv)
    torch.testing.assert_close(yv.to_nchw().cpu(), a, rtol=0, atol=0)
    ops.slice_copy(bv, yv, accumulate=True)
    torch.testing.assert_close(yv.to_nchw().cpu(), _round(a + b, dt), **_tol(dt))
    ops.slice_add(av, bv, yv)
    torch.testing.assert_close(yv.to_nchw().cpu(), _round(a + b, dt), **_tol(dt))


def test_f32_to_view_and_bias_grad():
    M, C = 1000, 30
    d = _rand(M, C, seed=41)
    y = View.alloc(1, 1, M, 32, CY_F16)
    ops.f32_to_view(d.to(DEV), M, C, 4.0, y, 32)
    got = y.to_nchw().cpu()[0, :, 0, :].t()
    torch.testing.assert_close(got[:, :30], (4 * d).half().float(), rtol=1e-3, atol=1e-3)
    assert float(got[:, 30:].abs().max()) == 0
    gb = torch.ones(C, device=DEV)
    ops.bias_grad(d.to(DEV), M, C, 0.25, gb)
    torch.testing.assert_close(gb.cpu(), 1 + 0.25 * d.sum(0), rtol=1e-5, atol=1e-5)


def test_table_driven_pack_and_reduce_match_single_calls():
    """cy_pack_weights_multi / cy_wgrad_reduce_multi (one launch for many convs) vs the per-tensor entry points."""
    shapes = [(32, 3, 3, 8), (64, 32, 3, 32), (30, 256, 1, 256), (128, 64, 1, 64)]   # Co, Ci, ks, CiPad
    items, singles = [], []
    for i, (Co, Ci, ks, cip) in enumerate(shapes):
        w = _rand(Co, Ci, ks, ks, seed=50 + i).to(DEV)
        cop = (Co + 31) // 32 * 32
        wf = torch.full((cop, ks * ks * cip), float('nan'), dtype=torch.float16, device=DEV)
        wd = torch.full((cip, ks * ks * cop), float('nan'), dtype=torch.float16, device=DEV) if i else None
        items.append((w, wf, wd, cop, cip))
        singles.append(ops.pack_weights(w, cop, cip, CY_F16, want_dgrad=bool(i)))
    desc, blocks = ops.make_pack_table(items, DEV)
    ops.pack_weights_multi(desc, blocks, CY_F16)
    for (w, wf, wd, _, _), (rf, rd) in zip(items, singles):
        torch.testing.assert_close(wf, rf, rtol=0, atol=0)
        if wd is not None:
            torch.testing.assert_close(wd, rd, rtol=0, atol=0)
    ritems, refs = [], []
    for i, (Co, Ci, ks, cip) in enumerate(shapes):
        cop, split = (Co + 31) // 32 * 32, (3, 40, 5, 130)[i]      # 40 / 130 slabs: 2 and 8 threads share a (co, ci) pair
        part = _rand(split, cop, ks * ks * cip, seed=60 + i).to(DEV)
        grad = torch.ones(Co, Ci, ks, ks, device=DEV)
        ref = torch.ones(Co, Ci, ks, ks, device=DEV)
        ops.wgrad_reduce(part, split, cop, cip, ks, Co, Ci, 0.25, True, ref)
        ritems.append((part, grad, split, cop, cip, ks, Co, Ci))
        refs.append(ref)
    desc, blocks = ops.make_reduce_table(ritems, DEV)
    ops.wgrad_reduce_multi(desc, blocks, 0.25, True)
    for it, ref in zip(ritems, refs):
        torch.testing.assert_close(it[1], ref, rtol=1e-5, atol=1e-5)


def test_fused_adam_matches_torch_adam():
    """cy_adam_multi vs torch.optim.Adam (same three parameter groups as the reference's create_optimizer), 5 steps."""
    from complex_yolov4_pytorch_amd.optim import FusedAdam
    shapes = [(64, 32, 3, 3), (64,), (64,), (30, 256, 1, 1), (30,), (1000,)]
    ref_p = [_rand(*s, seed=70 + i).to(DEV).requires_grad_(True) for i, s in enumerate(shapes)]
    my_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.Adam(ref_p[:2], lr=1e-2)
    ref.add_param_group({'params': ref_p[2:4], 'weight_decay': 5e-4})
    ref.add_param_group({'params': ref_p[4:]})
    mine = FusedAdam(my_p[:2], lr=1e-2)
    mine.add_param_group({'params': my_p[2:4], 'weight_decay': 5e-4})
    mine.add_param_group({'params': my_p[4:]})
    for step in range(5):
        for i, (a, b) in enumerate(zip(ref_p, my_p)):
            g = _rand(*a.shape, seed=100 + 10 * step + i).to(DEV)
            a.grad = g.clone()
            b.grad = g.clone()
        if step == 3:
            for opt in (ref, mine):
                opt.param_groups[0]['lr'] = 3e-3          # what a LambdaLR scheduler does
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)


def test_nonfinite_gradient_skips_the_fused_step():
    """cy_grad_nonfinite + the optimizers' device-side skip flag (dynamic loss scaling without a host round trip)."""
    from complex_yolov4_pytorch_amd.optim import FusedAdam, FusedSGD
    for n in (1000, 1003, 4):
        g = _rand(n, seed=5).to(DEV)
        flag = torch.ones(1, dtype=torch.int32, device=DEV)
        ops.grad_nonfinite(g, flag)
        assert int(flag) == 0
        for bad in (float('inf'), float('-inf'), float('nan')):
            g2 = g.clone(); g2[n - 1] = bad
            ops.grad_nonfinite(g2, flag)
            assert int(flag) == 1
    for cls, kw in ((FusedAdam, {}), (FusedSGD, dict(momentum=0.9, nesterov=True))):
        p = _rand(64, 32, seed=6).to(DEV).requires_grad_(True)
        before = p.detach().clone()
        opt = cls([p], lr=1e-2, **kw)
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        opt.skip_flag = flag
        p.grad = _rand(64, 32, seed=7).to(DEV)
        p.grad[3, 3] = float('inf')
        ops.grad_nonfinite(p.grad, flag)
        opt.step()
        torch.testing.assert_close(p.detach(), before, rtol=0, atol=0)      # skipped as a whole
        p.grad[3, 3] = 0.5
        ops.grad_nonfinite(p.grad, flag)
        opt.step()
        assert not torch.equal(p.detach(), before)


@pytest.mark.parametrize('nesterov', [True, False])
def test_fused_sgd_matches_torch_sgd(nesterov):
    """cy_sgd_multi vs torch.optim.SGD(momentum, nesterov) with the reference's three parameter groups, 5 steps."""
    from complex_yolov4_pytorch_amd.optim import FusedSGD
    shapes = [(64, 32, 3, 3), (64,), (64,), (30, 256, 1, 1), (30,), (1000,)]
    ref_p = [_rand(*s, seed=170 + i).to(DEV).requires_grad_(True) for i, s in enumerate(shapes)]
    my_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref = torch.optim.SGD(ref_p[:2], lr=1e-2, momentum=0.949, nesterov=nesterov)
    ref.add_param_group({'params': ref_p[2:4], 'weight_decay': 5e-4})
    ref.add_param_group({'params': ref_p[4:]})
    mine = FusedSGD(my_p[:2], lr=1e-2, momentum=0.949, nesterov=nesterov)
    mine.add_param_group({'params': my_p[2:4], 'weight_decay': 5e-4})
    mine.add_param_group({'params': my_p[4:]})
    for step in range(5):
        for i, (a, b) in enumerate(zip(ref_p, my_p)):
            g = _rand(*a.shape, seed=300 + 10 * step + i).to(DEV)
            a.grad = g.clone()
            b.grad = g.clone()
        if step == 3:
            for opt in (ref, mine):
                opt.param_groups[0]['lr'] = 3e-3
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)
    for a, b in zip(ref_p, my_p):
        torch.testing.assert_close(mine.state[b]['momentum_buffer'], ref.state[a]['momentum_buffer'], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('kind', ['adam', 'sgd'])
def test_fused_optimizer_resume_from_state_dict(kind):
    """after optimizer.load_state_dict (resume from a Utils_*.pth, reference train.py:123-133) the
    fused optimizers continue exactly like torch's: Adam's bias correction goes on at t+1 (not t = 1) and SGD's
    restored momentum buffer is used, not overwritten by the first gradient."""
    from complex_yolov4_pytorch_amd.optim import FusedAdam, FusedSGD
    shapes = [(32, 16, 3, 3), (32,), (500,)]

    def make(ps):
        if kind == 'adam':
            return torch.optim.Adam(ps, lr=1e-2), FusedAdam
        return torch.optim.SGD(ps, lr=1e-2, momentum=0.9, nesterov=True), FusedSGD
    ref_p = [_rand(*s, seed=270 + i).to(DEV).requires_grad_(True) for i, s in enumerate(shapes)]
    my_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    ref, cls = make(ref_p)
    kw = dict(lr=1e-2) if kind == 'adam' else dict(lr=1e-2, momentum=0.9, nesterov=True)
    mine = cls(my_p, **kw)

    def step(opts_params, seed):
        for opt, ps in opts_params:
            for i, p_ in enumerate(ps):
                p_.grad = _rand(*p_.shape, seed=seed + i).to(DEV)
            opt.step()
    for t in range(4):
        step(((ref, ref_p), (mine, my_p)), 400 + 10 * t)
    import copy
    sd = copy.deepcopy(mine.state_dict())
    my_p2 = [p.detach().clone().requires_grad_(True) for p in my_p]
    resumed = cls(my_p2, **kw)
    resumed.load_state_dict(sd)
    for t in range(4, 7):
        step(((ref, ref_p), (resumed, my_p2)), 400 + 10 * t)
    for a, b in zip(ref_p, my_p2):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6)
    if kind == 'adam':
        assert all(int(st['step']) == 7 for st in resumed.state.values())


@pytest.mark.parametrize('dt', [CY_F16, CY_F32])
@pytest.mark.parametrize('act', ['mish', 'leaky', 'linear'])
def test_conv_bn_act_eval_fused(dt, act):
    """Eval-mode conv block in one kernel vs conv -> affine -> activation (+ shortcut) in float64."""
    N, Ci, H, W, Co, ks, st = 2, 64, 19, 19, 128, 3, 1
    x = _round(_rand(N, Ci, H, W, seed=1), dt)
    w = _round(_rand(Co, Ci, ks, ks, seed=2, scale=1 / math.sqrt(Ci * ks * ks)), dt)
    res = _round(_rand(N, Co, H, W, seed=3), dt)
    scale, shift = 1 + 0.2 * _rand(Co, seed=4), 0.3 * _rand(Co, seed=5)
    z = F.conv2d(x.double(), w.double(), None, st, 1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    a = z * torch.tanh(F.softplus(z)) if act == 'mish' else (F.leaky_relu(z, 0.1) if act == 'leaky' else z)
    ref = (a + res.double()).float()
    xv, rv = View.from_nchw(x.to(DEV), dt), View.from_nchw(res.to(DEV), dt)
    wf, _ = ops.pack_weights(w.to(DEV), Co, Ci, dt, want_dgrad=False)
    out = View.alloc(N, H, W, Co, dt, ld=Co + 32, zero=True)
    ops.conv_bn_act_eval(xv, wf, Co, out, ks, st, 1, scale.to(DEV), shift.to(DEV), ops.ACT[act], rv)
    tol = dict(rtol=3e-3, atol=3e-3) if dt == CY_F16 else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out.to_nchw().cpu(), ref, **tol)
