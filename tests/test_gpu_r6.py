"""Round 6 GPU parity tests: the slab kernel of conv_pipe.hip (3x3 / stride 1 / pad 1: one halo'd slab of input pixels per
64-channel chunk in LDS serves all nine taps) against torch conv2d in float64 and against the implicit-GEMM kernels of the same
library on the same call.  Reference work unit: /root/reference/src/models/darknet2pytorch.py:247-278."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.ops as ops  # noqa: E402
from complex_yolov4_pytorch_amd.ops import CY_BF16, CY_F16, View  # noqa: E402

DEV = 'cuda'


def _tol(dt):
    return dict(rtol=1.6e-2, atol=1.6e-2) if dt == CY_BF16 else dict(rtol=2e-3, atol=2e-3)


def _round(x, dt):
    return x.bfloat16().float() if dt == CY_BF16 else x.half().float()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


SLAB_CASES = [
    # N, Cin, H, W, Cout, (hint, pixels of the tile used; 0 = the policy's)
    (2, 64, 19, 19, 128, (13, 0)),          # one chunk: a single slab buffer
    (3, 128, 21, 17, 128, (13, 200)),       # tiles straddle image boundaries, partly used capacity, two chunks
    (2, 128, 13, 29, 160, (12, 0)),         # two channel tiles, the second one ragged; W > H
    (2, 192, 38, 38, 256, (12, 181)),       # 2 x 4 wave layout, three chunks (odd: the slab buffers end on buffer 0)
    (2, 256, 38, 38, 256, (13, 256)),
    (5, 128, 76, 76, 128, (11, 0)),         # v4's stride-8 shape: the widest slab (256 + 154 rows)
    (2, 64, 7, 5, 128, (13, 0)),            # one partial tile, images smaller than the halo
    (1, 128, 3, 40, 128, (12, 97)),         # three image rows: every pixel is a border pixel of some tap
    (16, 512, 19, 19, 1024, (11, 0)),       # v4's deepest 3x3 at batch 16
]


@pytest.fixture
def slab_config():
    yield
    ops.conv_slab_config(1, 0)
    ops.conv_pipe_config(mode=1)


# cy_conv_slab_config modes: the shipped K-split wave pairs with the ring depth the LDS allows / 3 stages / 4 stages (falls back
# to the default kernels where 4 do not fit), and the loader / compute variant
SLAB_MODES = {'kauto': 1, 'k3': 1 + 4, 'k4': 1 + 8, 'loaders': 1 + 2}


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16])
@pytest.mark.parametrize('mode', list(SLAB_MODES))
@pytest.mark.parametrize('case', SLAB_CASES)
def test_slab_conv_forward_dgrad(dt, mode, case, slab_config):
    """Forward + BN statistics, the eval-mode epilogue with shortcut, dgrad and dgrad-accumulate on the slab kernel: vs float64
    conv2d (storage rounding only) and vs the 4-wave implicit-GEMM kernels on the same call."""
    N, Ci, H, W, Co, (hint, eff) = case
    if mode == 'k4' and W >= 76:
        pytest.skip('four weight stages + two 76-wide slabs exceed the LDS')
    ops.conv_slab_config(SLAB_MODES[mode], eff)
    x = _round(_rand(N, Ci, H, W, seed=21), dt)
    w = _round(_rand(Co, Ci, 3, 3, seed=22, scale=1 / math.sqrt(Ci * 9)), dt)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).float()
    xv = View.from_nchw(x.to(DEV), dt, ld=Ci + 2 * ops.chunk(dt)).channels(0, Ci)
    wf, _ = ops.pack_weights(w.to(DEV), Co, Ci, dt)
    out = View.alloc(N, H, W, Co, dt, ld=Co + 32, zero=True)
    stats = torch.zeros(ops.conv_stats_rows(N * H * W, Co), 2, Co, device=DEV)
    n0 = ops.pipe_launches()
    ops.conv_igemm(xv, wf, Co, out, 3, 1, 1, flags=ops.CONV_STATS, stats=stats, tile=hint)
    assert ops.pipe_launches() == n0 + 1
    torch.testing.assert_close(out.to_nchw().cpu(), ref, **_tol(dt))
    assert float(out.buf.view(-1, Co + 32)[:, Co:].abs().max()) == 0.0          # nothing written beside the view
    s = stats.sum(0).cpu()
    torch.testing.assert_close(s[0], ref.double().sum((0, 2, 3)).float(), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(s[1], (ref.double() ** 2).sum((0, 2, 3)).float(), rtol=1e-3, atol=1e-2)
    # the 4-wave kernels on the same call: the same sums in another order
    out2 = View.alloc(N, H, W, Co, dt, zero=True)
    ops.conv_igemm(xv, wf, Co, out2, 3, 1, 1, tile=1)
    torch.testing.assert_close(out2.to_nchw().cpu(), out.to_nchw().cpu(), **_tol(dt))
    # eval-mode epilogue: BN affine + Mish + shortcut
    sc, sh = (_rand(Co, seed=14).abs() + 0.5).to(DEV), _rand(Co, seed=15).to(DEV)
    res = _round(_rand(N, Co, H, W, seed=16), dt)
    resv = View.from_nchw(res.to(DEV), dt, ld=Co + 8).channels(0, Co)
    out3 = View.alloc(N, H, W, Co, dt, zero=True)
    ops.conv_bn_act_eval(xv, wf, Co, out3, 3, 1, 1, sc, sh, ops.ACT['mish'], resv, tile=hint)
    assert ops.pipe_launches() == n0 + 2
    z = ref.double() * sc.cpu().double().view(1, -1, 1, 1) + sh.cpu().double().view(1, -1, 1, 1)
    want = (z * torch.tanh(F.softplus(z)) + res.double()).float()
    tol = _tol(dt)
    torch.testing.assert_close(out3.to_nchw().cpu(), want, rtol=2 * tol['rtol'], atol=2 * tol['atol'])
    # dgrad: the produced gradient has Ci channels -> the slab kernel wants more than 64 of them
    if Co % 64 == 0 and Ci > 64:
        dy = _round(_rand(N, Co, H, W, seed=13), dt)
        wq = _round(_rand(Co, Ci, 3, 3, seed=12, scale=1 / math.sqrt(Co * 9)), dt)
        _, wd = ops.pack_weights(wq.to(DEV), Co, Ci, dt)
        gref = torch.nn.grad.conv2d_input((N, Ci, H, W), wq.double(), dy.double(), 1, 1).float()
        dx = View.alloc(N, H, W, Ci, dt, ld=Ci + 16, zero=True)
        n1 = ops.pipe_launches()
        dyv = View.from_nchw(dy.to(DEV), dt)
        ops.conv_igemm(dyv, wd, Ci, dx, 3, 1, 1, flags=ops.CONV_TRANSPOSED, tile=hint)
        assert ops.pipe_launches() == n1 + 1
        torch.testing.assert_close(dx.to_nchw().cpu(), gref, **tol)
        ops.conv_igemm(dyv, wd, Ci, dx, 3, 1, 1, flags=ops.CONV_TRANSPOSED | ops.CONV_ACCUM, tile=hint)
        torch.testing.assert_close(dx.to_nchw().cpu(), 2 * gref, rtol=2 * tol['rtol'], atol=2 * tol['atol'])
        assert float(dx.buf.view(-1, Ci + 16)[:, Ci:].abs().max()) == 0.0


@pytest.mark.parametrize('dt', [CY_F16, CY_BF16])
@pytest.mark.parametrize('mode', ['kauto', 'loaders'])
@pytest.mark.parametrize('accum', [False, True])
@pytest.mark.parametrize('case', [(128, 128, 38, 4), (256, 256, 19, 16), (128, 256, 76, 2)])
def test_slab_dgrad_bn_sums(dt, mode, accum, case, slab_config):
    """cy_conv_dgrad_bn_sums on the slab kernel = the slab dgrad followed by cy_bn_act_bwd_reduce over (raw, stored gradient):
    same gradient tensor bit for bit, same (d beta, d gamma) sums up to fp32 summation order."""
    Cdy, Cg, H, N = case
    ops.conv_slab_config(SLAB_MODES[mode], 0)
    g = torch.Generator().manual_seed(Cdy * 3 + Cg + H)
    tdt = ops.torch_dtype(dt)
    dy = View.alloc(N, H, H, Cdy, dt); dy.buf.copy_(torch.randn(dy.buf.numel(), generator=g).to(tdt))
    raw = View.alloc(N, H, H, Cg, dt); raw.buf.copy_(torch.randn(raw.buf.numel(), generator=g).to(tdt))
    w = torch.randn(Cdy, Cg, 3, 3, generator=g).to(DEV) * (1.0 / (9 * Cdy) ** 0.5)
    _, wd = ops.pack_weights(w, Cdy, Cg, dt)
    vec = torch.stack([torch.randn(Cg, generator=g) * 0.1, torch.rand(Cg, generator=g) + 0.5,
                       torch.rand(Cg, generator=g) + 0.5, torch.randn(Cg, generator=g) * 0.2]).to(DEV)
    a = ops.ACT['mish']
    base = torch.randn(raw.buf.numel(), generator=g).to(DEV).to(tdt)
    flags = ops.CONV_TRANSPOSED | (ops.CONV_ACCUM if accum else 0)
    M = N * H * H
    rows = ops.conv_stats_rows(M, Cg)
    for hint in (12, 13):
        g1 = View.alloc(N, H, H, Cg, dt); g1.buf.copy_(base)
        n0 = ops.pipe_launches()
        ops.conv_igemm(dy, wd, Cg, g1, 3, 1, 1, flags=flags, tile=hint)
        part = torch.zeros(ops.bn_bwd_rows(M, Cg, dt), 2, Cg, device=DEV)
        ops.bn_act_bwd_reduce(raw, g1, vec[0], vec[1], vec[2], vec[3], a, part)
        ref_sums = part.double().sum(0)
        g2 = View.alloc(N, H, H, Cg, dt); g2.buf.copy_(base)
        tbl = torch.zeros(rows, 2, Cg, device=DEV)
        ops.conv_dgrad_bn_sums(dy, wd, Cg, g2, 3, 1, 1, raw, vec[0], vec[1], vec[2], vec[3], a, tbl, flags=flags, tile=hint)
        assert ops.pipe_launches() == n0 + 2
        assert torch.equal(g2.buf, g1.buf), hint
        # against the implicit-GEMM kernel's gradient as well (another summation order)
        g3 = View.alloc(N, H, H, Cg, dt); g3.buf.copy_(base)
        ops.conv_igemm(dy, wd, Cg, g3, 3, 1, 1, flags=flags, tile=4)
        tol = 2e-2 if dt == CY_BF16 else 3e-3
        torch.testing.assert_close(g2.buf.float(), g3.buf.float(), rtol=tol, atol=tol)
        got = tbl.double().sum(0)
        scale = ref_sums.abs().max(1, keepdim=True).values + 1e-6
        assert float(((got - ref_sums).abs() / scale).max()) < 2e-5, (hint, float(((got - ref_sums).abs() / scale).max()))


def test_slab_hint_falls_through_where_the_kernel_does_not_apply(slab_config):
    """Hints 11-13 on a call the slab kernel does not take (1x1, stride 2, 64 output channels, slab switched off) run the
    library's default kernel with the same results."""
    dt = CY_F16
    for (Ci, Co, ks, st) in ((64, 128, 1, 1), (64, 128, 3, 2), (128, 64, 3, 1)):
        pad = (ks - 1) // 2
        x = _round(_rand(2, Ci, 20, 20, seed=3), dt)
        w = _round(_rand(Co, Ci, ks, ks, seed=4, scale=1 / math.sqrt(Ci * ks * ks)), dt)
        ref = F.conv2d(x.double(), w.double(), None, st, pad).float()
        xv = View.from_nchw(x.to(DEV), dt)
        wf, _ = ops.pack_weights(w.to(DEV), Co, Ci, dt)
        out = View.alloc(2, ref.shape[2], ref.shape[3], Co, dt)
        ops.conv_igemm(xv, wf, Co, out, ks, st, pad, tile=13)
        torch.testing.assert_close(out.to_nchw().cpu(), ref, **_tol(dt))
    ops.conv_slab_config(0, 0)
    x = _round(_rand(2, 128, 20, 20, seed=3), dt)
    w = _round(_rand(128, 128, 3, 3, seed=4, scale=1 / math.sqrt(128 * 9)), dt)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).float()
    out = View.alloc(2, 20, 20, 128, dt)
    wf, _ = ops.pack_weights(w.to(DEV), 128, 128, dt)
    n0 = ops.pipe_launches()
    ops.conv_igemm(View.from_nchw(x.to(DEV), dt), wf, 128, out, 3, 1, 1, tile=13)
    assert ops.pipe_launches() == n0        # the 4-wave kernels: the library's default for a training launch without a hint
    torch.testing.assert_close(out.to_nchw().cpu(), ref, **_tol(dt))
