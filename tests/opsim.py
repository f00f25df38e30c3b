"""CPU simulator of the operator layer (complex_yolov4_pytorch_amd.ops) -- TEST INFRASTRUCTURE ONLY.

Each function restates the contract of one C-ABI entry point (include/cyolo_hip.h) with plain PyTorch-CPU
ops on the same NHWC views, so the host logic -- the cfg lowering (models/graph.py), the engine's buffer
wiring, gradient fan-in flags, the Darknet/YoloLayer modules, the data-parallel wrapper -- can be executed
and checked against the oracle without a GPU.  ``install(monkeypatch)`` swaps these in for the real
bindings inside a test; product code never imports this module.
"""
import torch
import torch.nn.functional as F

import complex_yolov4_pytorch_amd.ops as real
from oracle import yolo_layer_ref

CONV_STATS, CONV_BIAS_F32OUT, CONV_ACCUM, CONV_TRANSPOSED = 1, 2, 4, 8


def _t(v):
    """View -> NHWC strided torch tensor aliasing the storage."""
    flat = v.buf.view(-1)
    return torch.as_strided(flat, (v.N, v.H, v.W, v.C), (v.H * v.W * v.ld, v.W * v.ld, v.ld, 1), v.off)


def _nchw(v):
    return _t(v).permute(0, 3, 1, 2).float()


def _store(v, x_nchw, accumulate=False):
    t = _t(v)
    val = x_nchw.permute(0, 2, 3, 1)
    t.copy_((t.float() + val if accumulate else val).to(t.dtype))


def _act(z, act):
    if act == 2:
        return z * torch.tanh(F.softplus(z))
    if act == 1:
        return F.leaky_relu(z, 0.1)
    return z


def check_device_tensor(t, who):
    return None


def nchw_to_nhwc(x, cpad, dt, out=None):
    t = _t(out)
    t.zero_()
    t[..., :x.shape[1]] = x.permute(0, 2, 3, 1).to(t.dtype)
    return out


def pack_weights_into(w, co_pad, ci_pad, dt, wf, wd):
    Co, Ci, ks, _ = w.shape
    full = torch.zeros(co_pad, ks * ks, ci_pad)
    full[:Co, :, :Ci] = w.detach().permute(0, 2, 3, 1).reshape(Co, ks * ks, Ci)
    wf.copy_(full.reshape(co_pad, -1).to(wf.dtype))
    if wd is not None:
        wd.copy_(full.permute(2, 1, 0).reshape(ci_pad, -1).to(wd.dtype))


def make_pack_table(items, device):
    return (items, None)


def pack_weights_multi(desc, blocks, dt):
    for item in desc:
        w, wf, wd, cop, cip = item[:5]
        if len(item) > 5 and item[5] and wd is not None:      # ks = 1, a column range of a wider dgrad matrix: wd is a strided view
            Co, Ci = w.shape[:2]
            full = torch.zeros(cop, cip)
            full[:Co, :Ci] = w.detach().reshape(Co, Ci)
            wf.copy_(full.to(wf.dtype))
            wd.copy_(full.t().to(wd.dtype))
            continue
        pack_weights_into(w, cop, cip, dt, wf, wd)


def make_reduce_table(items, device):
    return (items, None)


def wgrad_reduce_multi(desc, blocks, scale, accumulate):
    for item in desc:
        part, grad, split, corows, cip, ks, Co, Ci = item[:8]
        wgrad_reduce(part, split, corows, cip, ks, Co, Ci, scale, accumulate, grad)
        if len(item) > 8 and item[8] & 1:           # atomic-mode slab: left zeroed by the fold
            part.view(-1)[:split * corows * ks * ks * cip].zero_()


def conv_dgrad_bn_sums(g, w, wrows, out, ks, stride, pad, raw, mean, invstd, scale, shift, act, sums, flags=0, tile=0):
    conv_igemm(g, w, wrows, out, ks, stride, pad, flags=flags)
    C = out.C
    dz = _dz(raw, out, scale, shift, act)
    xh = (_nchw(raw) - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    rows = real.conv_stats_rows(out.M, C)
    p = sums.view(-1)[:rows * 2 * C].view(rows, 2, C)
    assert float(p.abs().max()) == 0.0, 'the sums table must be zero on entry'
    p[0, 0] = dz.sum((0, 2, 3))
    p[0, 1] = (dz * xh).sum((0, 2, 3))


def conv_igemm(g, w, wrows, out, ks, stride, pad, flags=0, bias=None, stats=None, tile=0):
    x = _nchw(g)
    kk = ks * ks
    if flags & CONV_TRANSPOSED:
        # w: [ci rows][tap][co = g.C]  ->  conv weight [co, ci, kh, kw]
        wt = w.float().reshape(w.shape[0], kk, g.C)[:out.C].permute(2, 0, 1).reshape(g.C, out.C, ks, ks)
        y = torch.nn.grad.conv2d_input((out.N, out.C, out.H, out.W), wt, x, stride, pad)
    else:
        wt = w.float().reshape(w.shape[0], kk, g.C)[:out.C].permute(0, 2, 1).reshape(out.C, g.C, ks, ks)
        y = F.conv2d(x, wt, None, stride, pad)
    if flags & CONV_STATS:
        rows = real.conv_stats_rows(out.M, out.C)
        s = stats.view(-1)[:rows * 2 * out.C].view(rows, 2, out.C)
        assert float(s.abs().max()) == 0.0, 'stats table must be zero on entry'
        s[0, 0] = y.sum((0, 2, 3))
        s[0, 1] = (y * y).sum((0, 2, 3))
    if flags & CONV_BIAS_F32OUT and bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    _store(out, y, accumulate=bool(flags & CONV_ACCUM))


def conv_bn_act_eval(g, w, wrows, out, ks, stride, pad, scale, shift, act, res=None, tile=0):
    x = _nchw(g)
    wt = w.float().reshape(w.shape[0], ks * ks, g.C)[:out.C].permute(0, 2, 1).reshape(out.C, g.C, ks, ks)
    y = F.conv2d(x, wt, None, stride, pad)
    a = _act(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), act)
    if res is not None:
        a = a + _nchw(res)
    _store(out, a)


def conv_wgrad(dy, x, ks, stride, pad, part, split, use_tr=1):
    g, a = _nchw(dy), _nchw(x)
    dw = torch.nn.grad.conv2d_weight(a, (dy.C, x.C, ks, ks), g, stride, pad)     # [Co, Ci, kh, kw]
    ncols = ks * ks * x.C
    p = part.view(-1)[:split * dy.C * ncols].view(split, dy.C, ncols)
    p.zero_()
    p[0] = dw.permute(0, 2, 3, 1).reshape(dy.C, ncols)


def wgrad_reduce(part, split, co_rows, ci_pad, ks, Co, Ci, scale, accumulate, grad):
    ncols = ks * ks * ci_pad
    # (rows [0, Co) of every slab; the slab stride is co_rows * ncols -- `part` may start inside a wider slab and end with it)
    p = torch.as_strided(part.view(-1), (split, Co, ks * ks, ci_pad), (co_rows * ncols, ncols, ci_pad, 1)).sum(0)
    g = p[:Co, :, :Ci].permute(0, 2, 1).reshape(Co, Ci, ks, ks) * scale
    grad.copy_(grad + g if accumulate else g)


def bn_finalize(stats, rows, C, count, gamma, beta, rmean, rvar, nbt, momentum, eps, mean, invstd, scale, shift):
    s = stats.view(-1)[:rows * 2 * C].view(rows, 2, C).double().sum(0)
    m = s[0] / count
    var = (s[1] / count - m * m).clamp(min=0)
    stats.view(-1)[:rows * 2 * C].zero_()      # the finaliser leaves the binned table zeroed
    mean.copy_(m.float())
    invstd.copy_((1 / torch.sqrt(var + eps)).float())
    scale.copy_(gamma * invstd)
    shift.copy_(beta - mean * scale)
    if rmean is not None:
        unb = var * count / (count - 1) if count > 1 else var
        rmean.mul_(1 - momentum).add_(momentum * m.float())
        rvar.mul_(1 - momentum).add_(momentum * unb.float())
    if nbt is not None:
        nbt += 1


def bn_eval_affine(gamma, beta, rmean, rvar, eps, scale, shift):
    scale.copy_(gamma / torch.sqrt(rvar + eps))
    shift.copy_(beta - rmean * scale)


def bn_act_fwd(x, y, res, scale, shift, act):
    z = _nchw(x) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    a = _act(z, act)
    if res is not None:
        a = a + _nchw(res)
    _store(y, a)


def bn_act_fwd_fused(x, y, res, bins, rows, gamma, beta, rmean, rvar, nbt, momentum, eps, vec, zero_table, act, stats_ld=0, stats_c0=0,
                     vec_ld=0):
    C = x.C
    assert vec_ld in (0, vec.stride(0)), 'vec_ld names the row stride of the (possibly shared) [4][vec_ld] block'
    ld = stats_ld or C
    st = bins.view(-1)[:rows * 2 * ld].view(rows, 2, ld)[:, :, stats_c0:stats_c0 + C].reshape(-1).clone()
    bn_finalize(st, rows, C, x.M, gamma, beta, rmean, rvar, nbt, momentum, eps, vec[0], vec[1], vec[2], vec[3])
    if zero_table is not None:
        zero_table.zero_()
    bn_act_fwd(x, y, res, vec[2], vec[3], act)


def bn_act_bwd_apply_fused(x, dy, dx, res_grad, res_accum, mean, invstd, scale, shift, bins, rows, ggamma, gbeta, gscale,
                           zero_table, act, bins_ld=0, bins_c0=0):
    C = x.C
    dgs, dbs = torch.zeros(C), torch.zeros(C)
    ld = bins_ld or C
    st = bins.view(-1)[:rows * 2 * ld].view(rows, 2, ld)[:, :, bins_c0:bins_c0 + C].reshape(-1).clone()
    bn_bwd_finalize(st, rows, C, dgs, dbs, ggamma, gbeta, gscale)
    if zero_table is not None:
        zero_table.zero_()
    bn_act_bwd_apply(x, dy, dx, res_grad, res_accum, mean, invstd, scale, shift, dgs, dbs, act)


def _dz(x, dy, scale, shift, act):
    with torch.enable_grad():
        z = (_nchw(x) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).detach().requires_grad_(True)
        a = _act(z, act)
        (g,) = torch.autograd.grad(a, z, _nchw(dy))
    return g


def bn_act_bwd_reduce(x, dy, mean, invstd, scale, shift, act, part, rows=None):
    C = x.C
    rows = real.bn_bwd_rows(x.M, C, x.dt) if rows is None else rows
    dz = _dz(x, dy, scale, shift, act)
    xh = (_nchw(x) - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    p = part.view(-1)[:rows * 2 * C].view(rows, 2, C)
    assert float(p.abs().max()) == 0.0, 'partial table must be zero on entry'
    p[0, 0] = dz.sum((0, 2, 3))
    p[0, 1] = (dz * xh).sum((0, 2, 3))


def fold_rows(bins, rows, W, out):
    p = bins.view(-1)[:rows * W].view(rows, W)
    out.view(-1)[:W] = p.sum(0)
    p.zero_()
    return 1


def bias_grad_det(dlogits, M, C, scale, gbias, scratch, scale_dev=None):
    bias_grad(dlogits, M, C, scale, gbias, scale_dev=scale_dev)


def bn_bwd_finalize(part, rows, C, dgs, dbs, ggamma, gbeta, gscale):
    p = part.view(-1)[:rows * 2 * C].view(rows, 2, C).sum(0)
    part.view(-1)[:rows * 2 * C].zero_()
    dbs[:C] = p[0]
    dgs[:C] = p[1]
    if gbeta is not None:
        gbeta += gscale * p[0]
    if ggamma is not None:
        ggamma += gscale * p[1]


def bn_act_bwd_apply(x, dy, dx, res_grad, res_accum, mean, invstd, scale, shift, dgs, dbs, act):
    C, M = x.C, x.M
    g = _nchw(dy).clone()
    if res_grad is not None:
        _store(res_grad, g, accumulate=bool(res_accum))
    dz = _dz(x, dy, scale, shift, act)
    xh = (_nchw(x) - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    out = scale.view(1, -1, 1, 1) * (dz - dbs[:C].view(1, -1, 1, 1) / M - xh * dgs[:C].view(1, -1, 1, 1) / M)
    _store(dx, out)


def maxpool_argmax_bytes(N, H, OH, OW, C):
    return N * (OH + H) * OW * C


def maxpool_fwd(x, y, k, stride, pad, argmax, scratch=None):
    a = _nchw(x)
    N, C, H, W = a.shape
    # the C ABI's contract: windows start at o * stride - pad and are clipped to the image; the output extent is the caller's
    # (y.H, y.W) -- the trailing padding may differ from the leading one (MaxPoolDark, size 2 / stride 1)
    ph, pw = max(0, (y.H - 1) * stride + k - H - pad), max(0, (y.W - 1) * stride + k - W - pad)
    ap = F.pad(a, (pad, pw, pad, ph), value=float('-inf'))
    yv, idx = F.max_pool2d(ap, k, stride, 0, return_indices=True)
    yv, idx = yv[:, :, :y.H, :y.W], idx[:, :, :y.H, :y.W]
    Wp = ap.shape[3]
    _store(y, yv)
    if argmax is not None:
        OH, OW = yv.shape[2], yv.shape[3]
        ih, iw = idx // Wp - pad, idx % Wp - pad
        oh = torch.arange(OH).view(1, 1, OH, 1) * stride - pad
        ow = torch.arange(OW).view(1, 1, 1, OW) * stride - pad
        code = (ih - oh) * k + (iw - ow)
        argmax[:N * OH * OW * C].view(N, OH, OW, C).copy_(code.permute(0, 2, 3, 1).to(torch.uint8))


def maxpool_bwd(dy, argmax, dx, k, stride, pad, accumulate, scratch):
    g = _nchw(dy)
    N, C, OH, OW = g.shape
    code = argmax[:N * OH * OW * C].view(N, OH, OW, C).permute(0, 3, 1, 2).long()
    oh = torch.arange(OH).view(1, 1, OH, 1) * stride - pad
    ow = torch.arange(OW).view(1, 1, 1, OW) * stride - pad
    ih, iw = oh + code // k, ow + code % k
    out = torch.zeros(N, C, dx.H * dx.W)
    out.scatter_add_(2, (ih * dx.W + iw).reshape(N, C, -1), g.reshape(N, C, -1))
    _store(dx, out.view(N, C, dx.H, dx.W), accumulate=bool(accumulate))


def upsample_fwd(x, y, stride):
    _store(y, _nchw(x).repeat_interleave(stride, 2).repeat_interleave(stride, 3))


def upsample_bwd(dy, dx, stride, accumulate):
    g = _nchw(dy)
    N, C = g.shape[:2]
    _store(dx, g.view(N, C, dx.H, stride, dx.W, stride).sum((3, 5)), accumulate=bool(accumulate))


def slice_copy(x, y, accumulate=False):
    _store(y, _nchw(x), accumulate=accumulate)


def slice_add(a, b, y):
    _store(y, _nchw(a) + _nchw(b))


def f32_to_view(x, M, C, scale, y, cpad, scale_dev=None):
    t = _t(y).reshape(-1, y.C) if y.ld == y.C else None
    sc = scale * (float(scale_dev.reshape(-1)[0]) if scale_dev is not None else 1.0)
    full = torch.zeros(M, cpad)
    full[:, :C] = x.view(-1)[:M * C].view(M, C) * sc
    _t(y)[..., :cpad].copy_(full.view(y.N, y.H, y.W, cpad).to(y.buf.dtype))


def zero_view(y, dummy):
    _t(y).zero_()


def bias_grad(dlogits, M, C, scale, gbias, scale_dev=None, deterministic=False):
    sc = scale * (float(scale_dev.reshape(-1)[0]) if scale_dev is not None else 1.0)
    gbias += sc * dlogits.view(-1)[:M * C].view(M, C).sum(0)


def yolo_decode(logits, B, G, A, C, anchors_wh, img_size, out, rows_total, row_offset):
    x = logits.view(B, G, G, A * (7 + C)).permute(0, 3, 1, 2)
    anchors = [(a[0], a[1], 0.0, 1.0) for a in anchors_wh]
    d = yolo_layer_ref.decode(x, anchors, C, img_size)
    out[:, row_offset:row_offset + A * G * G] = d['output']


def yolo_loss(logits, B, G, A, C, targets, anchors, img_size, ignore_thresh, use_giou, workspace, metrics, dlogits):
    with torch.enable_grad():
        x = logits.view(B, G, G, A * (7 + C)).permute(0, 3, 1, 2).clone().requires_grad_(True)
        _, loss, met = yolo_layer_ref.head_forward(x, targets, anchors, C, ignore_thresh, img_size, bool(use_giou))
        loss.sum().backward()
    dlogits.view(B, G, G, A * (7 + C)).copy_(x.grad.permute(0, 2, 3, 1))
    from tests.golden.make_golden import METRIC_KEYS
    metrics[:18] = torch.tensor([met[k] for k in METRIC_KEYS])
    metrics[18:] = 0


NAMES = ['check_device_tensor', 'nchw_to_nhwc', 'pack_weights_into', 'make_pack_table', 'pack_weights_multi',
         'make_reduce_table', 'wgrad_reduce_multi', 'conv_bn_act_eval', 'conv_igemm', 'conv_dgrad_bn_sums', 'conv_wgrad', 'wgrad_reduce',
         'bn_finalize', 'bn_eval_affine', 'bn_act_fwd', 'bn_act_fwd_fused', 'bn_act_bwd_apply_fused', 'bn_act_bwd_reduce', 'bn_bwd_finalize', 'bn_act_bwd_apply',
         'maxpool_argmax_bytes', 'maxpool_fwd', 'maxpool_bwd', 'upsample_fwd', 'upsample_bwd', 'slice_copy', 'slice_add', 'f32_to_view',
         'zero_view', 'bias_grad', 'bias_grad_det', 'fold_rows', 'yolo_decode', 'yolo_loss']


def install(monkeypatch):
    g = globals()
    for n in NAMES:
        monkeypatch.setattr(real, n, g[n])
