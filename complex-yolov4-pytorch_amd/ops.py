"""Python-side operator layer: one function per C-ABI entry point (include/cyolo_hip.h), taking torch
device tensors / NHWC views and launching on torch's current HIP stream.  PyTorch is used for device
memory and streams only; every computation below happens in libcyolo_hip.so.  No CPU fallback."""
import ctypes
import threading

import torch

from ._lib import CyoloError, lib

CY_F16, CY_BF16, CY_F32 = 0, 1, 2
ACT = {'linear': 0, 'leaky': 1, 'mish': 2}
CONV_STATS, CONV_BIAS_F32OUT, CONV_ACCUM, CONV_TRANSPOSED, CONV_STATS_DET = 1, 2, 4, 8, 32
_TORCH_DT = {CY_F16: torch.float16, CY_BF16: torch.bfloat16, CY_F32: torch.float32}
_ELSIZE = {CY_F16: 2, CY_BF16: 2, CY_F32: 4}
DTYPE_NAME = {CY_F16: 'f16', CY_BF16: 'bf16', CY_F32: 'f32'}


class LaunchProfiler:
    """Brackets selected kernel launches with HIP events on the launch stream (torch's current stream) and keeps
    (kind, algorithmic flops, algorithmic bytes, start, end) so bench.py can report per-kernel achieved rates."""

    def __init__(self):
        self.records = []

    def bracket(self, kind, flops, nbytes):
        return _Bracket(self, kind, flops, nbytes)

    def bracket_overhead_ms(self, n=64):
        """Median duration of an EMPTY bracket (event record -> event record with nothing in between) on the current
        stream: what the two event records add to every measured launch."""
        pairs = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        t = sorted(s.elapsed_time(e) for s, e in pairs)
        return t[len(t) // 2]

    def summary(self):
        torch.cuda.synchronize()
        over = self.bracket_overhead_ms()
        out = {}
        for kind, flops, nbytes, s, e in self.records:
            d = out.setdefault(kind, dict(launches=0, ms=0.0, ms_raw=0.0, flops=0.0, bytes=0.0))
            d['launches'] += 1
            t = s.elapsed_time(e)
            d['ms_raw'] += t
            d['ms'] += max(t - over, 0.0)
            d['flops'] += flops
            d['bytes'] += nbytes
        return out


class _Bracket:
    def __init__(self, prof, kind, flops, nbytes):
        self.prof, self.kind, self.flops, self.nbytes = prof, kind, flops, nbytes

    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True)
        self.e = torch.cuda.Event(enable_timing=True)
        self.s.record()

    def __exit__(self, *a):
        self.e.record()
        self.prof.records.append((self.kind, self.flops, self.nbytes, self.s, self.e))


class _NoBracket:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOBRACKET = _NoBracket()
PROFILER = None


def prof(kind, flops=0.0, nbytes=0.0):
    """Context manager around a launch; a no-op unless ops.PROFILER is set (bench.py's roofline leg)."""
    return _NOBRACKET if PROFILER is None else PROFILER.bracket(kind, flops, nbytes)


def dtype_code(name):
    """'f16' / 'bf16' / 'f32' (or the codes themselves) -> CY_F16 / CY_BF16 / CY_F32."""
    if name in (CY_F16, 'f16', 'fp16', 'half', torch.float16):
        return CY_F16
    if name in (CY_BF16, 'bf16', 'bfloat16', torch.bfloat16):
        return CY_BF16
    if name in (CY_F32, 'f32', 'fp32', 'float', torch.float32):
        return CY_F32
    raise ValueError('unsupported dtype %r' % (name,))


def torch_dtype(code):
    return _TORCH_DT[code]


def chunk(code):
    return 4 if code == CY_F32 else 8


_TLS = threading.local()


class stream_scope:
    """Pins the HIP stream the wrappers launch on for the enclosed calls (thread-local).  torch.cuda.current_stream() costs
    several microseconds of Python per query; the engine issues ~1200 launches per step on streams it already knows
    (its pass's stream, or its side stream), so it states them once per section instead of asking per launch."""

    def __init__(self, stream):
        self.handle = ctypes.c_void_p(stream.cuda_stream)

    def __enter__(self):
        self.prev = getattr(_TLS, 'handle', None)
        _TLS.handle = self.handle
        return self

    def __exit__(self, *exc):
        _TLS.handle = self.prev
        return False


class Event:
    """A hipEvent owned by the library (cy_event_create): the fork / join primitive of the two-stream backward, issued through
    the C ABI so that it can be part of a recorded launch list (torch.cuda.Event records cannot)."""

    def __init__(self):
        h = ctypes.c_void_p()
        lib().call('cy_event_create', ctypes.c_void_p(ctypes.addressof(h)))
        self.handle = h

    def __del__(self):
        try:
            lib().raw('cy_event_destroy')(self.handle)
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


def stream_handle(stream):
    return ctypes.c_void_p(stream.cuda_stream)


def event_record(ev, handle):
    lib().call('cy_event_record', ev.handle, handle)


def stream_wait_event(handle, ev):
    lib().call('cy_stream_wait_event', handle, ev.handle)


def start_recording():
    """Every operator call from here to stop_recording() is executed AND recorded.  -> the recorder (``.py(fn)`` inserts a host
    hook).  One recording at a time per process."""
    from ._lib import PlanRecorder
    L = lib()
    if L.recorder is not None:
        raise CyoloError('a launch list is already being recorded')
    L.recorder = PlanRecorder(L)
    return L.recorder


def stop_recording(keep=True):
    """-> the recorded Program (None with keep=False: a pass that raised)."""
    L = lib()
    rec, L.recorder = L.recorder, None
    return rec.finish() if (keep and rec is not None) else None


def recording():
    return lib().recorder


def _stream():
    h = getattr(_TLS, 'handle', None)
    return h if h is not None else ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if isinstance(t, View):
        return ctypes.c_void_p(t.ptr)
    assert t.is_cuda and t.is_contiguous(), 'device pointer arguments must be contiguous CUDA tensors'
    return ctypes.c_void_p(t.data_ptr())


_GPU_SEEN = False


def _require_gpu():
    global _GPU_SEEN
    if _GPU_SEEN:
        return
    if not torch.cuda.is_available():
        raise CyoloError('no HIP device: the hot path has no CPU fallback')
    _GPU_SEEN = True


def check_device_tensor(t, who):
    """The hot path has no CPU fallback: a host tensor is an error, not a slow path."""
    if not t.is_cuda:
        raise CyoloError('%s runs on the HIP device only (no CPU fallback); got a %s tensor' % (who, t.device))


class View:
    """NHWC view (N,H,W,C) with channel stride ld over a flat device buffer, starting `off` elements in."""
    __slots__ = ('buf', 'off', 'N', 'H', 'W', 'C', 'ld', 'dt')

    def __init__(self, buf, off, N, H, W, C, ld, dt):
        self.buf, self.off, self.N, self.H, self.W, self.C, self.ld, self.dt = buf, off, N, H, W, C, ld, dt

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.off * _ELSIZE[self.dt]

    @property
    def M(self):
        return self.N * self.H * self.W

    def channels(self, c0, c):
        return View(self.buf, self.off + c0, self.N, self.H, self.W, c, self.ld, self.dt)

    def to_nchw(self):
        """Materialise as an NCHW float32 torch tensor (test/debug helper)."""
        flat = self.buf.view(-1)[self.off:self.off + (self.M - 1) * self.ld + self.C]
        t = torch.as_strided(flat, (self.N, self.H, self.W, self.C), (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1))
        return t.permute(0, 3, 1, 2).float().contiguous()

    @staticmethod
    def alloc(N, H, W, C, dt, ld=None, device='cuda', zero=False):
        ld = ld or C
        mk = torch.zeros if zero else torch.empty
        return View(mk(N * H * W * ld, dtype=_TORCH_DT[dt], device=device), 0, N, H, W, C, ld, dt)

    @staticmethod
    def from_nchw(x, dt, cpad=None, ld=None):
        """NCHW float tensor -> NHWC view of dtype dt (host-side permute; test helper)."""
        N, C, H, W = x.shape
        cp = cpad or C
        ld = ld or cp
        buf = torch.zeros(N, H, W, ld, dtype=_TORCH_DT[dt], device=x.device)
        buf[..., :C] = x.permute(0, 2, 3, 1).to(_TORCH_DT[dt])
        return View(buf.view(-1), 0, N, H, W, cp, ld, dt)


# ---- conv stack ---------------------------------------------------------------------------------

def pack_weights(w, co_pad, ci_pad, dt, want_dgrad=True):
    """w: fp32 [Co,Ci,k,k] device tensor -> (wf [CoPad, k*k*CiPad], wd [CiPad, k*k*CoPad] or None)."""
    _require_gpu()
    Co, Ci, ks, _ = w.shape
    wf = torch.empty(co_pad, ks * ks * ci_pad, dtype=_TORCH_DT[dt], device=w.device)
    wd = torch.empty(ci_pad, ks * ks * co_pad, dtype=_TORCH_DT[dt], device=w.device) if want_dgrad else None
    lib().call('cy_pack_weights', _p(w.detach().contiguous()), Co, Ci, ks, co_pad, ci_pad, dt, _p(wf), _p(wd), _stream())
    return wf, wd


def pack_weights_into(w, co_pad, ci_pad, dt, wf, wd):
    Co, Ci, ks, _ = w.shape
    lib().call('cy_pack_weights', _p(w), Co, Ci, ks, co_pad, ci_pad, dt, _p(wf), _p(wd), _stream())


MULTI_ELEMS = 1024


def _block_table(counts, per_block=MULTI_ELEMS):
    """[(descriptor index, element count)] -> int32 [nblocks, 2] of (descriptor, first element / 256)."""
    rows = []
    for i, n in enumerate(counts):
        for first in range(0, n, per_block):
            rows.append((i, first // 256))
    return rows


def make_pack_table(items, device):
    """items: [(w fp32 [Co,Ci,k,k], wf, wd or None, CoPad, CiPad)] -> (desc bytes tensor, block table tensor, keepalive)."""
    import struct
    raw, counts = b'', []
    for item in items:
        w, wf, wd, cop, cip = item[:5]
        wd_ld = int(item[5]) if len(item) > 5 else 0      # row stride of a wider dgrad matrix this layer fills a column range of
        Co, Ci, ks, _ = w.shape
        raw += struct.pack('<QQQiiiiii', w.data_ptr(), wf.data_ptr(), wd.data_ptr() if wd is not None else 0, Co, Ci, ks, cop, cip, wd_ld)
        assert ks <= 3
        counts.append(((cop + 63) // 64) * ((cip + 63) // 64) * MULTI_ELEMS)   # one block per 64 x 64 (co, ci) tile
    desc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    blocks = torch.tensor(_block_table(counts), dtype=torch.int32, device=device)
    return desc, blocks


def pack_weights_multi(desc, blocks, dt):
    lib().call('cy_pack_weights_multi', _p(desc), _p(blocks), blocks.shape[0], dt, _stream())


def make_reduce_table(items, device):
    """items: [(part tensor, grad tensor, split, CoRows, CiPad, ks, Co, Ci)]."""
    import struct
    raw, rows = b'', []
    for i, item in enumerate(items):
        part, grad, split, corows, cip, ks, Co, Ci = item[:8]
        flags = int(item[8]) if len(item) > 8 else 0      # bit 0: zero the slab elements after reading (atomic wgrad mode)
        # the fold's unit is one (co, ci) pair with all its taps.  Layers with few pairs and many slabs (the first
        # stages: split-K up to 128) let 2-8 threads share a pair, each folding every lanes-th slab
        lanes = 1
        if ks in (1, 3) and Co * Ci < 65536:
            while lanes < 8 and split >= 16 * lanes:
                lanes *= 2
        raw += struct.pack('<QQiiiiiiii', part.data_ptr(), grad.data_ptr(), split, corows, cip, ks, Co, Ci, lanes, flags)
        pb = 256 // lanes if ks in (1, 3) else 256
        rows += [(i, first // 32) for first in range(0, Co * Ci, pb)]
    desc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    blocks = torch.tensor(rows, dtype=torch.int32, device=device)
    return desc, blocks


def wgrad_reduce_multi(desc, blocks, scale, accumulate):
    lib().call('cy_wgrad_reduce_multi', _p(desc), _p(blocks), blocks.shape[0], float(scale), int(accumulate), _stream())


def make_adam_table(items, device):
    """items: [(p, g, m, v, group index)] fp32 contiguous device tensors of equal numel."""
    import struct
    raw, counts = b'', []
    for p_, g, m, v, grp in items:
        raw += struct.pack('<QQQQqii', p_.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p_.numel(), grp, 0)
        counts.append(p_.numel())
    desc = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    blocks = torch.tensor(_block_table(counts), dtype=torch.int32, device=device)
    return desc, blocks


def sgd_multi(desc, blocks, momentum, nesterov, first_step, lrs, wds, zero_grad=False, skip_flag=None):
    lib().call('cy_sgd_multi', _p(desc), _p(blocks), blocks.shape[0], float(momentum), int(nesterov), int(first_step),
               int(zero_grad), _farr(lrs), _farr(wds), len(lrs), _p(skip_flag), _stream())


def grad_nonfinite(flat_grad, flag):
    """flag (int32 device tensor [1]) <- 1 if flat_grad holds an inf / nan, else 0."""
    lib().call('cy_grad_nonfinite', _p(flat_grad), flat_grad.numel(), _p(flag), _stream())


def adam_multi(desc, blocks, beta1, beta2, eps, bc1, bc2, lrs, wds, zero_grad=False, skip_flag=None):
    lib().call('cy_adam_multi', _p(desc), _p(blocks), blocks.shape[0], float(beta1), float(beta2), float(eps), float(bc1),
               float(bc2), int(zero_grad), _farr(lrs), _farr(wds), len(lrs), _p(skip_flag), _stream())


def adam_multi_dev(desc, blocks, beta1, beta2, eps, step_in, step_out, lrs, wds, zero_grad=False, skip_flag=None):
    """cy_adam_multi_dev: bias corrections from the device step counter (step_in -> step_out, int32 [1] each)."""
    lib().call('cy_adam_multi_dev', _p(desc), _p(blocks), blocks.shape[0], float(beta1), float(beta2), float(eps), _p(step_in),
               _p(step_out), int(zero_grad), _farr(lrs), _farr(wds), len(lrs), _p(skip_flag), _stream())


def adam_multi_graph(desc, blocks, beta1, beta2, eps, step_counter, group_lr_wd, zero_grad=False, skip_flag=None):
    """cy_adam_multi_graph: step counter and (lr x 8, wd x 8) on the device -- the capturable form."""
    lib().call('cy_adam_multi_graph', _p(desc), _p(blocks), blocks.shape[0], float(beta1), float(beta2), float(eps),
               _p(step_counter), _p(group_lr_wd), int(zero_grad), _p(skip_flag), _stream())


def nchw_to_nhwc(x, cpad, dt, out=None):
    _require_gpu()
    N, C, H, W = x.shape
    out = out or View.alloc(N, H, W, cpad, dt, device=x.device)
    lib().call('cy_nchw_to_nhwc', _p(x.contiguous()), N, C, H, W, cpad, dt, _p(out), _stream())
    return out


def pipe_launches():
    """Kernel launches since load that ran on the 8-wave pipelined conv kernel (cy_pipe_launches)."""
    return int(lib().raw('cy_pipe_launches')())


def direct_launches():
    """Kernel launches since load that ran on the direct small-Cin 3x3 kernel (cy_direct_launches)."""
    return int(lib().raw('cy_direct_launches')())


def conv_pipe_config(mode=1, cap=0, bn=0, variant=0, bm_eff=0):
    """Tuning / A-B switch of the conv dispatch (cy_conv_pipe_config): mode 0 never / 1 policy / 2 whenever possible."""
    lib().call('cy_conv_pipe_config', int(mode), int(cap), int(bn), int(variant), int(bm_eff))


def conv_slab_config(mode=1, bm_eff=0):
    """Test / tool switch of the slab kernels (cy_conv_slab_config): mode 0 off, 1 on hints 11-13, + 2 the loader / compute variant,
    + 4 / + 8 the 3- / 4-stage weight ring of the K-split kernel; bm_eff forces the tile's used pixels."""
    lib().call('cy_conv_slab_config', int(mode), int(bm_eff))


def conv_stats_rows(M, OC, det=False):
    return lib().raw('cy_conv_stats_rows_det' if det else 'cy_conv_stats_rows')(M, OC)


def bn_scratch_rows():
    """Extra rows every BN partial table needs behind it (see cy_bn_scratch_rows)."""
    return lib().raw('cy_bn_scratch_rows')()


CONV_TILE_SHIFT = 8
CONV_TILE_HINTS = (1, 2, 3, 4, 5, 6, 7, 8, 9)
CONV_SLAB_HINTS = (12, 13)     # the slab kernel (3x3 / stride 1 / pad 1, > 64 output channels): 192 / 256-pixel tile (11: its own policy)   # 1: 4-wave kernels; 2-5: pipelined kernel, 128/192/256/384-pixel tile; 6: its own
#                                                policy; 7-9: its loader/compute split with a 128/192/256-pixel tile


def conv_igemm(g, w, wrows, out, ks, stride, pad, flags=0, bias=None, stats=None, tile=0):
    """Forward conv (or dgrad with CONV_TRANSPOSED).  g/out: Views; w: packed weight tensor; tile: kernel / tile hint."""
    _require_gpu()
    lib().call('cy_conv_igemm', _p(g), g.N, g.H, g.W, g.C, g.ld, _p(w), wrows, _p(out), out.H, out.W, out.C, out.ld, ks,
               stride, pad, g.dt, flags | (tile << CONV_TILE_SHIFT), _p(bias), _p(stats), None, _stream())


def conv_dgrad_bn_sums(g, w, wrows, out, ks, stride, pad, raw, mean, invstd, scale, shift, act, sums, flags=0, tile=0):
    """dgrad whose epilogue also adds the BN-backward sums of the layer that produced ``out``'s tensor into ``sums``
    (cy_conv_dgrad_bn_sums; raises CyoloError where the pipelined kernel does not apply)."""
    _require_gpu()
    lib().call('cy_conv_dgrad_bn_sums', _p(g), g.N, g.H, g.W, g.C, g.ld, _p(w), wrows, _p(out), out.H, out.W, out.C, out.ld,
               ks, stride, pad, g.dt, flags | (tile << CONV_TILE_SHIFT), _p(raw), raw.ld, _p(mean), _p(invstd), _p(scale),
               _p(shift), act, _p(sums), _stream())


def conv_bn_act_eval(g, w, wrows, out, ks, stride, pad, scale, shift, act, res=None, tile=0):
    """Eval-mode conv + BN affine + activation (+ shortcut) in one kernel."""
    _require_gpu()
    lib().call('cy_conv_bn_act_eval', _p(g), g.N, g.H, g.W, g.C, g.ld, _p(w), wrows, _p(out), out.H, out.W, out.C, out.ld, ks,
               stride, pad, g.dt, _p(scale), _p(shift), act, _p(res), res.ld if res is not None else 0,
               tile << CONV_TILE_SHIFT, _stream())


ERR_UNSUPPORTED = -3


def conv_bn_act_train(g, w, wrows, raw, out, res, ks, stride, pad, bins, gamma, beta, rmean, rvar, nbt, momentum, eps, vec,
                      zero_table, act, ticket, tile=0):
    """Training-mode conv + BatchNorm (batch statistics) + activation (+ shortcut) in ONE two-phase launch
    (cy_conv_bn_act_train).  -> False when this launch shape is not taken (grid not co-resident, kernel does not apply): the
    caller keeps conv_igemm + bn_act_fwd_fused.  rmean / rvar / nbt None: running statistics left alone (tuning launches)."""
    _require_gpu()
    L = lib()
    args = (_p(g), g.N, g.H, g.W, g.C, g.ld, _p(w), wrows, _p(raw), raw.H, raw.W, raw.C, raw.ld, _p(out), out.ld, _p(res),
            res.ld if res is not None else 0, ks, stride, pad, g.dt, tile << CONV_TILE_SHIFT, _p(bins), _p(gamma), _p(beta), _p(rmean),
            _p(rvar), _p(nbt), float(momentum), float(eps), _p(vec), _p(zero_table), zero_table.numel() if zero_table is not None else 0,
            act, _p(ticket), _stream())
    if L.recorder is not None:
        L.call('cy_conv_bn_act_train', *args)        # (a recorded pass only holds launches that were accepted before)
        return True
    rc = L.raw('cy_conv_bn_act_train')(*args)
    if rc == ERR_UNSUPPORTED:
        return False
    if rc != 0:
        raise CyoloError('cy_conv_bn_act_train failed with status %d' % rc)
    return True


def wgrad_split(M, Co, Ci, ks):
    return lib().raw('cy_conv_wgrad_split')(M, Co, Ci, ks)


def conv_wgrad(dy, x, ks, stride, pad, part, split, use_tr=1, atomic=False, tile64=False):
    """atomic: every split adds into slab 0 (pre-zeroed) with fp32 atomics instead of writing its own slab; tile64: tiles of at
    most 64 x 64 (a quarter of the split for the same number of blocks)."""
    _require_gpu()
    lib().call('cy_conv_wgrad', _p(dy), dy.N, dy.H, dy.W, dy.C, dy.ld, _p(x), x.H, x.W, x.C, x.ld, ks, stride, pad,
               dy.dt, _p(part), split, use_tr | (4 if atomic else 0) | (8 if tile64 else 0), _stream())


def conv_wgrad_bn(g, raw, x, ks, stride, pad, mean, invstd, scale, shift, bins, rows, ggamma, gbeta, gscale, zero_table, act, part, split):
    """BatchNorm backward + weight gradient of a conv without an input gradient in one kernel (cy_conv_wgrad_bn): the pre-BN
    gradient is never written.  Raises CyoloError(CY_ERR_UNSUPPORTED) where the kernel does not apply."""
    _require_gpu()
    lib().call('cy_conv_wgrad_bn', _p(g), g.N, g.H, g.W, g.C, g.ld, _p(raw), raw.ld, _p(x), x.H, x.W, x.C, x.ld, ks, stride, pad,
               g.dt, _p(mean), _p(invstd), _p(scale), _p(shift), _p(bins), rows, _p(ggamma), _p(gbeta), float(gscale), _p(zero_table),
               zero_table.numel() if zero_table is not None else 0, act, _p(part), split, _stream())


def conv1x1_bn_in(x, in_scale, in_shift, act_in, act_out, wf, cout, out, flags=0, stats=None):
    """cy_conv1x1_bn_in: the 1x1 conv `out = act_in(x * in_scale + in_shift) (*) wf` reading the producer's PRE-BatchNorm view x,
    writing the activated rows to act_out on the way.  Raises CyoloError(CY_ERR_UNSUPPORTED) for other shapes / dtypes."""
    _require_gpu()
    lib().call('cy_conv1x1_bn_in', _p(x), x.M, x.C, x.ld, _p(in_scale), _p(in_shift), act_in, _p(act_out), act_out.ld, _p(wf),
               wf.shape[0], _p(out), cout, out.ld, x.dt, flags, _p(stats), _stream())


def wgrad_reduce(part, split, co_rows, ci_pad, ks, Co, Ci, scale, accumulate, grad):
    lib().call('cy_wgrad_reduce', _p(part), split, co_rows, ci_pad, ks, Co, Ci, float(scale), int(accumulate), _p(grad),
               _stream())


def bn_finalize(stats, rows, C, count, gamma, beta, rmean, rvar, nbt, momentum, eps, mean, invstd, scale, shift):
    lib().call('cy_bn_finalize', _p(stats), rows, C, count, _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(nbt),
               float(momentum), float(eps), _p(mean), _p(invstd), _p(scale), _p(shift), _stream())


def bn_eval_affine(gamma, beta, rmean, rvar, eps, scale, shift):
    lib().call('cy_bn_eval_affine', _p(gamma), _p(beta), _p(rmean), _p(rvar), gamma.numel(), float(eps), _p(scale),
               _p(shift), _stream())


def bn_act_fwd(x, y, res, scale, shift, act):
    lib().call('cy_bn_act_fwd', _p(x), x.ld, _p(y), y.ld, _p(res), res.ld if res is not None else 0, x.M, x.C, _p(scale),
               _p(shift), act, x.dt, _stream())


def bn_act_fwd_fused(x, y, res, bins, rows, gamma, beta, rmean, rvar, nbt, momentum, eps, vec, zero_table, act, stats_ld=0, stats_c0=0,
                     vec_ld=0):
    """cy_bn_finalize + cy_bn_act_fwd in one launch; vec: float32 [4, C] (mean, invstd, scale, shift) written by the kernel;
    zero_table: the OTHER statistics table of the alternating pair (zeroed by this launch)."""
    lib().call('cy_bn_act_fwd_fused', _p(x), x.ld, _p(y), y.ld, _p(res), res.ld if res is not None else 0, x.M, x.C, _p(bins), rows,
               _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(nbt), float(momentum), float(eps), _p(vec[0] if vec_ld else vec), _p(zero_table),
               zero_table.numel() if zero_table is not None else 0, act, x.dt, int(stats_ld), int(stats_c0), int(vec_ld), _stream())


def bn_act_bwd_apply_fused(x, dy, dx, res_grad, res_accum, mean, invstd, scale, shift, bins, rows, ggamma, gbeta, gscale,
                           zero_table, act, bins_ld=0, bins_c0=0):
    """cy_bn_bwd_finalize + cy_bn_act_bwd_apply in one launch."""
    lib().call('cy_bn_act_bwd_apply_fused', _p(x), x.ld, _p(dy), dy.ld, _p(dx), dx.ld, _p(res_grad),
               res_grad.ld if res_grad is not None else 0, int(res_accum), x.M, x.C, _p(mean), _p(invstd), _p(scale), _p(shift),
               _p(bins), rows, _p(ggamma), _p(gbeta), float(gscale), _p(zero_table),
               zero_table.numel() if zero_table is not None else 0, act, x.dt, int(bins_ld), int(bins_c0), _stream())


def fold_rows_out(rows):
    return lib().raw('cy_fold_rows_out')(rows)


def fold_rows(bins, rows, W, out):
    """Deterministic mode: first stage of the two-stage fold; -> rows of ``out`` to hand to the finaliser."""
    ro = fold_rows_out(rows)
    lib().call('cy_fold_rows', _p(bins), rows, W, _p(out), ro, _stream())
    return ro


def bias_grad_det(dlogits, M, C, scale, gbias, scratch, scale_dev=None):
    lib().call('cy_bias_grad_det', _p(dlogits), M, C, float(scale), _p(scale_dev), _p(gbias), _p(scratch), _stream())


def bn_bwd_rows(M, C, dt, det=False):
    return lib().raw('cy_bn_bwd_rows_det' if det else 'cy_bn_bwd_rows')(M, C, dt)


def bn_act_bwd_reduce(x, dy, mean, invstd, scale, shift, act, part, rows=None):
    rows = bn_bwd_rows(x.M, x.C, x.dt) if rows is None else rows
    lib().call('cy_bn_act_bwd_reduce', _p(x), x.ld, _p(dy), dy.ld, x.M, x.C, _p(mean), _p(invstd), _p(scale), _p(shift),
               act, x.dt, _p(part), rows, _stream())


def bn_bwd_finalize(part, rows, C, dgs, dbs, ggamma, gbeta, gscale):
    lib().call('cy_bn_bwd_finalize', _p(part), rows, C, _p(dgs), _p(dbs), _p(ggamma), _p(gbeta), float(gscale), _stream())


def bn_act_bwd_apply(x, dy, dx, res_grad, res_accum, mean, invstd, scale, shift, dgs, dbs, act):
    lib().call('cy_bn_act_bwd_apply', _p(x), x.ld, _p(dy), dy.ld, _p(dx), dx.ld, _p(res_grad),
               res_grad.ld if res_grad is not None else 0, int(res_accum), x.M, x.C, _p(mean), _p(invstd), _p(scale),
               _p(shift), _p(dgs), _p(dbs), act, x.dt, _stream())


def maxpool_argmax_bytes(N, H, OH, OW, C):
    return int(lib().raw('cy_maxpool_argmax_bytes')(N, H, OH, OW, C))


def maxpool_fwd(x, y, k, stride, pad, argmax, scratch):
    """scratch: any device tensor with >= N*H*OW*C elements of x's dtype worth of bytes (row-pass maxima)."""
    lib().call('cy_maxpool_fwd', _p(x), x.N, x.H, x.W, x.C, x.ld, _p(y), y.H, y.W, y.ld, k, stride, pad, _p(argmax),
               _p(scratch), x.dt, _stream())


def maxpool_bwd(dy, argmax, dx, k, stride, pad, accumulate, scratch):
    lib().call('cy_maxpool_bwd', _p(dy), dy.N, dy.H, dy.W, dy.C, dy.ld, _p(argmax), _p(dx), dx.H, dx.W, dx.ld, k, stride,
               pad, int(accumulate), _p(scratch), dy.dt, _stream())


def upsample_fwd(x, y, stride):
    lib().call('cy_upsample_fwd', _p(x), x.N, x.H, x.W, x.C, x.ld, _p(y), y.ld, stride, x.dt, _stream())


def upsample_bwd(dy, dx, stride, accumulate):
    lib().call('cy_upsample_bwd', _p(dy), dx.N, dx.H, dx.W, dx.C, dy.ld, _p(dx), dx.ld, stride, int(accumulate), dx.dt,
               _stream())


def slice_copy(x, y, accumulate=False):
    lib().call('cy_slice_copy', _p(x), x.ld, _p(y), y.ld, x.M, x.C, int(accumulate), x.dt, _stream())


def slice_add(a, b, y):
    lib().call('cy_slice_add', _p(a), a.ld, _p(b), b.ld, _p(y), y.ld, a.M, a.C, a.dt, _stream())


def f32_to_view(x, M, C, scale, y, cpad, scale_dev=None):
    lib().call('cy_f32_to_view', _p(x), M, C, float(scale), _p(scale_dev), _p(y), y.ld, cpad, y.dt, _stream())


def zero_view(y, dummy):
    """Zero-fill a channel-slice view (f32_to_view with zero source channels)."""
    lib().call('cy_f32_to_view', _p(dummy), y.M, 0, 0.0, None, _p(y), y.ld, y.C, y.dt, _stream())


def bias_grad(dlogits, M, C, scale, gbias, scale_dev=None, deterministic=False):
    lib().call('cy_bias_grad', _p(dlogits), M, C, float(scale), _p(scale_dev), _p(gbias), int(deterministic), _stream())


# ---- YOLO head ------------------------------------------------------------------------------------

def _farr(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


_CONST_FARR = {}


def _const_farr(vals):
    """A host float array that lives as long as the process (per-model constants such as anchors): a launch list recorded with
    its address (cy_run_plan) stays valid."""
    key = tuple(float(v) for v in vals)
    a = _CONST_FARR.get(key)
    if a is None:
        a = _CONST_FARR[key] = _farr(key)
    return a


def yolo_decode(logits, B, G, A, C, anchors_wh, img_size, out, rows_total, row_offset):
    flat = [v for a in anchors_wh for v in a[:2]]
    lib().call('cy_yolo_decode', _p(logits), B, G, A, C, _const_farr(flat), float(img_size), _p(out), rows_total, row_offset,
               _stream())


def head_scratch_bytes():
    """Private-segment bytes per lane of the per-target loss kernels (cy_head_scratch_bytes); 0 = safe beside other kernels."""
    return int(lib().raw('cy_head_scratch_bytes')())


def yolo_loss_workspace(B, G, A, C, nT):
    return lib().raw('cy_yolo_loss_workspace')(B, G, A, C, nT)


def yolo_loss(logits, B, G, A, C, targets, anchors, img_size, ignore_thresh, use_giou, workspace, metrics, dlogits):
    nT = 0 if targets is None else targets.shape[0]
    flat = [v for a in anchors for v in a[:4]]
    lib().call('cy_yolo_loss', _p(logits), B, G, A, C, _p(targets) if nT else None, nT, _const_farr(flat), float(img_size),
               float(ignore_thresh), int(bool(use_giou)), _p(workspace), _p(metrics), _p(dlogits), _stream())


class _HeadIn(ctypes.Structure):
    _fields_ = [('logits', ctypes.c_void_p), ('dlogits', ctypes.c_void_p), ('metrics', ctypes.c_void_p), ('anchors', ctypes.c_void_p),
                ('G', ctypes.c_int), ('row_offset', ctypes.c_int)]


def yolo_loss_multi_workspace(Gs, B, A, C, nT):
    return int(lib().raw('cy_yolo_loss_multi_workspace')(len(Gs), (ctypes.c_int * len(Gs))(*Gs), B, A, C, nT))


def make_head_table(heads):
    """heads: [(logits, dlogits, metrics, anchors [(w, h, im, re)] * A, G, row_offset)] -> the host table cy_yolo_loss_multi takes
    (keeps the anchor arrays alive)."""
    arr = (_HeadIn * len(heads))()
    keep = []
    for i, (logits, dlogits, metrics, anchors, G, row_offset) in enumerate(heads):
        fa = _farr([v for a in anchors for v in a[:4]])
        keep.append(fa)
        arr[i].logits, arr[i].dlogits, arr[i].metrics = logits.data_ptr(), dlogits.data_ptr(), metrics.data_ptr()
        arr[i].anchors = ctypes.cast(fa, ctypes.c_void_p).value
        arr[i].G, arr[i].row_offset = int(G), int(row_offset)
    return arr, keep


def yolo_loss_multi(table, nheads, B, A, C, targets, img_size, ignore_thresh, use_giou, workspace, out, rows_total, cap=None, nt_dev=None):
    """Decode (out is not None) + loss of all heads in one sequence of launches (cy_yolo_loss_multi).  With ``cap`` / ``nt_dev``
    (cy_yolo_loss_multi_n): ``targets`` is a buffer of at least ``cap`` rows, the int32 device word ``nt_dev`` holds how many of
    them are this batch's."""
    if nt_dev is not None:
        lib().call('cy_yolo_loss_multi_n', nheads, ctypes.cast(table[0], ctypes.c_void_p), B, A, C, _p(targets), int(cap), _p(nt_dev),
                   float(img_size), float(ignore_thresh), int(bool(use_giou)), _p(workspace), _p(out), rows_total, _stream())
        return
    nT = 0 if targets is None else targets.shape[0]
    lib().call('cy_yolo_loss_multi', nheads, ctypes.cast(table[0], ctypes.c_void_p), B, A, C, _p(targets) if nT else None, nT,
               float(img_size), float(ignore_thresh), int(bool(use_giou)), _p(workspace), _p(out), rows_total, _stream())


# ---- geometry / NMS ---------------------------------------------------------------------------------

def riou_pairs(pred, target, giou):
    _require_gpu()
    n = pred.shape[0]
    dev = pred.device
    ious = torch.empty(n, device=dev)
    terms = torch.empty(n, device=dev)
    g = torch.empty(n, 6, device=dev)
    lib().call('cy_riou_pairs', _p(pred.float().contiguous()), _p(target.float().contiguous()), n, int(bool(giou)),
               _p(ious), _p(terms), _p(g), _stream())
    return ious, terms, g


def riou_anchors(anchors_wlir, targets_wlir):
    _require_gpu()
    nA, nT = anchors_wlir.shape[0], targets_wlir.shape[0]
    out = torch.empty(nA, nT, device=targets_wlir.device)
    lib().call('cy_riou_anchors', _p(anchors_wlir.float().contiguous()), nA, _p(targets_wlir.float().contiguous()), nT,
               _p(out), _stream())
    return out


def riou_matrix(a, b, eps=1e-16):
    _require_gpu()
    out = torch.empty(a.shape[0], b.shape[0], device=a.device)
    lib().call('cy_riou_matrix', _p(a.float().contiguous()), a.shape[0], _p(b.float().contiguous()), b.shape[0],
               float(eps), _p(out), _stream())
    return out


def rnms_greedy(boxes, confs, nms_thresh):
    _require_gpu()
    K = boxes.shape[0]
    dev = boxes.device
    ws = torch.empty(max(1, lib().raw('cy_rnms_workspace')(1, max(K, 1))), dtype=torch.uint8, device=dev)
    keep = torch.empty(max(K, 1), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    lib().call('cy_rnms_greedy', _p(boxes.float().contiguous()), _p(confs.float().contiguous()), K, float(nms_thresh),
               _p(ws), _p(keep), _p(count), _stream())
    return keep[:int(count.item())]


def pp2(pred, conf_thresh, nms_thresh):
    """post_processing_v2 on device.  pred [B,N,7+C] device fp32 -> (det [B,Kmax,9], src [B,Kmax], counts [B]) on host
    sizes; returns python lists per image (device tensors) and the source-row indices."""
    _require_gpu()
    B, N, NCH = pred.shape
    C = NCH - 7
    dev = pred.device
    pred = pred.float().contiguous()
    ws1 = torch.empty(B * N * 8, dtype=torch.uint8, device=dev)
    cand = torch.empty(B, N, dtype=torch.int32, device=dev)
    ccount = torch.zeros(B, dtype=torch.int32, device=dev)
    lib().call('cy_pp2_select', _p(pred), B, N, C, float(conf_thresh), _p(ws1), _p(cand), _p(ccount), _stream())
    counts = ccount.cpu()
    kmax = int(counts.max().item())
    if kmax == 0:
        return [None] * B, [None] * B
    kmax = (kmax + 63) // 64 * 64
    ws2 = torch.empty(lib().raw('cy_rnms_workspace')(B, kmax), dtype=torch.uint8, device=dev)
    det = torch.empty(B, kmax, 9, device=dev)
    src = torch.empty(B, kmax, dtype=torch.int32, device=dev)
    dcount = torch.zeros(B, dtype=torch.int32, device=dev)
    lib().call('cy_pp2_merge', _p(pred), B, N, C, _p(cand), _p(ccount), kmax, float(nms_thresh), _p(ws2), _p(det),
               _p(src), _p(dcount), _stream())
    dc = dcount.cpu()
    outs, srcs = [], []
    for b in range(B):
        n = int(dc[b])
        outs.append(det[b, :n] if n else None)
        srcs.append(src[b, :n] if n else None)
    return outs, srcs


def bev_workspace(H, W, device='cuda'):
    """Zeroed workspace for bev_rasterize (the kernel leaves it zeroed, so it can be reused frame after frame)."""
    return torch.zeros(int(lib().raw('cy_bev_workspace')(H, W)), dtype=torch.uint8, device=device)


def bev_rasterize(points, bounds, disc, H, W, workspace, out=None, zshift=None, max_height=None):
    """points: float32 [n,4] device tensor (x, y, z, intensity); bounds = (minX, maxX, minY, maxY, minZ, maxZ).
    zshift / max_height default to minZ / |maxZ - minZ| (raw points).  -> float32 [3,H,W] (intensity, height, density)."""
    _require_gpu()
    check_device_tensor(points, 'points')
    pts = points.contiguous()
    assert pts.dtype == torch.float32 and pts.dim() == 2 and pts.shape[1] == 4
    if pts.data_ptr() % 16:      # the kernel reads float4 points (a row slice of a larger tensor is only 4-byte aligned... 16 B per row keeps it aligned, a column offset does not)
        pts = pts.clone()
    if out is None:
        out = torch.empty(3, H, W, dtype=torch.float32, device=pts.device)
    zshift = bounds[4] if zshift is None else zshift
    max_height = abs(bounds[5] - bounds[4]) if max_height is None else max_height
    lib().call('cy_bev_rasterize', _p(pts) if pts.shape[0] else None, pts.shape[0], *[float(v) for v in bounds], float(zshift),
               float(max_height), float(disc), H, W, _p(workspace), _p(out), _stream())
    return out


# ---- augmentation on the rasterised maps ---------------------------------------------------------------

def _iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def bev_mosaic(tiles, img_size, rects, fill=0.5, out=None):
    """tiles: four float32 [C,h,w] device tensors; rects[k] = (x1a, y1a, x2a, y2a, x1b, y1b) -> [C, 2*img_size, 2*img_size]."""
    _require_gpu()
    t = [x.float().contiguous() for x in tiles]
    C, h, w = t[0].shape
    if out is None:
        out = torch.empty(C, 2 * img_size, 2 * img_size, dtype=torch.float32, device=t[0].device)
    lib().call('cy_bev_mosaic', _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), C, h, w, img_size, _iarr([v for r in rects for v in r]),
               float(fill), _p(out), _stream())
    return out


def bev_mosaic_targets(targets, tile_of, h, w, pads, img_size):
    """targets float32 [nT,8] (in place), tile_of int32 [nT] (0..3), pads[k] = (padw, padh)."""
    if targets.shape[0]:
        lib().call('cy_bev_mosaic_targets', _p(targets), targets.shape[0], _p(tile_of), h, w, _iarr([v for q in pads for v in q]),
                   img_size, _stream())
    return targets


def bev_flip_cutout(img, flip, holes, fill, targets=None, want_keep=False):
    """img float32 [C,H,W] -> new tensor: flipped along W (flip) with the holes (y1, y2, x1, x2) filled; targets updated in
    place; returns (out, keep mask uint8 [nT] or None)."""
    _require_gpu()
    C, H, W = img.shape
    out = torch.empty_like(img)
    nT = 0 if targets is None else targets.shape[0]
    keep = torch.ones(nT, dtype=torch.uint8, device=img.device) if (want_keep and nT) else None
    lib().call('cy_bev_flip_cutout', _p(img), C, H, W, int(bool(flip)), _iarr([v for q in holes for v in q]) if holes else None,
               len(holes), float(fill), _p(out), _p(targets) if nT else None, nT, _p(keep), _stream())
    return out, keep
