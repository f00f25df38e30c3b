"""Executes a ``graph.Plan`` on the HIP operator layer (``ops``): owns the device storages for one
(batch, height, width, dtype, mode) and issues the kernels of a forward / backward pass in plan order on
torch's current HIP stream.  Reference call stack replaced: Darknet.forward (darknet2pytorch.py:162-230)
with its per-block ATen calls, and the autograd graph ``total_loss.backward()`` walks (train.py:212).

Memory (288 GB HBM3E): every activation, its raw (pre-BN) twin and its gradient are resident for the
whole step -- nothing is recomputed, nothing is freed between steps.
"""
import contextlib
import os

import torch

from .. import ops, tune
from ..ops import CONV_ACCUM, CONV_BIAS_F32OUT, CONV_STATS, CONV_TRANSPOSED, CY_F32, View

BN_EPS, BN_MOMENTUM = 1e-5, 0.1          # torch.nn.BatchNorm2d defaults, as the reference uses them


def _pad32(c):
    return (c + 31) // 32 * 32


def _wgrad_rows(rec):
    """Rows of a weight-gradient slab = channels of the gradient view cy_conv_wgrad is handed: the real channel count of a
    BatchNorm conv's output (complex_yolov3_tiny.cfg's first layer has 16), the 32-padded staging tensor of a head conv."""
    return rec['cout'] if rec['bn'] else _pad32(rec['cout'])


_PW_DIRECT = ((64, 64), (128, 64), (64, 128), (64, 32), (32, 64))    # (Cin, Cout) of the 1x1 streaming kernel (conv_direct.hip)
def _slab_shape(rec):
    """3x3 / stride 1 / pad 1: what the slab kernel of conv_pipe.hip takes (forward and dgrad alike)."""
    return rec['ks'] == 3 and rec['stride'] == 1 and rec['pad'] == 1


_CONV_TUNE_MEMO = {}      # (kind, shape key) -> best kernel / tile hint, shared by every engine of the process
# tools/make_tune_cache.py only: lets the process that PRODUCES the persisted table time deterministic engines too (their
# statistics-table layout differs, so they have keys of their own); everywhere else deterministic=True never times
_DET_TIMING = os.environ.get('CY_TUNE_DET_TIMING') == '1'


class Arena:
    """Allocates an engine's device buffers.  With ``guard`` > 0 (tests/test_gpu_redzone.py; CY_GUARD_BYTES) every buffer sits
    between two bands of ``guard`` bytes of 0xFF -- NaN in every floating-point type the kernels read -- inside its own
    allocation: ``violations()`` names the buffers whose bands a kernel wrote into, and an over-READ drags NaNs into results
    that the test compares bit for bit with an unguarded run."""

    def __init__(self, device, guard=0):
        self.device, self.guard, self.blocks = device, int(guard), []

    def new(self, name, shape, dtype, zero=False):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        if not self.guard:
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        g = self.guard
        raw = torch.full((g + nbytes + (-nbytes % 256) + g,), 0xFF, dtype=torch.uint8, device=self.device)
        mid = raw[g:g + nbytes].view(dtype).view(shape)
        if zero:
            mid.zero_()
        self.blocks.append((name, raw, nbytes))
        return mid

    def violations(self):
        """[(buffer name, 'below' | 'above', first damaged byte's distance from the buffer)] -- empty when every band is intact."""
        bad, g = [], self.guard
        for name, raw, nbytes in self.blocks:
            lo, hi = raw[:g] != 0xFF, raw[g + nbytes:] != 0xFF
            if bool(lo.any()):
                bad.append((name, 'below', g - int(torch.nonzero(lo)[-1])))
            if bool(hi.any()):
                bad.append((name, 'above', int(torch.nonzero(hi)[0]) + 1))
        return bad


class Engine:
    GUARD_BYTES = int(os.environ.get('CY_GUARD_BYTES', '0'))      # red zones around every buffer (tests only; see Arena)

    def __init__(self, plan, N, dt, device, training, deterministic=False):
        self.plan, self.N, self.dt, self.device, self.training = plan, N, dt, device, training
        self.arena = Arena(device, self.GUARD_BYTES if getattr(device, 'type', str(device)) == 'cuda' else 0)
        new = self.arena.new
        # deterministic: every reduction that the default mode runs through fp32 atomics into shared bins (BatchNorm batch
        # statistics in the conv epilogue, BatchNorm backward sums) gets one table row per contributing block and is folded
        # in a fixed order -- two runs of the same step are bit-identical.  Costs larger tables and slower folds.
        self.det = bool(deterministic)
        self._fwd_tile, self._dgrad_tile, self._fwd_tuned, self._dgrad_tuned = {}, {}, False, False
        # conv idx -> tile hint of its two-phase conv + BN + activation launch (ops.conv_bn_act_train): the layers whose grid
        # is one co-resident round.  CY_CONV_BN_FUSED: 0 (default) never, 1 timed per layer against conv + the separate BN /
        # activation pass, 2 wherever the kernel takes the launch.  OFF by default because it is slower on the MI355X: measured
        # per layer (round 4, profiles/r04_two_phase_conv.txt) the grid-wide wait costs 8-18 us where the separate pass costs
        # 6-14 us -- e.g. 256->256 3x3 @38x38: 28.9 + 10.2 us in two launches, 47.4 us fused; the timed mode picks it for none
        # of complex_yolov4.cfg's 85 eligible layers, forcing it everywhere costs 5 % of the step
        self._fwd_fused = {}
        self.conv_bn_fused = int(os.environ.get('CY_CONV_BN_FUSED', '0'))
        self._ticket = None
        # BatchNorm-backward sums taken in the epilogue of the dgrad that last writes a layer's output gradient
        # (ops.conv_dgrad_bn_sums) instead of a separate pass over (raw, gradient).  The plan marks where that is possible
        # (graph.py::_mark_dgrad_bn_sums); the first backward times fused against separate per layer.  CY_DGRAD_BN_SUMS:
        # 0 never, 1 (default) timed, 2 every marked layer untimed (what the CPU operator simulator's tests use).
        self.dgrad_bn_sums = int(os.environ.get('CY_DGRAD_BN_SUMS', '1'))
        self._dgrad_sums, self._sums_fused = {}, set()      # (P idx, run c0) -> (L record, tile hint);  {L idx}
        self.tdt = ops.torch_dtype(dt)
        self.act, self.gact = {}, {}
        # Sibling 1x1 convs over ONE input (the two branches of every CSP stage, reference cfg complex_yolov4.cfg:44-64,124,212,421,
        # 635) as one launch each for the forward conv, the weight gradient and the input gradient (round 6): lead conv idx ->
        # pair record, follower idx -> lead idx.  Training engines of the default 16-bit mode only; see _find_siblings.
        self._sib, self._sib_follow = {}, {}
        self._views, self._view_refs = {}, []
        sf = os.environ.get('CY_SIBLING_FUSE', '1')      # 0: off; 1: the 16-bit modes; 2: the fp32 default mode too (tests: the CPU simulator)
        if (training and (dt != CY_F32 or sf == '2') and not deterministic and hasattr(ops, 'bn_act_fwd_fused')
                and os.environ.get('CY_FUSED_BN', '1') != '0' and sf != '0'):
            self._find_siblings(plan)
        # BatchNorm-backward sums of the TWO producers of a CSP concatenation in the epilogue of the closing conv's dgrad (round 6;
        # graph.py 'dx_sums_cat'): conv idx of that closing conv P -> record; member layer idx -> (record, first column of its
        # channels in the record's tables).  CY_CAT_SUMS: 0 off, 1 (default) timed per stage against dgrad + two reduce passes,
        # 2 untimed (the CPU simulator's tests).  See _find_cats.
        self._cat, self._cat_of, self._cat_on = {}, {}, {}
        self.cat_sums = int(os.environ.get('CY_CAT_SUMS', '1'))
        if (training and self.cat_sums and self.dgrad_bn_sums and not deterministic and hasattr(ops, 'bn_act_fwd_fused')
                and hasattr(ops, 'conv_dgrad_bn_sums') and os.environ.get('CY_FUSED_BN', '1') != '0'
                and (dt != CY_F32 or getattr(device, 'type', str(device)) != 'cuda')):
            self._find_cats(plan)
        sib_raw = {r['raw'].st.sid for pr in self._sib.values() for r in (pr['a'], pr['b'])}
        sib_raw |= {L['raw'].st.sid for ct in self._cat.values() for L in (ct['L1'], ct['L2'])}
        max_raw = 0
        for st in plan.storages:
            n = N * st.H * st.W * st.C
            if st.kind == 'logits':
                self.act[st.sid] = new('act %r' % (st,), n, torch.float32)
            elif st.kind == 'raw':
                if training:
                    if st.sid not in sib_raw:      # (a sibling pair's two pre-BN tensors are slices of ONE joint buffer, below)
                        self.act[st.sid] = new('raw %r' % (st,), n, self.tdt)
                else:
                    max_raw = max(max_raw, n)
            else:
                self.act[st.sid] = new('act %r' % (st,), n, self.tdt)
                if training and st.kind == 'act':
                    self.gact[st.sid] = new('gact %r' % (st,), n, self.tdt)
        self.raw_scratch = new('raw_scratch', max(max_raw, 1), self.tdt)
        for ct in self._cat.values():
            # the pre-BN tensors of the concatenation's two producers side by side: [L1 | L2], or [L1 | A | B] when L2 is the lead A of
            # a sibling pair (whose joint conv then writes columns C1 ... of a 3-wide buffer).  Each layer's own view has the joint
            # row stride; the closing conv's dgrad epilogue reads columns 0 ... C1 + C2 as ONE pre-BN tensor
            L1, L2, C1, C2 = ct['L1'], ct['L2'], ct['C1'], ct['C2']
            pr = self._sib.get(L2['idx'])
            wide = C1 + C2 + (pr['Cb'] if pr is not None else 0)
            H, W = L1['H'], L1['W']
            buf = new('raw concat %d+%d' % (L1['idx'], L2['idx']), N * H * W * wide, self.tdt)
            ct['raw'] = View(buf, 0, N, H, W, C1 + C2, wide, dt)
            members = [(L1, 0, C1), (L2, C1, C2)]
            if pr is not None:
                members.append((pr['b'], C1 + C2, pr['Cb']))
                pr['raw'] = (View(buf, C1, N, H, W, C2, wide, dt), View(buf, C1 + C2, N, H, W, pr['Cb'], wide, dt),
                             View(buf, C1, N, H, W, C2 + pr['Cb'], wide, dt))
            for r, c0, C in members:
                self._views[(id(r['raw']), False)] = View(buf, c0, N, H, W, C, wide, dt)
                self._view_refs.append(r['raw'])
        for lead, pr in self._sib.items():
            A, B, Ca, Cb = pr['a'], pr['b'], pr['Ca'], pr['Cb']
            CJ, H, W = Ca + Cb, A['H'], A['W']
            for nm in ('raw', 'draw'):      # [A | B]: the conv's output / the two BatchNorm backwards' output (dY of the fused wgrad / dgrad)
                if nm in pr:
                    continue                # (the pre-BN buffer is part of a concatenation's, above)
                buf = new('%s siblings %d+%d' % (nm, A['idx'], B['idx']), N * H * W * CJ, self.tdt)
                pr[nm] = (View(buf, 0, N, H, W, Ca, CJ, dt), View(buf, Ca, N, H, W, Cb, CJ, dt), View(buf, 0, N, H, W, CJ, CJ, dt))
            for r, v in ((A, pr['raw'][0]), (B, pr['raw'][1])):
                self._views[(id(r['raw']), False)] = v
                self._view_refs.append(r['raw'])
            pr['rec'] = dict(A, cout=CJ)          # the joint conv as the launch helpers see it (A's index, Ca + Cb output channels)
            pr['rec'].pop('_work', None)
            pr['rec'].pop('_names', None)
        # per-conv persistent BN vectors and packed weights; shared scratch for the reductions
        self.bnvec, self.wf, self.wd = {}, {}, {}
        for ct in self._cat.values():
            # (mean, invstd, scale, shift) of the two producers as ONE [4][C1 + C2] block: each layer's own vectors are column
            # ranges of it (cy_bn_act_fwd_fused writes them with vec_ld = C1 + C2)
            ct['vec'] = new('bnvec concat %d+%d' % (ct['L1']['idx'], ct['L2']['idx']), (4, ct['C1'] + ct['C2']), torch.float32)
            self.bnvec[ct['L1']['idx']] = ct['vec'][:, :ct['C1']]
            self.bnvec[ct['L2']['idx']] = ct['vec'][:, ct['C1']:]
        max_stats = max_bnrows = max_c = 1
        max_wpart = 0
        self.wsplit, self.wslab_off, self.wsplit_cap = {}, {}, {}
        self.watomic = {}          # conv idx -> split-K factor of its ATOMIC weight-gradient launch (all splits add into ONE slab)
        self.wtile64 = set()       # conv idx whose weight gradient runs on 64 x 64 tiles (a quarter of the split-K slabs)
        self._wgrad_tuned = False
        for rec in plan.convs:
            C, M = rec['cout'], N * rec['H'] * rec['W']
            cop, cip = _pad32(C), rec['cin_pad']
            kk = rec['ks'] * rec['ks']
            pr = self._sib.get(rec['idx'])
            if pr is not None:
                # one forward matrix [Ca + Cb][Cin] (A's rows, then B's) and one dgrad matrix [Cin][Ca + Cb] (each module packs its
                # own rows / columns: cy_pack_desc.wd_ld); self.wd[lead] is the JOINT matrix the fused dgrad multiplies
                CJ = pr['Ca'] + pr['Cb']
                pr['wf'] = new('wf siblings %d' % rec['idx'], (CJ, cip), self.tdt)
                pr['wd'] = new('wd siblings %d' % rec['idx'], (cip, CJ), self.tdt)
                self.wf[rec['idx']], self.wf[pr['b']['idx']] = pr['wf'][:pr['Ca']], pr['wf'][pr['Ca']:]
                self.wd[rec['idx']], self.wd[pr['b']['idx']] = pr['wd'], pr['wd'][:, pr['Ca']:]
                max_stats = max(max_stats, ops.conv_stats_rows(M, CJ, self.det) * 2 * CJ)
            elif rec['idx'] not in self._sib_follow:
                self.wf[rec['idx']] = new('wf[%d]' % rec['idx'], (cop, kk * cip), self.tdt)
                self.wd[rec['idx']] = new('wd[%d]' % rec['idx'], (cip, kk * cop), self.tdt) if training and not rec['first'] else None
            if rec['bn']:
                if rec['idx'] not in self.bnvec:
                    self.bnvec[rec['idx']] = new('bnvec[%d]' % rec['idx'], (4, C), torch.float32)
                max_stats = max(max_stats, ops.conv_stats_rows(M, C, self.det) * 2 * C)
                max_c = max(max_c, C)
                if training:
                    max_bnrows = max(max_bnrows, ops.bn_bwd_rows(M, C, dt, self.det) * 2 * C)
            if training and rec['idx'] in self._sib_follow:
                # its weight gradient is the lower rows of the lead's joint slab
                self.wsplit[rec['idx']], self.wsplit_cap[rec['idx']], self.wslab_off[rec['idx']] = 0, 0, max_wpart
            elif training:
                wr = _wgrad_rows(rec) if pr is None else pr['Ca'] + pr['Cb']
                sp = ops.wgrad_split(M, wr, cip, rec['ks'])
                # room for the split autotuner (first backward) to move away from the heuristic's choice
                cap = max(1, min((M + 511) // 512, max(2 * sp, sp + 4), 128))
                while cap > sp and cap * wr * kk * cip * 4 > (512 << 20):
                    cap -= 1
                self.wsplit[rec['idx']] = sp
                self.wsplit_cap[rec['idx']] = max(cap, sp)
                self.wslab_off[rec['idx']] = max_wpart
                max_wpart += self.wsplit_cap[rec['idx']] * wr * kk * cip
        # binned-atomics tables: zero once, every finaliser leaves its table zeroed for the next layer
        # (the first statistics table and both BN-backward tables share one allocation: a training step zeroes them with ONE
        # fill at the top of the forward pass -- nothing touches the backward tables before the backward pass)
        _r64 = lambda n: (n + 63) // 64 * 64
        cat_floats = sum(_r64(ops.conv_stats_rows(N * ct['L1']['H'] * ct['L1']['W'], ct['C1'] + ct['C2']) * 2 * (ct['C1'] + ct['C2']))
                         for ct in self._cat.values())
        self._ztab = new('ztab (stats + bn-backward tables)', _r64(max_stats) + 2 * _r64(max_bnrows) + cat_floats, torch.float32, zero=True)
        # (the concatenation sums tables live behind them: filled by the closing conv's dgrad, read by two BatchNorm backward passes
        # that may run much later -- not part of the alternating pair; zeroed with everything else at the top of the forward pass)
        off = _r64(max_stats) + 2 * _r64(max_bnrows)
        self._cat_tables = self._ztab[off:]
        for ct in self._cat.values():
            n = ops.conv_stats_rows(N * ct['L1']['H'] * ct['L1']['W'], ct['C1'] + ct['C2']) * 2 * (ct['C1'] + ct['C2'])
            ct['tbl'] = self._ztab[off:off + n]
            off += _r64(n)
        self.stats = self._ztab[:max_stats]
        self.bnpart = self._ztab[_r64(max_stats):_r64(max_stats) + max_bnrows]
        # default (non-deterministic) mode: the fold runs in the prologue of the consuming kernel (cy_bn_act_fwd_fused /
        # cy_bn_act_bwd_apply_fused), which cannot zero the table it reads -- two tables alternate from layer to layer and
        # every launch zeroes the other one
        self.fused_bn = not self.det and hasattr(ops, 'bn_act_fwd_fused') and os.environ.get('CY_FUSED_BN', '1') != '0'
        self.stats_pair = [self.stats, new('stats_pair[1]', max_stats, torch.float32, zero=True)] if self.fused_bn else None
        self.bnpart_pair = ([self.bnpart, self._ztab[_r64(max_stats) + _r64(max_bnrows):][:max_bnrows]]
                            if (self.fused_bn and training) else None)
        self._sp = self._bp = 0
        self._bn_tables_fwd = -1          # fwd_serial of the forward pass whose table fill also covered the backward tables
        self.dgs, self.dbs = new('dgs', max_c, torch.float32), new('dbs', max_c, torch.float32)
        # deterministic mode: second-stage table of the two-stage folds (<= 256 rows) / partial rows of the head bias gradient
        self.fold_tmp = new('fold_tmp', max(256 * 2 * max_c, 256 * 32), torch.float32, zero=True) if self.det else None
        # split-K slabs of every conv stay resident until their group is folded (table-driven, a few launches per step)
        self.wpart = new('wpart (split-K slabs)', max(max_wpart, 1), torch.float32)
        self._pack_table = self._pack_key = self._pack_epoch = None
        self._in_side_head = self._heads_on_side = False
        # the heads of a model as one batched sequence of launches (cy_yolo_loss_multi) when they share A, C and the threshold
        same = len({(h['A'], h['C'], h['ignore_thresh']) for h in plan.heads}) == 1
        self._multi_heads = (hasattr(ops, 'yolo_loss_multi') and same and 1 <= len(plan.heads) <= 3
                             and os.environ.get('CY_HEADS_MULTI', '1') != '0' and getattr(device, 'type', str(device)) == 'cuda')
        self._pending_heads, self._head_table = [], None
        # superseded head workspaces are kept for the engine's lifetime: a hipGraph captured while one of them was current
        # (graphed.GraphedTrainStep) has its pointer baked into its kernel arguments and keeps writing there on every replay
        self._retired_ws = []
        # ... unless nothing was ever captured: then only the recorded launch lists can still point at a superseded table, and
        # those are dropped together with the key they were recorded under -- the list is bounded to the last MAX_RETIRED
        # objects (work already queued on the two streams may still read the most recent ones).  graphed.GraphedTrainStep sets
        # pin_retired before its first capture.  (a caller whose parameter / gradient tensors move every epoch --
        # EMA swaps, load_state_dict(assign=True) -- used to leak a table per move for the engine's lifetime.)
        self.pin_retired = False
        # the first layer's BatchNorm backward inside its weight-gradient kernel (ops.conv_wgrad_bn; CY_FIRST_FUSED=0: two launches)
        self._first_fused = (training and getattr(device, 'type', str(device)) == 'cuda' and dt != CY_F32
                             and hasattr(ops, 'conv_wgrad_bn') and os.environ.get('CY_FIRST_FUSED', '1') != '0')
        self._wgrad_ev, self._main_stream, self._side_scope = {}, None, None
        self._main_h = self._side_h = None      # raw stream handles of the pass under way (ops.stream_handle)
        self._fork_ev = self._join_ev = None    # ops.Event: main -> side before a fold, side -> main at the end of backward
        # Recorded launch lists (ops.start_recording / cy_run_plan): once every kernel / tile / split choice is made the
        # calls of a pass are the same from step to step -- static storages, scalars on the device -- so the pass is recorded
        # once per key (target rows, parameter / gradient buffers, loss scale, stream) and re-issued from C in one call:
        # the host's share of a step drops from ~12 ms of Python + ctypes to ~2 ms.  Not a hipGraph: a replay IS the eager
        # launch sequence.  CY_PLAN_REPLAY=0 keeps every pass eager; bench.py's launch brackets (ops.PROFILER) and the
        # opt-in side-stream heads do too.
        self.replay = (os.environ.get('CY_PLAN_REPLAY', '1') != '0' and getattr(device, 'type', str(device)) == 'cuda'
                       and hasattr(ops, 'start_recording') and os.environ.get('CY_HEADS_SIDE', '0') != '1')
        self._fwd_progs, self._bwd_progs, self._fwd_key = {}, {}, None
        self.replayed = 0                        # passes issued through cy_run_plan (tests / probes)
        self.passes = 0                          # forward + backward passes issued in all (replayed / passes = the hit rate)
        self._on_module_done = None
        self._tg_buf = None                      # engine-owned copy of the target rows (a recorded list needs a fixed address)
        self._nt_dev = None                      # int32 device word: this batch's number of target rows
        self._gout_buf = new('gout', 1, torch.float32, zero=True) if training else None
        self.fwd_serial = 0
        self._reduce_groups = None
        use_side = training and getattr(device, 'type', str(device)) == 'cuda' and os.environ.get('CY_WGRAD_SIDE_STREAM', '1') != '0'
        self.side = torch.cuda.Stream(device=device) if use_side else None
        self.dummy = new('dummy', 16, torch.float32, zero=True)
        # pools
        self.argmax, self.pool_scratch = {}, None
        max_pool_in = 1
        for rec in plan.fwd:
            if rec['op'] == 'pool':
                o, x = rec['out'], rec['x']
                if training:
                    self.argmax[rec['idx']] = new('argmax[%d]' % rec['idx'], ops.maxpool_argmax_bytes(N, x.st.H, o.st.H, o.st.W, o.C), torch.uint8)
                # row-pass intermediate N x H x OW x C (forward: tensor dtype, backward: fp32)
                max_pool_in = max(max_pool_in, N * x.st.H * max(o.st.W, x.st.W) * x.C)
        self.pool_scratch = new('pool_scratch', max_pool_in, torch.float32)
        # heads
        self.outputs = new('outputs', (N, plan.rows_total, 7 + plan.heads[0]['C']), torch.float32) if plan.heads else None
        self.metrics = [new('metrics[%d]' % i, 20, torch.float32, zero=True) for i in range(len(plan.heads))]
        self.dlogits, self.head_tmp, self.loss_ws = [], [], []
        for h in plan.heads:
            n = N * h['G'] * h['G']
            self.dlogits.append(new('dlogits[%d]' % len(self.dlogits), n * h['A'] * (7 + h['C']), torch.float32) if training else None)
            self.head_tmp.append(View(new('head_tmp[%d]' % len(self.head_tmp), N * h['G'] * h['G'] * 32, self.tdt), 0, N, h['G'], h['G'], 32, 32, dt)
                                 if training else None)
            self.loss_ws.append(None)
        self.nT = -1

    def buffer_map(self):
        """[(name, device address, bytes)] of every device tensor this engine owns (activations, gradients, packed weights,
        tables, workspaces): what bench.py's supervisor maps a GPU memory-access fault address onto."""
        out, seen = [], set()

        def add(name, t):
            if isinstance(t, View):
                t = t.buf
            if isinstance(t, torch.Tensor) and t.is_cuda and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                out.append((name, t.data_ptr(), t.numel() * t.element_size()))
            elif isinstance(t, dict):
                for k, v in t.items():
                    add('%s[%s]' % (name, k), v)
            elif isinstance(t, (list, tuple)):
                for i, v in enumerate(t):
                    add('%s[%d]' % (name, i), v)
        sts = {st.sid: st for st in self.plan.storages}
        for sid, t in self.act.items():
            add('act %r' % (sts.get(sid),), t)
        for sid, t in self.gact.items():
            add('gact %r' % (sts.get(sid),), t)
        for name, t in vars(self).items():
            if name not in ('act', 'gact', 'plan', 'params', 'grads', '_views', '_view_refs'):
                add(name, t)
        return out

    # ---- views -----------------------------------------------------------------------------------
    def _find_siblings(self, plan):
        """Pairs of 1x1 / stride-1 BatchNorm convs (A, B) that read the SAME tensor and sit next to each other in the forward
        plan -- the `route -2` pattern of every CSP stage -- and whose backward ops are adjacent too (B's, then A's: B stores the
        input gradient, A accumulates onto it).  For such a pair the conv of A runs over the joint weight matrix and writes both
        pre-BN tensors ([A | B], one joint buffer), the two BatchNorm passes stay per module (separate parameters in the
        reference's state dict), the two BatchNorm backwards write their dRaw into one joint buffer, and ONE weight gradient
        and ONE input gradient (K = Ca + Cb, no fan-in read-add-store) follow."""
        fwd = plan.fwd
        bpos = {id(b['fwd']): i for i, b in enumerate(plan.bwd) if b['op'] == 'conv_bwd'}
        for i in range(len(fwd) - 1):
            A, B = fwd[i], fwd[i + 1]
            if A['op'] != 'conv' or B['op'] != 'conv' or not (A['bn'] and B['bn']) or A['first'] or B['first']:
                continue
            if (A['ks'], A['stride'], A['pad'], B['ks'], B['stride'], B['pad']) != (1, 1, 0, 1, 1, 0):
                continue
            xa, xb = A['x'], B['x']
            if xa.st is not xb.st or xa.c0 != xb.c0 or xa.C != xb.C or A['cout'] % 32 or B['cout'] % 32 or A['act'] != B['act']:
                continue
            if A['idx'] in self._sib_follow or A['idx'] in self._sib:
                continue
            ia, ib = bpos.get(id(A)), bpos.get(id(B))
            if ia is None or ib is None or ia != ib + 1:
                continue
            bA, bB = plan.bwd[ia], plan.bwd[ib]
            if len(bA['dx']) != 1 or len(bB['dx']) != 1:
                continue
            (ra, acc_a), (rb, acc_b) = bA['dx'][0], bB['dx'][0]
            if acc_b or not acc_a or ra.st is not xa.st or (ra.c0, ra.C, rb.c0, rb.C) != (xa.c0, xa.C, xa.c0, xa.C):
                continue
            self._sib[A['idx']] = dict(a=A, b=B, Ca=A['cout'], Cb=B['cout'])
            self._sib_follow[B['idx']] = A['idx']

    def _find_cats(self, plan):
        """The closing 1x1 conv P of a CSP stage reads [L1 | L2] (the B path's last conv and the sibling A: route layers=-1,-7) and its
        input gradient is the last writer of both layers' output gradients -- the plan marks such runs as 'dx_sums_cat'.  Taken here
        when neither layer is the FOLLOWER of a sibling pair and L1 is in no pair (the joint pre-BN buffer is then [L1 | L2] or
        [L1 | A | B])."""
        for b in plan.bwd:
            for ri, (L1, L2) in b.get('dx_sums_cat', {}).items():
                P = b['fwd']
                if (P['idx'] in self._sib_follow or P['idx'] in self._sib or L1['idx'] in self._sib or L1['idx'] in self._sib_follow
                        or L2['idx'] in self._sib_follow or P['idx'] in self._cat or ri != 0 or len(b['dx']) != 1):
                    continue
                if L1['idx'] in self._cat_of or L2['idx'] in self._cat_of or L1['cout'] % 32 or L2['cout'] % 32:
                    continue
                ct = dict(P=P, L1=L1, L2=L2, C1=L1['cout'], C2=L2['cout'], act=L1['act'])
                self._cat[P['idx']] = ct
                self._cat_of[L1['idx']] = (ct, 0)
                self._cat_of[L2['idx']] = (ct, L1['cout'])

    def _pair(self, idx):
        """-> (pair record, 0 for the lead | 1 for the follower) or (None, None)."""
        pr = self._sib.get(idx)
        if pr is not None:
            return pr, 0
        lead = self._sib_follow.get(idx)
        return (self._sib[lead], 1) if lead is not None else (None, None)

    def view(self, ref, grad=False):
        # views are static (fixed storages): built once per (tensor reference, forward / gradient)
        key = (id(ref), grad)
        v = self._views.get(key)
        if v is None:
            st = ref.st
            if st.kind == 'raw' and not self.training:
                buf = self.raw_scratch
            else:
                buf = (self.gact if grad else self.act)[st.sid]
            dt = CY_F32 if st.kind == 'logits' else self.dt
            v = self._views[key] = View(buf, ref.c0, self.N, st.H, st.W, ref.C, st.C, dt)
            self._view_refs.append(ref)     # keeps id(ref) unique for the life of the cache
        return v

    # ---- forward ---------------------------------------------------------------------------------
    def _scope(self):
        """ops.stream_scope on the pass's stream (device engines); a no-op context for the CPU operator simulator."""
        if getattr(self.device, 'type', str(self.device)) == 'cuda' and hasattr(ops, 'stream_scope'):
            return ops.stream_scope(torch.cuda.current_stream(self.device))
        return contextlib.nullcontext()

    def forward(self, x, targets, params, use_giou, img_size, weights_epoch=None):
        plan = self.plan
        self.fwd_serial += 1
        self.passes += 1
        self._pending_heads = []          # (a forward that raised after queueing a head must not leak it into this one)
        if self.stats_pair is not None and self.training:
            # the alternating statistics tables restart from a defined state every pass (complex_yolov4.cfg has an ODD number of
            # BatchNorm layers: the parity used to carry over from step to step, which a captured and replayed step cannot do)
            self._sp = 0
            (self._ztab if self.bnpart_pair is not None else self.stats_pair[0]).zero_()
            self._bn_tables_fwd = self.fwd_serial
        targets = self._own_targets(targets)
        with self._scope():
            ops.nchw_to_nhwc(x, plan.input.C, self.dt, out=self.view(plan.input))
            self.params = params
            key = prog = None
            can = (self.replay and self._fwd_tuned and ops.PROFILER is None and ops.recording() is None
                   and (not self.training or (self._wgrad_tuned and self._dgrad_tuned)) and (self.training or weights_epoch is not None))
            if can:
                key = self._fwd_key = self._forward_key(targets, params, use_giou, img_size, weights_epoch)
                prog = self._fwd_progs.get(key)
            else:
                self._fwd_key = None
            if prog is not None:
                if self.training:
                    self._sp = prog.sp_after
                prog.run()
                self.replayed += 1
            else:
                rec_on = can and (self.training or weights_epoch == self._pack_epoch)     # (eval: record a pass that packs nothing)
                if rec_on:
                    ops.start_recording()
                try:
                    self._pack_all(weights_epoch)
                    if not self._fwd_tuned:
                        self._autotune_fwd()
                    for rec in plan.fwd:
                        getattr(self, '_f_' + rec['op'])(rec, targets, use_giou, img_size)
                except BaseException:
                    if rec_on:
                        ops.stop_recording(keep=False)
                    raise
                if rec_on:
                    prog = ops.stop_recording()
                    prog.sp_after = self._sp
                    self._remember(self._fwd_progs, key, prog)
        if self._heads_on_side:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self._heads_on_side = False
        return self.outputs

    MAX_RETIRED = 8          # superseded tables / workspaces kept alive behind the current ones (see pin_retired)

    def _retire(self, obj):
        self._retired_ws.append(obj)
        if not self.pin_retired and len(self._retired_ws) > self.MAX_RETIRED:
            del self._retired_ws[:len(self._retired_ws) - self.MAX_RETIRED]

    def nbytes(self):
        """Device bytes this engine holds (activations, gradients, packed weights, slabs, tables): what Darknet's engine cache
        budgets (models/darknet2pytorch.py)."""
        return sum(b for _, _, b in self.buffer_map())

    MAX_PROGRAMS = 64        # recorded launch lists kept per direction (one per 64-row bucket of the target count, loss scale, ...: ~150 KB each)

    def _remember(self, table, key, prog):
        if len(table) >= self.MAX_PROGRAMS:      # (KITTI batches differ in their number of boxes: oldest recordings go first)
            table.pop(next(iter(table)))
        table[key] = prog

    @staticmethod
    def _rows_bucket(nT):
        """Row capacity the head launches are sized for: the batch's count rounded up to a multiple of 64 (one wave of targets)."""
        return max(64, (int(nT) + 63) // 64 * 64)

    def _own_targets(self, targets):
        """The caller's target rows copied into an engine-owned buffer (a recorded launch list needs them at a fixed address)
        and their count into a device word (``_nt_dev``): the batched heads (cy_yolo_loss_multi_n) are sized for the count's
        64-row bucket and read the live count on the device, so ONE recorded pass serves every batch of the bucket -- the
        reference's dataloader yields a different number of boxes almost every step.  -> the view of the buffer
        holding this batch's rows."""
        if targets is None:
            return None
        nT = int(targets.shape[0])
        cap = self._rows_bucket(nT)
        if self._tg_buf is None or self._tg_buf.shape[0] < cap or self._tg_buf.shape[1] != targets.shape[1]:
            if self._tg_buf is not None:
                self._retire(self._tg_buf)
            self._tg_buf = self.arena.new('targets', (2 * cap, targets.shape[1]), torch.float32, zero=True)
        if self._nt_dev is None:
            self._nt_dev = self.arena.new('live target rows', 1, torch.int32, zero=True)
        self._nt_dev.fill_(nT)          # (the scalar travels in the fill's kernel arguments: no host buffer to race with)
        own = self._tg_buf[:nT]
        if nT:
            own.copy_(targets, non_blocking=True)
        return own

    def _forward_key(self, targets, params, use_giou, img_size, weights_epoch):
        """Everything a recorded forward pass depends on besides the static storages: target-row count and buffer, the address
        of EVERY parameter and buffer the pass reads or writes (conv masters, biases, BatchNorm vectors, running statistics,
        num_batches_tracked: optimizers update in place, but .to() / load_state_dict(assign=True) / re-initialising one module
        moves tensors, and a replay would keep writing running statistics through the stale address -- ADVICE r4), loss
        variant, image size, stream.  One tuple of ~540 integers per forward: ~60 us."""
        pk = tuple(t.data_ptr() for t in params.values())
        rows = int(targets.shape[0]) if targets is not None else 0
        return (None if targets is None else ((self._rows_bucket(rows), 'bucket') if self._multi_heads else (rows, 'exact'), targets.data_ptr()),
                bool(use_giou), int(img_size), pk,
                torch.cuda.current_stream(self.device).cuda_stream, None if self.training else weights_epoch)

    def _pack_all(self, weights_epoch=None):
        """fp32 master weights -> packed f16/f32 matrices of every conv, one table-driven launch, on every forward.
        Only when the model vouches for its parameters (``weights_epoch`` = Darknet._weights_epoch under
        ``static_eval_weights``) an eval engine skips the pack while that counter stands still.  (Tensor version
        counters cannot be used: the fused optimizers write through raw pointers and ``p.data`` aliases restart at 0.)"""
        ws = [self.params['models.%d.conv%d.weight' % (r['idx'], r['n'])] for r in self.plan.convs]
        key = tuple(w.data_ptr() for w in ws)
        if self._pack_key != key:
            if self._pack_table is not None:
                # programs recorded against the old table can no longer be looked up (their key holds the old addresses) but
                # stay in the FIFO for a while: the table they point to must outlive them
                self._retire(self._pack_table)
                # every recorded pass is keyed by the parameter addresses: none of them can be looked up again
                self._fwd_progs.clear()
                self._bwd_progs.clear()
            items = []
            for w, r in zip(ws, self.plan.convs):
                pr = self._sib.get(r['idx']) or self._sib.get(self._sib_follow.get(r['idx']))
                if pr is None:
                    items.append((w, self.wf[r['idx']], self.wd[r['idx']], _pad32(r['cout']), r['cin_pad']))
                else:      # a sibling pair: its own rows of the joint forward matrix, its own COLUMNS of the joint dgrad matrix
                    c0 = 0 if r is pr['a'] else pr['Ca']
                    items.append((w, self.wf[r['idx']], pr['wd'][:, c0:c0 + r['cout']], r['cout'], r['cin_pad'], pr['Ca'] + pr['Cb']))
            self._pack_table = ops.make_pack_table(items, self.device)
            self._pack_key = key
            self._pack_epoch = None
        if not self.training and weights_epoch is not None:
            if weights_epoch == self._pack_epoch:
                return
            self._pack_epoch = weights_epoch
        else:
            self._pack_epoch = None
        ops.pack_weights_multi(self._pack_table[0], self._pack_table[1], self.dt)
        if not self.training:
            # eval: BatchNorm is an affine of the running statistics; its (scale, shift) vectors change exactly when the packed
            # weights do, so they are refreshed here (107 five-microsecond launches per forward otherwise: 4.5 % of the
            # batch-32 inference step)
            P = self.params
            for rec in self.plan.convs:
                if rec['bn']:
                    _, bname = self._names(rec)
                    vec = self.bnvec[rec['idx']]
                    ops.bn_eval_affine(P[bname + '.weight'], P[bname + '.bias'], P[bname + '.running_mean'],
                                       P[bname + '.running_var'], BN_EPS, vec[2], vec[3])

    def _build_reduce_groups(self):
        """Partition the convs (backward order) into fold groups, one launch per group.  A fold costs its slab bytes, and
        the LAST one runs after the main stream has nothing left to hide it behind (rocprofv3 timeline, round 2: a last
        group cut by gradient count held 0.68 ms of slabs, exposed before the optimizer) -- so the groups are cut by slab
        bytes (a quarter of the total each) and the final group only holds the last layers, <= 8 MB of slabs."""
        recs = [b['fwd'] for b in self.plan.bwd if b['op'] in ('conv_bwd', 'head_conv_bwd')]
        jrows = lambda r: (self._sib[r['idx']]['Ca'] + self._sib[r['idx']]['Cb']) if r['idx'] in self._sib else _wgrad_rows(r)
        nbytes = [4 * self.wsplit[r['idx']] * jrows(r) * r['ks'] * r['ks'] * r['cin_pad'] for r in recs]
        tail, acc = len(recs) - 1, nbytes[-1] if recs else 0
        while tail > 0 and acc + nbytes[tail - 1] <= (8 << 20):
            tail -= 1
            acc += nbytes[tail]
        target = max(1, sum(nbytes[:tail]) // 4)
        groups, cur, n = [], [], 0
        while 0 < tail < len(recs) and recs[tail - 1]['idx'] in self._sib_follow:      # (never cut between a follower and its lead)
            tail -= 1
        for i, rec in enumerate(recs[:tail]):
            cur.append(rec)
            n += nbytes[i]
            if n >= target and rec['idx'] not in self._sib_follow:
                groups.append(cur); cur, n = [], 0
        if cur:
            groups.append(cur)
        if recs[tail:]:
            groups.append(recs[tail:])
        if self._reduce_groups:
            self._retire(self._reduce_groups)      # (recorded backward passes hold the old tables' addresses ...)
            self._bwd_progs.clear()                # (... and are keyed by the gradient addresses that just changed)
        self._reduce_groups = []
        for g in groups:
            items = []
            for rec in g:
                idx = rec['idx']
                pr, role = self._pair(idx)
                if pr is not None:
                    # rows [0, Ca) / [Ca, Ca + Cb) of the LEAD's joint slabs (row stride Ca + Cb): one descriptor per module.  (The
                    # lead is the later of the two in backward order; the fold plan below keeps a pair inside one group.)
                    lead = pr['a']['idx']
                    cop, cip = pr['Ca'] + pr['Cb'], rec['cin_pad']
                    sp, off = self.wsplit[lead], self.wslab_off[lead]
                    r0 = 0 if role == 0 else pr['Ca']
                    part = self.wpart[off + r0 * cip:off + sp * cop * cip]
                    items.append((part, self.grads['models.%d.conv%d.weight' % (idx, rec['n'])], sp, cop, cip, 1,
                                  rec['cout'], rec['cin'], 1 if lead in self.watomic else 0))
                    continue
                cop, cip, kk = _wgrad_rows(rec), rec['cin_pad'], rec['ks'] * rec['ks']
                sp = self.wsplit[idx]
                off = self.wslab_off[idx]
                part = self.wpart[off:off + sp * cop * kk * cip]
                items.append((part, self.grads['models.%d.conv%d.weight' % (idx, rec['n'])], sp, cop, cip, rec['ks'],
                              rec['cout'], rec['cin'], 1 if idx in self.watomic else 0))
            desc, blocks = ops.make_reduce_table(items, self.device)
            self._reduce_groups.append(dict(last=g[-1]['idx'], mods=[r['idx'] for r in g], desc=desc, blocks=blocks))
        self._reduce_key = self._grads_key(self.grads)

    @staticmethod
    def _grads_key(grads):
        """Address of every gradient tensor the backward pass writes (fold tables, BatchNorm / bias gradients, recorded passes)."""
        return tuple(t.data_ptr() for t in grads.values())

    def _conv_work(self, rec):
        """(algorithmic flops, algorithmic bytes) of one pass of this conv: 2*M*Cout*k*k*Cin with the REAL channel
        counts, and input + output + weights each touched once in the storage dtype."""
        cw = rec.get('_work')
        if cw is None or cw[0] != (self.N, self.dt):
            M = self.N * rec['H'] * rec['W']
            kk = rec['ks'] * rec['ks']
            es = 4 if self.dt == ops.CY_F32 else 2
            flops = 2.0 * M * rec['cout'] * kk * rec['cin']
            nbytes = es * (self.N * rec['xH'] * rec['xW'] * rec['cin'] + M * rec['cout'] + rec['cout'] * kk * rec['cin'])
            cw = rec['_work'] = ((self.N, self.dt), (flops, nbytes))
        return cw[1]

    def _names(self, rec):
        nm = rec.get('_names')
        if nm is None:
            i, n = rec['idx'], rec['n']
            nm = rec['_names'] = ('models.%d.conv%d' % (i, n), 'models.%d.bn%d' % (i, n))
        return nm

    def _f_conv(self, rec, targets, use_giou, img_size):
        cname, bname = self._names(rec)
        P = self.params
        idx = rec['idx']
        cop = _pad32(rec['cout'])
        xv = self.view(rec['x'])
        if not rec['bn']:
            ops.conv_igemm(xv, self.wf[idx], cop, self.view(rec['out']), rec['ks'], rec['stride'], rec['pad'],
                           flags=CONV_BIAS_F32OUT, bias=P[cname + '.bias'])
            return
        raw = self.view(rec['raw'])
        vec = self.bnvec[idx]
        mean, invstd, scale, shift = vec[0], vec[1], vec[2], vec[3]
        C, M = rec['cout'], raw.M
        pr, role = self._pair(idx)
        vld = vec.stride(0) if idx in self._cat_of else 0      # (its vectors are columns of a concatenation's [4][C1 + C2] block)
        if pr is not None:
            # sibling pair: the lead's launch convolves with the joint matrix and leaves ONE statistics table of Ca + Cb channels;
            # each module's BatchNorm + activation pass reads its slice of it (and of the joint pre-BN buffer).  The tables
            # alternate once per PAIR: both passes read table _sp and zero the other one.
            tbl, other = self.stats_pair[self._sp], self.stats_pair[self._sp ^ 1]
            CJ = pr['Ca'] + pr['Cb']
            if role == 0:
                with ops.prof('igemm', *self._conv_work(pr['rec'])):
                    ops.conv_igemm(xv, pr['wf'], CJ, pr['raw'][2], 1, 1, 0, flags=CONV_STATS, stats=tbl, tile=self._fwd_tile.get(idx, 0))
            else:
                self._sp ^= 1
            res = self.view(rec['res']) if rec['res'] is not None else None
            ops.bn_act_fwd_fused(raw, self.view(rec['out']), res, tbl, ops.conv_stats_rows(M, CJ), P[bname + '.weight'],
                                 P[bname + '.bias'], P[bname + '.running_mean'], P[bname + '.running_var'],
                                 P.get(bname + '.num_batches_tracked'), BN_MOMENTUM, BN_EPS, vec, other, ops.ACT[rec['act']],
                                 stats_ld=CJ, stats_c0=0 if role == 0 else pr['Ca'], vec_ld=vld)
            return
        if self.training and self.fused_bn:
            tbl, other = self.stats_pair[self._sp], self.stats_pair[self._sp ^ 1]
            self._sp ^= 1
            fh = self._fwd_fused.get(idx) if not vld else None
            if fh is not None:
                res = self.view(rec['res']) if rec['res'] is not None else None
                with ops.prof('igemm', *self._conv_work(rec)):
                    ops.conv_bn_act_train(xv, self.wf[idx], cop, raw, self.view(rec['out']), res, rec['ks'], rec['stride'], rec['pad'],
                                          tbl, P[bname + '.weight'], P[bname + '.bias'], P[bname + '.running_mean'],
                                          P[bname + '.running_var'], P.get(bname + '.num_batches_tracked'), BN_MOMENTUM, BN_EPS, vec,
                                          other, ops.ACT[rec['act']], self._ticket, tile=fh)
                return
            with ops.prof('igemm', *self._conv_work(rec)):
                ops.conv_igemm(xv, self.wf[idx], cop, raw, rec['ks'], rec['stride'], rec['pad'], flags=CONV_STATS, stats=tbl,
                               tile=self._fwd_tile.get(idx, 0))
            res = self.view(rec['res']) if rec['res'] is not None else None
            ops.bn_act_fwd_fused(raw, self.view(rec['out']), res, tbl, ops.conv_stats_rows(M, C), P[bname + '.weight'],
                                 P[bname + '.bias'], P[bname + '.running_mean'], P[bname + '.running_var'],
                                 P.get(bname + '.num_batches_tracked'), BN_MOMENTUM, BN_EPS, vec, other, ops.ACT[rec['act']], vec_ld=vld)
            return
        if self.training:
            with ops.prof('igemm', *self._conv_work(rec)):
                ops.conv_igemm(xv, self.wf[idx], cop, raw, rec['ks'], rec['stride'], rec['pad'], flags=self._stat_flags,
                               stats=self.stats, tile=self._fwd_tile.get(idx, 0))
            tbl, rows = self.stats, ops.conv_stats_rows(M, C, self.det)
            if rows > 256:       # deterministic mode on a large grid: one row per pixel tile -> two-stage fold
                tbl, rows = self.fold_tmp, ops.fold_rows(self.stats, rows, 2 * C, self.fold_tmp)
            ops.bn_finalize(tbl, rows, C, M, P[bname + '.weight'], P[bname + '.bias'],
                            P[bname + '.running_mean'], P[bname + '.running_var'],
                            P.get(bname + '.num_batches_tracked'), BN_MOMENTUM, BN_EPS, mean, invstd, scale, shift)
        else:
            # eval: BN is an affine of the running statistics -> conv + BN + activation (+ shortcut) in one kernel,
            # the pre-BN tensor is never materialised
            res = self.view(rec['res']) if rec['res'] is not None else None     # (scale, shift): refreshed by _pack_all
            with ops.prof('igemm', *self._conv_work(rec)):
                ops.conv_bn_act_eval(xv, self.wf[idx], cop, self.view(rec['out']), rec['ks'], rec['stride'], rec['pad'],
                                     scale, shift, ops.ACT[rec['act']], res, tile=self._fwd_tile.get(idx, 0))
            return
        res = self.view(rec['res']) if rec['res'] is not None else None
        ops.bn_act_fwd(raw, self.view(rec['out']), res, scale, shift, ops.ACT[rec['act']])

    def _f_pool(self, rec, *_):
        ops.maxpool_fwd(self.view(rec['x']), self.view(rec['out']), rec['k'], rec['stride'], rec['pad'],
                        self.argmax.get(rec['idx']), self.pool_scratch)

    def _f_upsample(self, rec, *_):
        ops.upsample_fwd(self.view(rec['x']), self.view(rec['out']), rec['stride'])

    def _f_copy(self, rec, *_):
        ops.slice_copy(self.view(rec['x']), self.view(rec['out']))

    def _f_add(self, rec, *_):
        ops.slice_add(self.view(rec['a']), self.view(rec['b']), self.view(rec['out']))

    def _heads_side_ok(self):
        """The heads may run beside the trunk only while their per-target kernels have no private segment in the code
        object that is actually loaded (see cy_head_scratch_bytes); asked once."""
        ok = getattr(self, '_heads_ok', None)
        if ok is None:
            ok = self._heads_ok = (not hasattr(ops, 'head_scratch_bytes')) or ops.head_scratch_bytes() == 0
        return ok

    def _f_yolo(self, rec, targets, use_giou, img_size):
        if (self.side is not None and targets is not None and not self._in_side_head
                and os.environ.get('CY_HEADS_SIDE', '0') == '1' and self._heads_side_ok()):
            # OPT-IN (CY_HEADS_SIDE=1).  The decode + loss kernels of a head are a dozen one-wave launches, ~0.1 ms of latency that
            # nothing downstream needs before the loss is read; on the side stream they hide beside the trunk convs that follow
            # the head (+1 % in round 2, per-head launches).  Rounds 2-4 kept them off it because their results changed in lanes
            # 48-63 beside two conv instantiations; round 5 found the cause (SLP-packed float32 arithmetic beside MFMA waves,
            # profiles/r05_head_race.txt) and removed it at build time, so the option is safe again -- it stays an option because
            # the batched three-heads-in-one-sequence form (cy_yolo_loss_multi) runs at the LAST head, where no trunk conv is left
            # to hide behind, and because this path is issued through torch events (not part of a recorded launch list).
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.side.wait_event(ev)
            self._in_side_head = True
            try:
                with torch.cuda.stream(self.side), ops.stream_scope(self.side):
                    self._f_yolo(rec, targets, use_giou, img_size)
            finally:
                self._in_side_head = False
            self._heads_on_side = True
            return
        h = rec['head']
        logits = self.act[rec['logits'].st.sid]
        if (targets is not None and self._multi_heads and not self._in_side_head):
            # training: the heads' decode + loss are batched into ONE sequence of launches issued at the last head (head =
            # blockIdx.y: 9 launches per step instead of 8 per head -- their latency sits on the trunk's stream since round 3)
            self._pending_heads.append(rec)
            if len(self._pending_heads) == len(self.plan.heads):
                self._run_heads(targets, use_giou, img_size)
            return
        ops.yolo_decode(logits, self.N, rec['G'], rec['A'], rec['C'], rec['anchors'], img_size, self.outputs,
                        self.plan.rows_total, rec['row_offset'])
        if targets is None:
            return
        nT = targets.shape[0]
        need = ops.yolo_loss_workspace(self.N, rec['G'], rec['A'], rec['C'], nT)
        if self.loss_ws[h] is None or self.loss_ws[h].numel() < need:
            if self.loss_ws[h] is not None:
                self._retire(self.loss_ws[h])
            self.loss_ws[h] = self.arena.new('loss_ws[%d]' % h, need, torch.uint8)
        dl = self.dlogits[h]
        if dl is None:
            dl = self.dlogits[h] = self.arena.new('dlogits[%d]' % h, logits.numel(), torch.float32)
        ops.yolo_loss(logits, self.N, rec['G'], rec['A'], rec['C'], targets, rec['anchors'], img_size,
                      rec['ignore_thresh'], use_giou, self.loss_ws[h], self.metrics[h], dl)

    def _run_heads(self, targets, use_giou, img_size):
        recs, self._pending_heads = self._pending_heads, []
        r0 = recs[0]
        cap = self._rows_bucket(targets.shape[0])
        if self._head_table is None or self._head_table[2] < cap:
            wcap = cap if self._head_table is None else max(cap, 2 * self._head_table[2])     # room to grow: KITTI batches vary
            need = ops.yolo_loss_multi_workspace([r['G'] for r in recs], self.N, r0['A'], r0['C'], wcap)
            if self._head_table is not None:
                self._retire(self._head_table)      # (table and workspace: see __init__)
            ws = self.arena.new('heads workspace', need, torch.uint8)
            heads = []
            for r in recs:
                h = r['head']
                if self.dlogits[h] is None:
                    self.dlogits[h] = self.arena.new('dlogits[%d]' % h, self.act[r['logits'].st.sid].numel(), torch.float32)
                heads.append((self.act[r['logits'].st.sid], self.dlogits[h], self.metrics[h], r['anchors'], r['G'], r['row_offset']))
            self._head_table = (ops.make_head_table(heads), ws, wcap)
        table, ws, _ = self._head_table
        # sized for the bucket, live count on the device (the workspace layout follows the capacity PASSED, not the allocated one)
        ops.yolo_loss_multi(table, len(recs), self.N, r0['A'], r0['C'], self._tg_buf, img_size, r0['ignore_thresh'], use_giou, ws,
                            self.outputs, self.plan.rows_total, cap=cap, nt_dev=self._nt_dev)

    def check_grid_waits(self):
        """Raise if a two-phase launch (ops.conv_bn_act_train) ever gave up waiting for its grid (ticket[2]): its outputs were
        invalid.  Costs a device synchronisation: for tests, bench.py's end and whoever wants the assurance, not the step path."""
        if self._ticket is not None and int(self._ticket[2]) != 0:
            raise ops.CyoloError('a two-phase conv + BN launch timed out waiting for its grid (blocks not co-resident?)')

    # ---- backward --------------------------------------------------------------------------------
    def backward(self, grads, gout_dev, loss_scale, on_module_done=None, act_scale=None):
        """grads: {param name: fp32 gradient tensor (accumulated into)}; gout_dev: device scalar d(loss).
        ``act_scale`` multiplies d(logits) before the half-precision backward; parameter gradients are divided by
        ``loss_scale`` (= act_scale, or act_scale * world so that a plain SUM all-reduce yields the mean).
        on_module_done(idx) is called after the kernels that finish module idx's parameter gradients are queued."""
        assert self.training
        self.passes += 1
        self.grads, self.ls = grads, float(loss_scale)
        self._gout_buf.copy_(gout_dev.reshape(-1)[:1], non_blocking=True)      # fixed address for the recorded list
        self.gout = self._gout_buf
        self.act_scale = float(loss_scale if act_scale is None else act_scale)
        self._on_module_done = on_module_done
        tuned = self._wgrad_tuned and self._dgrad_tuned
        if not self._wgrad_tuned:
            self._autotune_wgrad()
        if not self._dgrad_tuned:
            self._autotune_dgrad()
        if self._reduce_groups is None or self._reduce_key != self._grads_key(grads):
            self._build_reduce_groups()
        flush_at = {g['last']: g for g in self._reduce_groups}
        on_dev = getattr(self.device, 'type', str(self.device)) == 'cuda'
        self._main_stream = torch.cuda.current_stream(self.device) if on_dev else None
        if self.side is not None:
            self._side_scope = ops.stream_scope(self.side)
            self._main_h, self._side_h = ops.stream_handle(self._main_stream), ops.stream_handle(self.side)
            if self._fork_ev is None:
                self._fork_ev, self._join_ev = ops.Event(), ops.Event()
        if self.bnpart_pair is not None:
            self._bp = 0
            if self._bn_tables_fwd != self.fwd_serial:     # (zeroed by this step's forward pass otherwise)
                self.bnpart_pair[0].zero_()
                self.bnpart_pair[1].zero_()
                if self._cat:
                    self._cat_tables.zero_()
            self._bn_tables_fwd = -1
        key = prog = None
        can = (self.replay and tuned and on_dev and self._fwd_key is not None and ops.PROFILER is None and ops.recording() is None)
        if can:
            key = (self._fwd_key, self._reduce_key, self.ls, self.act_scale, on_module_done is not None, self._main_stream.cuda_stream,
                   self.side is not None)
            prog = self._bwd_progs.get(key)
        if prog is not None:
            self._bp = prog.bp_after
            prog.run()
            self.replayed += 1
            return
        if can:
            ops.start_recording()
        try:
            with self._scope():
                for rec in self.plan.bwd:
                    getattr(self, '_b_' + rec['op'])(rec)
                    if rec['op'] in ('conv_bwd', 'head_conv_bwd'):
                        g = flush_at.get(rec['fwd']['idx'])
                        if g is not None:
                            self._flush_group(g)
                if self.side is not None:      # join: the pass's stream waits for the weight-gradient stream
                    ops.event_record(self._join_ev, self._side_h)
                    ops.stream_wait_event(self._main_h, self._join_ev)
        except BaseException:
            if can:
                ops.stop_recording(keep=False)
            raise
        if can:
            prog = ops.stop_recording()
            prog.bp_after = self._bp
            self._remember(self._bwd_progs, key, prog)

    def _flush_group(self, g):
        """Fold the split-K slabs of one group of convs into the flat gradient and announce its modules as final.  With
        the side stream this also runs there (after the main stream's BN-parameter gradients of the group, which the
        side stream waits for), so the main stream never stalls on weight gradients before the end of backward.  The
        announcement (the data-parallel wrapper's bucketed all-reduce) is host code: in a recorded pass it sits between two
        C segments and reads the CURRENT backward's hook."""
        mods = g['mods']
        if self.side is None:
            ops.wgrad_reduce_multi(g['desc'], g['blocks'], 1.0 / self.ls, True)

            def announce():
                for idx in mods:
                    self._on_module_done(idx)
        else:
            ops.event_record(self._fork_ev, self._main_h)
            ops.stream_wait_event(self._side_h, self._fork_ev)
            with self._side_scope:
                ops.wgrad_reduce_multi(g['desc'], g['blocks'], 1.0 / self.ls, True)

            def announce():
                with torch.cuda.stream(self.side):
                    for idx in mods:
                        self._on_module_done(idx)
        if self._on_module_done is not None:
            announce()
            rec = ops.recording() if hasattr(ops, 'recording') else None
            if rec is not None:
                rec.py(announce)

    def _autotune_wgrad(self):
        """Pick the split-K factor of every weight-gradient launch (once, at the first backward; shapes are static): a hit in
        the persisted table (tune.py) is used as it is; otherwise the default mode times the candidates.  The heuristic of
        cy_conv_wgrad_split is within ~10-25 % of the best split for most layers but the optimum depends on how
        tiles x split quantises over the CUs; cost = kernel time + the fold's share for the slabs.  deterministic=True
        never times (the split decides how the pixel sum is partitioned, i.e. the fp32 rounding): table or heuristic, both
        functions of the shape alone.  CY_WGRAD_AUTOTUNE=0 keeps the heuristic."""
        self._wgrad_tuned = True
        if getattr(self.device, 'type', str(self.device)) != 'cuda' or os.environ.get('CY_WGRAD_AUTOTUNE', '1') == '0':
            return
        memo = {}
        reps = int(os.environ.get('CY_TUNE_REPS', '3'))
        heads = {id(h['conv']): i for i, h in enumerate(self.plan.heads)}
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rec in self.plan.convs:
            idx = rec['idx']
            cop, cip, kk = _wgrad_rows(rec), rec['cin_pad'], rec['ks'] * rec['ks']
            dy = self.head_tmp[heads[id(rec)]] if id(rec) in heads else self.view(rec['out'], grad=True)
            pr, role = self._pair(idx)
            if role == 1:
                continue                  # (its weight gradient = rows of its sibling's joint launch)
            if pr is not None:
                dy, cop = pr['draw'][2], pr['Ca'] + pr['Cb']
            xv = self.view(rec['x'])
            key = ('wgrad', self.dt, dy.N, dy.H, dy.W, dy.C, dy.ld, xv.H, xv.W, xv.C, xv.ld, rec['ks'], rec['stride'], rec['pad'])
            if key not in memo:
                s0, cap = self.wsplit[idx], self.wsplit_cap[idx]
                hit = tune.get(key)
                if hit is not None and self._wgrad_choice_ok(int(hit[0]), cap, dy.M):
                    memo[key] = int(hit[0])
                    tune.put(key, *hit)
                elif self.det and not _DET_TIMING:
                    memo[key] = s0
                else:
                    cands = sorted({c for c in list(range(max(1, s0 // 3), min(cap, s0 + 8) + 1)) + [cap, (s0 + cap) // 2] if 1 <= c <= cap})
                    if len(cands) > 24:
                        cands = sorted(set(cands[::max(1, len(cands) // 24)] + [s0]))
                    slab_us = cop * kk * cip * 4 / 4.2e6   # fold: 4.2 TB/s over the slabs (2.49 GB in 0.58 ms, profiles/r03_*)
                    best, best_cost = s0, None
                    off = self.wslab_off[idx]
                    # atomic mode (default mode only): every split adds into ONE resident slab -- the fold then reads a single
                    # slab and writes zeros back (2 slab passes whatever the split), and more splits cost no memory
                    am = os.environ.get('CY_WGRAD_ATOMIC', '0')      # opt-in: measured no better than slabs
                    modes = [(False, False)] if (self.det or am == '0') else ([(True, False)] if am == '2' else [(False, False), (True, False)])
                    # 64 x 64 tiles for the 16-bit layers whose 128 x 128 tiling has few tiles and therefore a deep split: a
                    # quarter of the slabs for the same number of blocks (choice encoded as split + 1000)
                    if self.dt != CY_F32 and not self.det and s0 >= 8 and cop >= 128 and kk * cip >= 128 and os.environ.get('CY_WGRAD_TILE64', '1') != '0':
                        modes.append((False, True))
                    for atomic, t64 in modes:
                        cl = cands if not atomic else sorted(set(cands + [2 * cands[-1], 4 * cands[-1]]))
                        if t64:
                            cl = sorted({max(1, c // 4) for c in cands} | {max(1, c // 3) for c in cands[:4]})
                        for c in cl:
                            if (c - 1) * 512 >= dy.M and c > 1:
                                continue
                            part = self.wpart[off:off + (1 if atomic else c) * cop * kk * cip]
                            ops.conv_wgrad(dy, xv, rec['ks'], rec['stride'], rec['pad'], part, c, atomic=atomic, tile64=t64)
                            ev0.record()
                            for _ in range(reps):
                                ops.conv_wgrad(dy, xv, rec['ks'], rec['stride'], rec['pad'], part, c, atomic=atomic, tile64=t64)
                            ev1.record()
                            ev1.synchronize()
                            cost = ev0.elapsed_time(ev1) * 1e3 / reps + (2 if atomic else c) * slab_us
                            if best_cost is None or cost < best_cost:
                                best, best_cost = (-c if atomic else (c + 1000 if t64 else c)), cost
                    memo[key] = best
                    tune.put(key, best, best_cost * 1e-3)
            if memo[key] < 0:
                self.watomic[idx], self.wsplit[idx] = -memo[key], 1
            elif memo[key] >= 1000:
                self.wtile64.add(idx)
                self.wsplit[idx] = memo[key] - 1000
            else:
                self.wsplit[idx] = memo[key]
        for rec in self.plan.convs:
            if self._first_fused_rec(rec):
                # its own kernel (register-staged, 256-thread blocks over 64-pixel steps, HBM-bound on g + raw): as many pixel
                # splits as the slab region holds, up to four blocks per compute unit -- 128 blocks ran at 2.6 TB/s
                idx = rec['idx']
                self.wsplit[idx] = max(1, min(self.wsplit_cap[idx], 1024))
                self.wtile64.discard(idx)
                self.watomic.pop(idx, None)
        if self.watomic:
            self.wpart.zero_()        # atomic slabs start from zero (the timing launches added into them); the folds keep them so
        self._reduce_groups = None

    def _wgrad_choice_ok(self, v, cap, M):
        """Is a persisted weight-gradient choice usable by THIS engine?  Encoding: s = plain split into s slabs, 1000 + s = the
        same on 64 x 64 tiles, -s = atomic (s splits add into ONE slab).  The slab region of a layer holds ``cap`` slabs, so a
        slab-writing mode needs s <= cap (a table from another cap formula or a CY_TUNE_CACHE_PATH file would otherwise write
        into the next layer's slabs); an atomic split is bounded by the 512-pixel minimum per split; deterministic engines
        take plain splits only."""
        if v == 0:
            return False
        if v < 0:
            return not self.det and 1 <= -v <= max(1, (M + 511) // 512)
        if v >= 1000:
            return not self.det and 1 <= v - 1000 <= cap
        return 1 <= v <= cap

    # ---- conv kernel / tile choice ---------------------------------------------------------------------
    @property
    def _stat_flags(self):
        return CONV_STATS | (ops.CONV_STATS_DET if self.det else 0)

    def _tunable(self):
        return (getattr(self.device, 'type', str(self.device)) == 'cuda' and self.dt != CY_F32 and hasattr(ops, 'CONV_TILE_HINTS')
                and os.environ.get('CY_CONV_AUTOTUNE', '1') != '0')

    def _time_hints(self, key, launch, cin, cout, ks=0, slab=False):
        """Best kernel / tile hint for one conv launch shape: looked up (tune.py) or timed (1 warm-up + CY_TUNE_REPS
        launches per candidate between HIP events, the fastest kept).  The 4-wave and the 8-wave kernels are within +-10 % of each other on v4's layers and
        the winner depends on how tiles quantise over the 256 CUs, so it is measured, once per shape and process."""
        return self._time_hints_t(key, launch, cin, cout, ks=ks, slab=slab)[0]

    def _time_hints_t(self, key, launch, cin, cout, pipe_only=False, ks=0, hints=None, extra=(), slab=False):
        """-> (best hint, its time in ms per launch); pipe_only leaves the 4-wave kernels out and returns (None, None)
        when the pipelined kernel does not take the shape.  Order of authority: this process's memo, the persisted table
        (tune.py), then -- default mode only -- a timing run over the candidates (``hints`` overrides the candidate list).
        deterministic=True never times: a table miss takes hint 0, the library's shape-only default."""
        memo = _CONV_TUNE_MEMO.get(key)
        if memo is not None:
            return memo
        hit = tune.get(key)
        if hit is not None:
            hit = (hit[0], hit[1])
            _CONV_TUNE_MEMO[key] = hit
            tune.put(key, *hit)
            return hit
        if self.det and not _DET_TIMING:
            return (None if pipe_only else 0, None)      # (not memoised: the key may be shared with default-mode engines)
        if hints is None:
            hints = list(extra) + ([] if pipe_only else [1])
            if ks == 3 and cin in (8, 32) and not pipe_only:
                hints.append(0)       # forward 3 -> 32 / 32 -> 64: the library default is the direct small-Cin kernel (conv_direct.hip)
            if ks == 1 and (cin, cout) in _PW_DIRECT and not pipe_only:
                hints.append(10)      # 1x1 streams: the direct kernel, also below the library's own size threshold
            if cin % 64 == 0 and cout % 8 == 0:
                hints += [h for h in ops.CONV_TILE_HINTS if h != 1 and not (h in (3, 8) and cout <= 64)]
                if slab and cout > 64:
                    hints += list(ops.CONV_SLAB_HINTS)      # 3x3 / stride 1: the slab kernel (conv_pipe.hip)
        reps = int(os.environ.get('CY_TUNE_REPS', '3'))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best, best_t = (None if pipe_only else 1), None
        for h in hints:
            try:
                launch(h)
            except ops.CyoloError:
                if h in extra:
                    continue          # an optional kernel that does not take this shape
                if pipe_only:
                    break
                raise
            ev0.record()
            for _ in range(reps):
                launch(h)
            ev1.record()
            ev1.synchronize()
            t = ev0.elapsed_time(ev1) / reps
            if best_t is None or t < best_t * 0.98:      # a challenger must win by 2 %: ties keep the earlier candidate
                best, best_t = h, t
        _CONV_TUNE_MEMO[key] = (best, best_t)
        tune.put(key, best, best_t)
        return best, best_t

    def _autotune_fwd(self):
        self._fwd_tuned = True
        if not self._tunable():
            return
        for rec in self.plan.convs:
            if not rec['bn']:
                continue
            idx, cop = rec['idx'], _pad32(rec['cout'])
            xv = self.view(rec['x'])
            pr, role = self._pair(idx)
            if role == 1:
                continue                  # (convolved by its sibling's launch)
            if self.training:
                raw, wf = self.view(rec['raw']), self.wf[idx]
                if pr is not None:
                    raw, wf, cop = pr['raw'][2], pr['wf'], pr['Ca'] + pr['Cb']
                key = ('fwd', self.dt, self.det, xv.N, xv.H, xv.W, xv.C, xv.ld, raw.C, raw.ld, rec['ks'], rec['stride'])
                self._fwd_tile[idx] = self._time_hints(key, lambda h: ops.conv_igemm(
                    xv, wf, cop, raw, rec['ks'], rec['stride'], rec['pad'], flags=self._stat_flags, stats=self.stats,
                    tile=h), xv.C, raw.C, ks=rec['ks'], slab=_slab_shape(rec))
            else:
                out = self.view(rec['out'])
                res = self.view(rec['res']) if rec['res'] is not None else None
                vec = self.bnvec[idx]
                key = ('eval', self.dt, xv.N, xv.H, xv.W, xv.C, xv.ld, out.C, out.ld, rec['ks'], rec['stride'], res is not None)
                self._fwd_tile[idx] = self._time_hints(key, lambda h: ops.conv_bn_act_eval(
                    xv, self.wf[idx], cop, out, rec['ks'], rec['stride'], rec['pad'], vec[2], vec[3], ops.ACT[rec['act']], res,
                    tile=h), xv.C, out.C, ks=rec['ks'], slab=_slab_shape(rec))
        if self.training and self.fused_bn and self.conv_bn_fused and not self.det and hasattr(ops, 'conv_bn_act_train'):
            self._autotune_fwd_fused()      # (atomic statistics bins: never in the bit-reproducible mode)
        self.stats.zero_()        # the timed launches added into the statistics table
        if self.stats_pair is not None:
            self.stats_pair[1].zero_()

    def _autotune_fwd_fused(self):
        """Which BN convs run as ONE two-phase launch (conv -> grid ticket -> BN + activation from the accumulators).  Per layer:
        the best separate conv (already chosen) + the BN / activation pass against the fused launch over the pipelined kernel's
        tiles that keep the grid co-resident; persisted like every other choice."""
        self._ticket = self.arena.new('ticket', 4, torch.int32, zero=True)
        P = self.params
        for rec in self.plan.convs:
            if not rec['bn']:
                continue
            idx, cop = rec['idx'], _pad32(rec['cout'])
            xv, raw, out = self.view(rec['x']), self.view(rec['raw']), self.view(rec['out'])
            res = self.view(rec['res']) if rec['res'] is not None else None
            if xv.C % 64 or raw.C % 8 or self._pair(idx)[0] is not None or idx in self._cat_of:
                continue      # (a concatenation's producers keep their vectors in a shared block: the two-phase kernel writes [4][C])
            _, bname = self._names(rec)
            vec, act = self.bnvec[idx], ops.ACT[rec['act']]
            tbl, other = self.stats_pair
            base = (self.dt, xv.N, xv.H, xv.W, xv.C, xv.ld, raw.C, raw.ld, out.ld, rec['ks'], rec['stride'], act, res is not None)

            def fused(h):
                if not ops.conv_bn_act_train(xv, self.wf[idx], cop, raw, out, res, rec['ks'], rec['stride'], rec['pad'], tbl,
                                             P[bname + '.weight'], P[bname + '.bias'], None, None, None, BN_MOMENTUM, BN_EPS, vec,
                                             other, act, self._ticket, tile=h):
                    raise ops.CyoloError('not taken')
            cands = [h for h in (2, 3, 4, 5, 7, 8, 9) if not (h in (3, 8) and raw.C <= 64)]
            fhint, t_fused = self._time_hints_t(('fwd+bn',) + base, fused, xv.C, raw.C, pipe_only=True, extra=tuple(cands), hints=list(cands))
            if fhint is None:
                continue
            if self.conv_bn_fused < 2:
                sep_hint = self._fwd_tile.get(idx, 0)
                _, t_conv = self._time_hints_t(('fwd@',) + base + (sep_hint,), lambda h: ops.conv_igemm(
                    xv, self.wf[idx], cop, raw, rec['ks'], rec['stride'], rec['pad'], flags=self._stat_flags, stats=tbl, tile=sep_hint),
                    0, 0, hints=[sep_hint])
                _, t_bn = self._time_hints_t(('bn_act_fwd',) + base, lambda h: ops.bn_act_fwd_fused(
                    raw, out, res, tbl, ops.conv_stats_rows(raw.M, raw.C), P[bname + '.weight'], P[bname + '.bias'], None, None, None,
                    BN_MOMENTUM, BN_EPS, vec, other, act), 0, 0, hints=[1])
                if os.environ.get('CY_TUNE_VERBOSE'):
                    print('conv+bn L%d (k%d s%d %d->%d @%d): conv hint %s %.1f us + bn/act %.1f us vs fused hint %s %.1f us'
                          % (idx, rec['ks'], rec['stride'], xv.C, raw.C, raw.H, sep_hint, 1e3 * t_conv, 1e3 * t_bn, fhint, 1e3 * t_fused), flush=True)
                if t_fused is None or t_conv is None or t_bn is None or t_fused >= t_conv + t_bn:
                    continue
            self._fwd_fused[idx] = fhint
        self._ticket.zero_()

    def _autotune_dgrad(self):
        self._dgrad_tuned = True
        can_fuse = self.fused_bn and self.training and self.dgrad_bn_sums and hasattr(ops, 'conv_dgrad_bn_sums')
        if can_fuse and self.dgrad_bn_sums == 2:
            sim = getattr(self.device, 'type', str(self.device)) != 'cuda'
            for b in self.plan.bwd:
                for ri, L in b.get('dx_sums', {}).items():
                    if sim or (self.dt != CY_F32 and b['fwd']['cout'] % 64 == 0 and L['cout'] % 8 == 0):   # what the kernel takes
                        self._dgrad_sums[(b['fwd']['idx'], b['dx'][ri][0].c0)] = (L, 6)
                        self._sums_fused.add(L['idx'])
                ct = self._cat.get(b['fwd']['idx']) if b['op'] == 'conv_bwd' else None
                if ct is not None and (sim or b['fwd']['cout'] % 64 == 0):
                    self._take_cat(b, ct, 6)
            return
        if not self._tunable():
            return
        heads = {id(h['conv']): i for i, h in enumerate(self.plan.heads)}
        for b in self.plan.bwd:
            if b['op'] not in ('conv_bwd', 'head_conv_bwd'):
                continue
            rec = b['fwd']
            dy = self.head_tmp[heads[id(rec)]] if id(rec) in heads else self.view(rec['out'], grad=True)
            wd, x = self.wd[rec['idx']], rec['x']
            pr, role = self._pair(rec['idx'])
            if role == 1:
                continue                  # (its input gradient is part of its sibling's joint launch)
            runs = b['dx']
            if pr is not None:
                dy, runs = pr['draw'][2], [(b['dx'][0][0], False)]      # K = Ca + Cb, the ONE writer of the input gradient: a store
            for ri, (ref, acc) in enumerate(runs):
                r0 = ref.c0 - x.c0
                gv = self.view(ref, grad=True)
                flags = CONV_TRANSPOSED | (CONV_ACCUM if acc else 0)
                key = ('dgrad', self.dt, dy.N, dy.H, dy.W, dy.C, dy.ld, gv.H, gv.W, gv.C, gv.ld, rec['ks'], rec['stride'], acc)
                # (timing an accumulating launch adds garbage into a gradient buffer that the real backward has not written
                # yet at this point: every first writer of the step stores)
                # stride-2 3x3: hint 10 = the direct small-channel kernel (conv_direct.hip) where it takes the shape; every other
                # hint runs the four parity classes as one launch of the implicit-GEMM kernels
                s2 = (10,) if (rec['ks'] == 3 and rec['stride'] == 2 and dy.C == 64 and ref.C == 32) else ()
                hint, t_plain = self._time_hints_t(key, lambda h: ops.conv_igemm(
                    dy, wd[r0:r0 + ref.C], ref.C, gv, rec['ks'], rec['stride'], rec['pad'], flags=flags, tile=h), dy.C, ref.C,
                    ks=(1 if rec['ks'] == 1 and rec['stride'] == 1 else 0), extra=s2, slab=_slab_shape(rec))
                self._dgrad_tile[(rec['idx'], ref.c0)] = hint
                L = b.get('dx_sums', {}).get(ri) if can_fuse else None
                ct = self._cat.get(rec['idx']) if (can_fuse and L is None and pr is None and ri == 0 and id(rec) not in heads) else None
                if L is None and ct is None:
                    continue
                # fused (dgrad + sums in its epilogue) against separate (best dgrad, then the reduce pass over raw and gradient)
                if ct is not None:      # the two producers of a concatenation as ONE 'layer' of C1 + C2 channels
                    vec, raw, act, tbl = ct['vec'], ct['raw'], ops.ACT[ct['act']], ct['tbl']
                else:
                    vec, raw, act = self.bnvec[L['idx']], self.view(L['raw']), ops.ACT[L['act']]
                    tbl = self.bnpart_pair[0]
                fhint, t_fused = self._time_hints_t(('dgrad+sums', act, raw.ld) + key[1:], lambda h: ops.conv_dgrad_bn_sums(
                    dy, wd[r0:r0 + ref.C], ref.C, gv, rec['ks'], rec['stride'], rec['pad'], raw, vec[0], vec[1], vec[2], vec[3],
                    act, tbl, flags=flags, tile=h), dy.C, ref.C, pipe_only=True,
                    extra=s2 or ((10,) if (rec['ks'] == 1 and rec['stride'] == 1 and (dy.C, ref.C) in _PW_DIRECT) else ()),
                    slab=_slab_shape(rec))
                if fhint is None:
                    continue
                if ct is not None:
                    t_reduce = 0.0
                    for Lp in (ct['L1'], ct['L2']):      # what the stage pays now: one reduce pass per producer
                        rv, gp, vp = self.view(Lp['raw']), self.view(Lp['out'], grad=True), self.bnvec[Lp['idx']]
                        rows = ops.bn_bwd_rows(rv.M, rv.C, self.dt, False)
                        _, t = self._time_hints_t(('bn_bwd_reduce', act, self.dt, rv.M, rv.C, rv.ld, gp.ld), lambda h: ops.bn_act_bwd_reduce(
                            rv, gp, vp[0], vp[1], vp[2], vp[3], act, self.bnpart_pair[0], rows), 0, 0, hints=[1])
                        t_reduce += t
                    if os.environ.get('CY_TUNE_VERBOSE'):
                        print('dgrad+sums L%d|L%d <- conv %d (k%d %d->%d @%d): plain hint %s %.1f us + 2 reduces %.1f us vs fused hint %s %.1f us'
                              % (ct['L1']['idx'], ct['L2']['idx'], rec['idx'], rec['ks'], dy.C, ref.C, gv.H, hint, 1e3 * t_plain,
                                 1e3 * t_reduce, fhint, 1e3 * t_fused), flush=True)
                    if self.cat_sums == 2 or t_fused < t_plain + t_reduce:
                        self._take_cat(b, ct, fhint)
                    continue
                rows = ops.bn_bwd_rows(raw.M, raw.C, self.dt, False)
                _, t_reduce = self._time_hints_t(('bn_bwd_reduce', act, self.dt, raw.M, raw.C, raw.ld, gv.ld), lambda h: ops.bn_act_bwd_reduce(
                    raw, gv, vec[0], vec[1], vec[2], vec[3], act, tbl, rows), 0, 0, hints=[1])
                if os.environ.get('CY_TUNE_VERBOSE'):
                    print('dgrad+sums L%d <- conv %d (k%d s%d %d->%d @%d): plain hint %s %.1f us + reduce %.1f us vs fused hint %s %.1f us'
                          % (L['idx'], rec['idx'], rec['ks'], rec['stride'], dy.C, ref.C, gv.H, hint, 1e3 * t_plain, 1e3 * t_reduce,
                             fhint, 1e3 * t_fused), flush=True)
                if t_fused < t_plain + t_reduce:
                    self._dgrad_sums[(rec['idx'], ref.c0)] = (L, fhint)
                    self._sums_fused.add(L['idx'])
        if self.bnpart_pair is not None:     # the timed launches added into the sum tables
            self.bnpart_pair[0].zero_()
            self.bnpart_pair[1].zero_()
            if self._cat:
                self._cat_tables.zero_()

    def _take_cat(self, b, ct, hint):
        """The closing conv's dgrad takes the BatchNorm-backward sums of both producers of its concatenation (hint = its tile)."""
        rec = b['fwd']
        self._dgrad_sums[(rec['idx'], b['dx'][0][0].c0)] = (dict(idx=None, act=ct['act'], cat=ct), hint)
        for L, c0 in ((ct['L1'], 0), (ct['L2'], ct['C1'])):
            self._sums_fused.add(L['idx'])
            self._cat_on[L['idx']] = (ct, c0)

    def _wgrad(self, rec, dy, xv):
        """Weight gradient of one conv.  It is off the critical path of backward (only the optimizer needs it), so it
        is issued on a side HIP stream: its MFMA blocks fill the tails of, and run beside, the HBM-bound BN passes and
        the dgrad of the following layers on the main stream.  The fold of each group waits for the side stream."""
        if self.side is not None:
            # (host cost matters here: 110 of these per step.  One reusable event per conv; the launch goes to the side stream
            # through the operator layer's stream scope alone -- no torch op runs inside, so torch's notion of the current
            # stream does not have to follow)
            ev = self._wgrad_ev.get(rec['idx'])
            if ev is None:
                ev = self._wgrad_ev[rec['idx']] = ops.Event()
            ops.event_record(ev, self._main_h)
            ops.stream_wait_event(self._side_h, ev)
            with self._side_scope:
                self._wgrad_launch(rec, dy, xv)
            return
        self._wgrad_launch(rec, dy, xv)

    def _first_fused_rec(self, rec):
        """Does this conv's backward run as ONE kernel (BatchNorm backward inside the weight gradient, ops.conv_wgrad_bn)?  A
        BatchNorm conv without an input gradient and without a folded shortcut, at most 32 output channels: the first layer."""
        return (self._first_fused and self.fused_bn and rec['first'] and rec['bn'] and rec.get('res') is None
                and rec['cout'] <= 32 and rec['cout'] % 8 == 0)

    def _wgrad_bn(self, rec, raw, g, tbl, rows, other, act):
        idx = rec['idx']
        _, bname = self._names(rec)
        vec = self.bnvec[idx]
        cop, cip = _wgrad_rows(rec), rec['cin_pad']
        sp = self.wsplit[idx]
        off = self.wslab_off[idx]
        part = self.wpart[off:off + sp * cop * rec['ks'] * rec['ks'] * cip]

        def launch():
            with ops.prof('wgrad', *self._conv_work(rec)):
                ops.conv_wgrad_bn(g, raw, self.view(rec['x']), rec['ks'], rec['stride'], rec['pad'], vec[0], vec[1], vec[2], vec[3],
                                  tbl, rows, self.grads[bname + '.weight'], self.grads[bname + '.bias'], 1.0 / self.ls, other, act,
                                  part, sp)
        if self.side is not None:
            ev = self._wgrad_ev.get(idx)
            if ev is None:
                ev = self._wgrad_ev[idx] = ops.Event()
            ops.event_record(ev, self._main_h)
            ops.stream_wait_event(self._side_h, ev)
            with self._side_scope:
                launch()
            return
        launch()

    def _wgrad_launch(self, rec, dy, xv):
        idx = rec['idx']
        cname, _ = self._names(rec)
        cop, cip = _wgrad_rows(rec), rec['cin_pad']
        sp = self.wsplit[idx]
        off = self.wslab_off[idx]
        part = self.wpart[off:off + sp * cop * rec['ks'] * rec['ks'] * cip]
        with ops.prof('wgrad', *self._conv_work(rec)):
            if idx in self.watomic:
                ops.conv_wgrad(dy, xv, rec['ks'], rec['stride'], rec['pad'], part, self.watomic[idx], atomic=True)
            elif idx in self.wtile64:
                ops.conv_wgrad(dy, xv, rec['ks'], rec['stride'], rec['pad'], part, sp, tile64=True)
            else:
                ops.conv_wgrad(dy, xv, rec['ks'], rec['stride'], rec['pad'], part, sp)

    def _dgrad(self, rec, dy, runs):
        wd = self.wd[rec['idx']]
        x = rec['x']
        for ref, acc in runs:
            r0 = ref.c0 - x.c0
            fl, by = self._conv_work(rec)
            fused = self._dgrad_sums.get((rec['idx'], ref.c0))
            # (bench.py's launch brackets: a fused launch also reads the producer layer's pre-BN tensor)
            with ops.prof('igemm_sums' if fused is not None else 'igemm', fl * ref.C / x.C,
                          by + (self.view(ref, grad=True).M * ref.C * 2 if fused is not None else 0)):
                if fused is not None:
                    # the sums of layer L go into the table L's backward will pick next (zeroed by the apply pass that just ran);
                    # those of a concatenation's two producers into that concatenation's own table
                    L, hint = fused
                    ct = L.get('cat')
                    vec = self.bnvec[L['idx']] if ct is None else ct['vec']
                    ops.conv_dgrad_bn_sums(dy, wd[r0:r0 + ref.C], ref.C, self.view(ref, grad=True), rec['ks'], rec['stride'],
                                           rec['pad'], self.view(L['raw']) if ct is None else ct['raw'], vec[0], vec[1], vec[2], vec[3],
                                           ops.ACT[L['act']], self.bnpart_pair[self._bp] if ct is None else ct['tbl'],
                                           flags=CONV_TRANSPOSED | (CONV_ACCUM if acc else 0), tile=hint)
                    continue
                ops.conv_igemm(dy, wd[r0:r0 + ref.C], ref.C, self.view(ref, grad=True), rec['ks'], rec['stride'],
                               rec['pad'], flags=CONV_TRANSPOSED | (CONV_ACCUM if acc else 0),
                               tile=self._dgrad_tile.get((rec['idx'], ref.c0), 0))

    def _b_conv_bwd(self, b):
        rec = b['fwd']
        idx = rec['idx']
        _, bname = self._names(rec)
        vec = self.bnvec[idx]
        mean, invstd, scale, shift = vec[0], vec[1], vec[2], vec[3]
        raw, g = self.view(rec['raw']), self.view(rec['out'], grad=True)
        C, M = rec['cout'], raw.M
        act = ops.ACT[rec['act']]
        rows = ops.bn_bwd_rows(M, C, self.dt, self.det)
        bld = bc0 = 0
        if self.fused_bn:
            tbl, other = self.bnpart_pair[self._bp], self.bnpart_pair[self._bp ^ 1]
            self._bp ^= 1
            if idx in self._sums_fused:      # the dgrad that last wrote g already left the sums in tbl
                rows = ops.conv_stats_rows(M, C, False)
                if idx in self._cat_on:      # ... in its concatenation's table (this layer's columns); the pair alternates as ever
                    ct, bc0 = self._cat_on[idx]
                    tbl, bld = ct['tbl'], ct['C1'] + ct['C2']
            else:
                ops.bn_act_bwd_reduce(raw, g, mean, invstd, scale, shift, act, tbl, rows)
        else:
            ops.bn_act_bwd_reduce(raw, g, mean, invstd, scale, shift, act, self.bnpart, rows)
            ftbl, frows = self.bnpart, rows
            if rows > 256:
                ftbl, frows = self.fold_tmp, ops.fold_rows(self.bnpart, rows, 2 * C, self.fold_tmp)
            ops.bn_bwd_finalize(ftbl, frows, C, self.dgs, self.dbs,
                                self.grads[bname + '.weight'], self.grads[bname + '.bias'], 1.0 / self.ls)
        res_view, res_acc = None, False
        runs = b['res_runs']
        if len(runs) == 1:
            res_view, res_acc = self.view(runs[0][0], grad=True), runs[0][1]
        elif len(runs) > 1:
            res = rec['res']
            for ref, acc in runs:
                ops.slice_copy(g.channels(ref.c0 - res.c0, ref.C), self.view(ref, grad=True), accumulate=acc)
        pr, role = self._pair(idx)
        if pr is not None:
            # sibling pair (backward order: the follower B first, then the lead A): each module's BatchNorm backward writes its
            # dRaw into its slice of the joint buffer instead of in place; after A's, ONE weight gradient (dY = [A | B]) and ONE
            # input gradient (K = Ca + Cb, a store: no fan-in read-add-store) run on the joint operands
            ops.bn_act_bwd_apply_fused(raw, g, pr['draw'][role], res_view, res_acc, mean, invstd, scale, shift, tbl, rows,
                                       self.grads[bname + '.weight'], self.grads[bname + '.bias'], 1.0 / self.ls, other, act,
                                       bins_ld=bld, bins_c0=bc0)
            if role == 0:
                self._wgrad(pr['rec'], pr['draw'][2], self.view(rec['x']))
                self._dgrad(pr['rec'], pr['draw'][2], [(b['dx'][0][0], False)])
            return
        if self._first_fused_rec(rec) and not b['dx'] and not runs:
            # a conv block without an input gradient (the first layer): nothing but its own weight gradient reads dRaw -- the
            # BatchNorm backward is applied inside the weight-gradient kernel and the 378 MB tensor is neither written nor read
            # back (cy_conv_wgrad_bn); on the weight-gradient stream, like every weight gradient
            self._wgrad_bn(rec, raw, g, tbl, rows, other, act)
            return
        if self.fused_bn:
            ops.bn_act_bwd_apply_fused(raw, g, g, res_view, res_acc, mean, invstd, scale, shift, tbl, rows,
                                       self.grads[bname + '.weight'], self.grads[bname + '.bias'], 1.0 / self.ls, other, act,
                                       bins_ld=bld, bins_c0=bc0)
        else:
            ops.bn_act_bwd_apply(raw, g, g, res_view, res_acc, mean, invstd, scale, shift, self.dgs, self.dbs, act)
        self._wgrad(rec, g, self.view(rec['x']))
        self._dgrad(rec, g, b['dx'])

    def _b_head_conv_bwd(self, b):
        rec = b['fwd']
        cname, _ = self._names(rec)
        h = next(i for i, hd in enumerate(self.plan.heads) if hd['conv'] is rec)
        hd = self.plan.heads[h]
        M, nch = self.N * hd['G'] * hd['G'], hd['A'] * (7 + hd['C'])
        tmp = self.head_tmp[h]
        ops.f32_to_view(self.dlogits[h], M, nch, self.act_scale, tmp, 32, scale_dev=self.gout)
        if self.det:
            ops.bias_grad_det(self.dlogits[h], M, nch, self.act_scale / self.ls, self.grads[cname + '.bias'], self.fold_tmp,
                              scale_dev=self.gout)
        else:
            ops.bias_grad(self.dlogits[h], M, nch, self.act_scale / self.ls, self.grads[cname + '.bias'], scale_dev=self.gout)
        self._wgrad(rec, tmp, self.view(rec['x']))
        self._dgrad(rec, tmp, b['dx'])

    def _b_pool_bwd(self, b):
        rec = b['fwd']
        if len(b['dx']) == 1:
            ref, acc = b['dx'][0]
        else:
            # mixed fan-in state (part of the input already holds a gradient): zero the rest, then accumulate
            for r, written in b['dx']:
                if not written:
                    ops.zero_view(self.view(r, grad=True), self.dummy)
            ref, acc = rec['x'], True
        ops.maxpool_bwd(self.view(rec['out'], grad=True), self.argmax[rec['idx']], self.view(ref, grad=True), rec['k'],
                        rec['stride'], rec['pad'], acc, self.pool_scratch)

    def _b_upsample_bwd(self, b):
        rec = b['fwd']
        g, x = self.view(rec['out'], grad=True), rec['x']
        for ref, acc in b['dx']:
            ops.upsample_bwd(g.channels(ref.c0 - x.c0, ref.C), self.view(ref, grad=True), rec['stride'], acc)

    def _b_copy_bwd(self, b):
        rec = b['fwd']
        g, x = self.view(rec['out'], grad=True), rec['x']
        for ref, acc in b['dx']:
            ops.slice_copy(g.channels(ref.c0 - x.c0, ref.C), self.view(ref, grad=True), accumulate=acc)

    def _b_add_bwd(self, b):
        rec = b['fwd']
        g = self.view(rec['out'], grad=True)
        for key, src in (('da', rec['a']), ('db', rec['b'])):
            for ref, acc in b[key]:
                ops.slice_copy(g.channels(ref.c0 - src.c0, ref.C), self.view(ref, grad=True), accumulate=acc)

    def _b_zero_grad(self, b):
        ops.zero_view(self.view(b['ref'], grad=True), self.dummy)
