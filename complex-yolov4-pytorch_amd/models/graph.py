"""Darknet cfg -> execution plan for the HIP operator set.

The reference walks ``blocks`` in Python every forward and keeps every activation in a dict
(reference src/models/darknet2pytorch.py:162-230).  Here the walk happens once per input geometry:
the cfg is lowered to a list of operator records over pre-sized NHWC storages so that

  * [route] concatenations cost nothing: producers write straight into channel slices of the
    concatenated storage (a tensor can live in one place only; a second cat copies);
  * [route] with groups / single-layer routes are views;
  * a [shortcut] whose left operand is a conv block consumed nowhere else is folded into that block's
    BatchNorm+activation pass;
  * gradients of multi-consumer tensors are accumulated in place, the first writer storing and the
    later ones adding (tracked per storage channel at plan time).

Nothing here touches the device; tests/test_plan_sim.py runs the plan against a CPU simulator of the
operator layer.
"""
import math


def lower_blocks(blocks):
    """One dict per module (non-[net] block) with absolute source indices -- the module list layout of
    reference ``create_network`` (darknet2pytorch.py:235-401), state-dict names included."""
    mods = []
    conv_id = 0
    ch = int(blocks[0].get('channels', 3))
    out_ch = []
    for blk in blocks[1:]:
        i = len(mods)
        t = blk['type']
        m = dict(type=t, idx=i)
        if t == 'convolutional':
            conv_id += 1
            k = int(blk['size'])
            m.update(n=conv_id, cin=ch, cout=int(blk['filters']), k=k, stride=int(blk['stride']),
                     pad=(k - 1) // 2 if int(blk['pad']) else 0, bn=int(blk['batch_normalize']),
                     act=blk['activation'])
            ch = m['cout']
        elif t == 'maxpool':
            m.update(k=int(blk['size']), stride=int(blk['stride']))
        elif t == 'upsample':
            m.update(stride=int(blk['stride']))
        elif t == 'route':
            src = [int(s) for s in blk['layers'].split(',')]
            src = [s if s > 0 else s + i for s in src]
            m.update(src=src, groups=int(blk.get('groups', 1)), group_id=int(blk.get('group_id', 0)))
            ch = sum(out_ch[s] for s in src) // m['groups']
        elif t == 'shortcut':
            f = int(blk['from'])
            m.update(src=f if f > 0 else f + i, act=blk['activation'])
            ch = out_ch[i - 1]
        elif t == 'yolo':
            mask = [int(v) for v in blk['mask'].split(',')]
            a = [float(v) for v in blk['anchors'].split(',')]
            trip = [(a[j], a[j + 1], math.sin(a[j + 2]), math.cos(a[j + 2])) for j in range(0, len(a), 3)]
            m.update(anchors=[trip[j] for j in mask], classes=int(blk['classes']),
                     ignore_thresh=float(blk['ignore_thresh']), scale_x_y=float(blk.get('scale_x_y', 1.0)))
        else:
            # the reference prints and skips unknown blocks; the hot path refuses them loudly
            raise ValueError('unsupported cfg block type %r (module %d)' % (t, i))
        m['cout_total'] = ch
        out_ch.append(ch)
        mods.append(m)
    return mods


def pool_geometry(n, k, s):
    """-> (output extent, padding before) of a [maxpool] along one axis of extent n, as reference create_network picks its
    module (darknet2pytorch.py:281-292): size odd / stride 1 -> nn.MaxPool2d(k, 1, k // 2); stride == size -> nn.MaxPool2d(k, k, 0);
    anything else -> MaxPoolDark (:30-59): replicate padding of (k - 1) // 2 before and the same -- or one more, when the
    darknet and torch output sizes differ -- after, then an unpadded pool.  A replicated border element is a copy of an element
    the window already holds, so for a MAX it is the same as ignoring the taps beyond the border: value, winner and gradient
    are those of the clipped window, which is what cy_maxpool_* computes for any (out, pad)."""
    if s == 1 and k % 2:
        return n, k // 2
    if s == k:
        return n // k, 0
    p = k // 2
    p1 = (k - 1) // 2
    p2 = p1 + 1 if ((n - 1) // s) != ((n + 2 * p - k) // s) else p1
    return (n + p1 + p2 - k) // s + 1, p1


def trace_shapes(blocks, H, W):
    """[(type, (C, H, W)) per module] for an input of H x W."""
    mods = lower_blocks(blocks)
    shapes = []
    h, w = H, W
    for i, m in enumerate(mods):
        t = m['type']
        if t == 'convolutional':
            h = (h + 2 * m['pad'] - m['k']) // m['stride'] + 1
            w = (w + 2 * m['pad'] - m['k']) // m['stride'] + 1
        elif t == 'maxpool':
            h, w = pool_geometry(h, m['k'], m['stride'])[0], pool_geometry(w, m['k'], m['stride'])[0]
        elif t == 'upsample':
            h, w = h * m['stride'], w * m['stride']
        elif t == 'route':
            _, h, w = shapes[m['src'][0]][1]
        elif t == 'shortcut':
            _, h, w = shapes[i - 1][1]
        shapes.append((t, (m['cout_total'], h, w)))
    return shapes


class Storage:
    """A dense NHWC buffer [N, H, W, C]; ``kind``: 'act' (has a gradient twin in training), 'raw',
    'logits' (fp32), 'input'."""

    def __init__(self, sid, H, W, C, kind):
        self.sid, self.H, self.W, self.C, self.kind = sid, H, W, C, kind

    def __repr__(self):
        return 'S%d[%dx%dx%d %s]' % (self.sid, self.H, self.W, self.C, self.kind)


class TRef:
    """Channel slice [c0, c0+C) of a storage."""
    __slots__ = ('st', 'c0', 'C')

    def __init__(self, st, c0, C):
        self.st, self.c0, self.C = st, c0, C

    def sub(self, c0, C):
        return TRef(self.st, self.c0 + c0, C)

    def __repr__(self):
        return '%r[%d:%d]' % (self.st, self.c0, self.c0 + self.C)


class Plan:
    """Static lowering of a cfg for one (H, W).  Attributes:
    storages, fwd (list of op dicts), bwd (list of op dicts), heads, convs, rows_total."""

    def __init__(self, blocks, H, W, chunk):
        self.mods = lower_blocks(blocks)
        self.shapes = [s for _, s in trace_shapes(blocks, H, W)]
        self.H, self.W, self.chunk = H, W, chunk
        self.in_ch = int(blocks[0].get('channels', 3))
        self.storages = []
        self._build_forward()
        self._build_backward()

    # ---- helpers ---------------------------------------------------------------------------------
    def _new_storage(self, H, W, C, kind):
        s = Storage(len(self.storages), H, W, C, kind)
        self.storages.append(s)
        return s

    def _resolve(self, i):
        """Follow single-source ungrouped routes to the module that really produces the tensor."""
        m = self.mods[i]
        while m['type'] == 'route' and len(m['src']) == 1 and m['groups'] == 1:
            i = m['src'][0]
            m = self.mods[i]
        return i

    def _build_forward(self):
        mods, shapes = self.mods, self.shapes
        n = len(mods)
        # consumers of every real tensor (after alias resolution)
        consumers = {i: [] for i in range(n)}
        for i, m in enumerate(mods):
            t = m['type']
            if t in ('convolutional', 'maxpool', 'upsample', 'yolo'):
                if i > 0:
                    consumers[self._resolve(i - 1)].append(i)
            elif t == 'route':
                if len(m['src']) == 1 and m['groups'] == 1:
                    continue  # pure alias: its own consumers are attributed to the source
                for s in m['src']:
                    consumers[self._resolve(s)].append(i)
            elif t == 'shortcut':
                consumers[self._resolve(i - 1)].append(i)
                consumers[self._resolve(m['src'])].append(i)
        self.consumers = consumers

        # shortcut fusion: left operand (module i-1) is a BN conv used by this shortcut only
        fused_into = {}   # conv idx -> shortcut idx
        for i, m in enumerate(mods):
            if m['type'] == 'shortcut' and m['act'] == 'linear':
                a = self._resolve(i - 1)
                if a == i - 1 and mods[a]['type'] == 'convolutional' and mods[a]['bn'] and consumers[a] == [i] \
                        and self._resolve(m['src']) != a:
                    fused_into[a] = i
        self.fused_into = fused_into
        fused_shortcuts = set(fused_into.values())

        # cat placement: a producer writes directly into the first concatenation that lists it
        placed = {}  # producer idx -> (cat idx, channel offset)
        cat_copy = {}  # cat idx -> [(src producer idx or ('view', ...), offset)]
        producing = ('convolutional', 'maxpool', 'upsample', 'shortcut')
        for i, m in enumerate(mods):
            if m['type'] != 'route' or (len(m['src']) == 1):
                continue
            off = 0
            for s in m['src']:
                b = self._resolve(s)
                pm = mods[b]
                ok = pm['type'] in producing and b not in placed and not (pm['type'] == 'convolutional' and not pm['bn'])
                if pm['type'] == 'convolutional' and b in fused_into:
                    ok = False  # its tensor does not exist on its own
                if ok:
                    placed[b] = (i, off)
                else:
                    cat_copy.setdefault(i, []).append((s, off))
                off += shapes[s][0]
        # a fused shortcut's tensor is produced by its conv: placement of the shortcut applies to the conv's output
        self.placed = placed

        # storages for module outputs
        out = [None] * n      # TRef of each module's output
        cat_storage = {}
        for i, m in enumerate(mods):
            if m['type'] == 'route' and len(m['src']) > 1:
                C, H, W = shapes[i]
                cat_storage[i] = self._new_storage(H, W, C, 'act')

        def own_or_placed(i):
            C, H, W = shapes[i]
            if i in placed:
                ci, off = placed[i]
                return TRef(cat_storage[ci], off, C)
            return TRef(self._new_storage(H, W, C, 'act'), 0, C)

        cpad = self.chunk
        self.input = TRef(self._new_storage(self.H, self.W, cpad, 'input'), 0, cpad)
        self.fwd, self.convs, self.heads = [], [], []
        rows = 0
        for i, m in enumerate(mods):
            t = m['type']
            C, H, W = shapes[i]
            src_prev = out[i - 1] if i > 0 else self.input
            if t == 'convolutional':
                x = src_prev
                rec = dict(op='conv', idx=i, n=m['n'], x=x, cin=m['cin'], cin_pad=x.C, cout=m['cout'], ks=m['k'],
                           stride=m['stride'], pad=m['pad'], bn=bool(m['bn']), act=m['act'], H=H, W=W,
                           xH=x.st.H, xW=x.st.W, first=(i == 0))
                if m['bn']:
                    rec['raw'] = TRef(self._new_storage(H, W, C, 'raw'), 0, C)
                    if i in fused_into:
                        sc = fused_into[i]
                        out_ref = own_or_placed(sc)
                        rec['res'] = out[self._resolve(mods[sc]['src'])]
                        rec['out'] = out_ref
                        out[sc] = out_ref
                        out[i] = None
                    else:
                        rec['res'] = None
                        rec['out'] = own_or_placed(i)
                        out[i] = rec['out']
                else:
                    rec['out'] = TRef(self._new_storage(H, W, C, 'logits'), 0, C)
                    rec['res'] = None
                    out[i] = rec['out']
                self.fwd.append(rec)
                self.convs.append(rec)
            elif t == 'maxpool':
                k, s = m['k'], m['stride']
                rec = dict(op='pool', idx=i, x=src_prev, out=own_or_placed(i), k=k, stride=s,
                           pad=pool_geometry(src_prev.st.H, k, s)[1])
                out[i] = rec['out']
                self.fwd.append(rec)
            elif t == 'upsample':
                rec = dict(op='upsample', idx=i, x=src_prev, out=own_or_placed(i), stride=m['stride'])
                out[i] = rec['out']
                self.fwd.append(rec)
            elif t == 'route':
                if len(m['src']) == 1:
                    base = out[m['src'][0]]
                    if m['groups'] > 1:
                        c = base.C // m['groups']
                        base = base.sub(c * m['group_id'], c)
                    out[i] = base
                else:
                    out[i] = TRef(cat_storage[i], 0, C)
                    for s, off in cat_copy.get(i, []):
                        self.fwd.append(dict(op='copy', idx=i, x=out[s], out=TRef(cat_storage[i], off, shapes[s][0])))
            elif t == 'shortcut':
                if i in fused_shortcuts:
                    continue  # out[i] set by the conv
                rec = dict(op='add', idx=i, a=out[i - 1], b=out[m['src']], out=own_or_placed(i), act=m['act'])
                if m['act'] != 'linear':
                    raise ValueError('shortcut activation %r is not used by the hot-path cfgs' % m['act'])
                out[i] = rec['out']
                self.fwd.append(rec)
            elif t == 'yolo':
                G = H
                A = len(m['anchors'])
                rec = dict(op='yolo', idx=i, head=len(self.heads), logits=src_prev, G=G, A=A, C=m['classes'],
                           anchors=m['anchors'], ignore_thresh=m['ignore_thresh'], row_offset=rows,
                           conv=self.fwd[-1] if self.fwd and self.fwd[-1]['op'] == 'conv' else None)
                rows += A * G * G
                if src_prev.st.kind != 'logits':
                    raise ValueError('a [yolo] block must follow a linear convolution without batch_normalize')
                self.heads.append(rec)
                self.fwd.append(rec)
                out[i] = src_prev
        self.rows_total = rows
        self.out = out

    # ---- backward ---------------------------------------------------------------------------------
    def _grad_runs(self, written, ref):
        """Split ``ref`` into maximal channel runs of uniform written-state; mark them written.
        -> [(TRef, accumulate)]"""
        w = written.setdefault(ref.st.sid, [False] * ref.st.C)
        runs = []
        c = ref.c0
        end = ref.c0 + ref.C
        while c < end:
            state = w[c]
            e = c
            while e < end and w[e] == state:
                e += 1
            runs.append((TRef(ref.st, c, e - c), state))
            c = e
        for c in range(ref.c0, end):
            w[c] = True
        return runs

    def _has_grad(self, written, ref, zero_ops):
        """True when some consumer wrote a gradient for ``ref``.  Channels nobody wrote (e.g. the half of a
        tensor that only a grouped route skips) get an explicit zero-fill op first."""
        w = written.get(ref.st.sid)
        if w is None or not any(w[ref.c0:ref.c0 + ref.C]):
            return False
        c, end = ref.c0, ref.c0 + ref.C
        while c < end:
            if w[c]:
                c += 1
                continue
            e = c
            while e < end and not w[e]:
                w[e] = True
                e += 1
            zero_ops.append(dict(op='zero_grad', ref=TRef(ref.st, c, e - c)))
            c = e
        return True

    def _build_backward(self):
        written = {}
        bwd = []
        for rec in reversed(self.fwd):
            op = rec['op']
            if op == 'yolo':
                continue
            if op == 'conv':
                if not rec['bn']:
                    # head conv: gradient arrives as fp32 d(logits) from the loss kernel
                    b = dict(op='head_conv_bwd', fwd=rec)
                    b['dx'] = self._grad_runs(written, rec['x'])
                    bwd.append(b)
                    continue
                if not self._has_grad(written, rec['out'], bwd):
                    continue  # dead branch: nothing downstream reaches the loss
                b = dict(op='conv_bwd', fwd=rec)
                if rec['res'] is not None:
                    runs = self._grad_runs(written, rec['res'])
                    b['res_runs'] = runs
                else:
                    b['res_runs'] = []
                b['dx'] = [] if rec['first'] else self._grad_runs(written, rec['x'])
                bwd.append(b)
            elif op in ('pool', 'upsample', 'copy'):
                if not self._has_grad(written, rec['out'], bwd):
                    continue
                bwd.append(dict(op=op + '_bwd', fwd=rec, dx=self._grad_runs(written, rec['x'])))
            elif op == 'add':
                if not self._has_grad(written, rec['out'], bwd):
                    continue
                bwd.append(dict(op='add_bwd', fwd=rec, da=self._grad_runs(written, rec['a']),
                                db=self._grad_runs(written, rec['b'])))
        self.bwd = bwd
        self._mark_dgrad_bn_sums()

    def _mark_dgrad_bn_sums(self):
        """Which input-gradient launches may take the BatchNorm-backward sums of the layer that produced their output
        tensor (cy_conv_dgrad_bn_sums): the dgrad run of conv P that is the LAST writer of every channel of
        dL/d(out of BN conv L), covers exactly that tensor, with P and L adjacent among the conv backward ops (the engine's
        alternating sum tables rely on that).  Marks ``b_P['dx_sums'][run index] = L's record`` and
        ``b_L['sums_from'] = P's index``; the engine decides per layer whether to use it."""
        last = {}            # (storage id, channel) -> (bwd index, kind, run index) of the latest write
        cat = {}             # (bwd index of P, run index) -> BN conv layers whose whole output gradient that run last writes
        prev_conv = None     # bwd index of the latest conv_bwd / head_conv_bwd
        for bi, b in enumerate(self.bwd):
            op = b['op']
            if op == 'conv_bwd':
                L = b['fwd']
                out = L['out']
                writers = {last.get((out.st.sid, c)) for c in range(out.c0, out.c0 + out.C)}
                b['sums_from'] = None
                w = writers.pop() if len(writers) == 1 else None
                if w is not None and w[1] == 'dgrad' and w[0] == prev_conv:
                    pb = self.bwd[w[0]]
                    ref, _ = pb['dx'][w[2]]
                    # (a stride-2 consumer qualifies too since its four parity classes run as ONE launch)
                    if ref.st is out.st and ref.c0 == out.c0 and ref.C == out.C and L['cout'] % 8 == 0:
                        pb.setdefault('dx_sums', {})[w[2]] = L
                        b['sums_from'] = pb['fwd']['idx']
                if w is not None and w[1] == 'dgrad' and b['sums_from'] is None:
                    # L's output is PART of what that dgrad run writes (a [route] concatenation of several layers' outputs, each
                    # consumed by nothing else): a candidate for sums over the whole run -- see 'dx_sums_cat' below
                    pb = self.bwd[w[0]]
                    ref, _ = pb['dx'][w[2]]
                    if ref.st is out.st and ref.c0 <= out.c0 and out.c0 + out.C <= ref.c0 + ref.C and L['cout'] % 8 == 0:
                        cat.setdefault((w[0], w[2]), []).append(L)
            if op in ('conv_bwd', 'head_conv_bwd'):
                prev_conv = bi
            writes = []
            if op == 'conv_bwd':
                writes += [(ref, 'res', ri) for ri, (ref, _) in enumerate(b['res_runs'])]
            if op in ('conv_bwd', 'head_conv_bwd', 'pool_bwd', 'upsample_bwd', 'copy_bwd'):
                writes += [(ref, 'dgrad' if op.endswith('conv_bwd') else op, ri) for ri, (ref, _) in enumerate(b['dx'])]
            elif op == 'add_bwd':
                writes += [(ref, op, ri) for ri, (ref, _) in enumerate(b['da'] + b['db'])]
            elif op == 'zero_grad':
                writes.append((b['ref'], op, 0))
            for ref, kind, ri in writes:
                for c in range(ref.c0, ref.c0 + ref.C):
                    last[(ref.st.sid, c)] = (bi, kind, ri)
        # 'dx_sums_cat': a stride-1 dgrad run that is the last writer of the output gradients of exactly TWO BatchNorm conv layers
        # with the same activation, side by side (the closing 1x1 conv of a CSP stage over [branch | sibling]: complex_yolov4.cfg's
        # route layers=-1,-7).  The run may take both layers' BatchNorm-backward sums in its epilogue when the engine keeps their
        # pre-BN tensors and their (mean, invstd, scale, shift) vectors side by side; the sums live in a table of their own, so
        # the two layers' backward ops need not follow P directly.  b_P['dx_sums_cat'][run] = (first layer, second layer).
        for (bi, ri), parts in cat.items():
            pb = self.bwd[bi]
            ref, _ = pb['dx'][ri]
            parts = sorted(parts, key=lambda L: L['out'].c0)
            if (len(parts) == 2 and pb['op'] == 'conv_bwd' and pb['fwd']['stride'] == 1 and parts[0]['out'].c0 == ref.c0
                    and parts[0]['out'].c0 + parts[0]['out'].C == parts[1]['out'].c0
                    and parts[1]['out'].c0 + parts[1]['out'].C == ref.c0 + ref.C and parts[0]['act'] == parts[1]['act']
                    and parts[0]['H'] == parts[1]['H'] and parts[0]['W'] == parts[1]['W']
                    and parts[0].get('res') is None and parts[1].get('res') is None):
                pb.setdefault('dx_sums_cat', {})[ri] = (parts[0], parts[1])
