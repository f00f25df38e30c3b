"""ORACLE (test infrastructure).  The Darknet graph on plain PyTorch-CPU float32 ops.

Restates reference src/models/darknet2pytorch.py:
  create_network :235-401  (conv pad=(k-1)//2 if pad else 0; bias only without BN; module/param names
                            models.{i}.conv{n} / bn{n}; BN = torch defaults eps 1e-5 momentum 0.1)
  forward        :162-230  (route / grouped route / cat, shortcut add, maxpool, expand-upsample, yolo)
  MaxPoolDark    :30-59    (replicate-padded pool for size / stride pairs nn.MaxPool2d is not used for)
  Mish           :22-28    (x * tanh(softplus(x)))
as a functional graph walk over a {state-dict-name: tensor} dict, so gradients are available by name.
The three/two YoloLayer heads use oracle/yolo_layer_ref.py.  This is also what bench.py times as
``cpu_baseline`` (kind "port").
"""
import math

import torch
import torch.nn.functional as F

from . import yolo_layer_ref


def _layers(blocks):
    """Module list description: one dict per non-[net] block, with resolved absolute indices."""
    mods = []
    conv_id = 0
    ch = int(blocks[0].get('channels', 3))
    out_ch = []
    for blk in blocks[1:]:
        i = len(mods)
        t = blk['type']
        m = dict(type=t)
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
                     ignore_thresh=float(blk['ignore_thresh']))
        else:
            raise ValueError('oracle: unsupported block type %r' % t)
        out_ch.append(ch)
        mods.append(m)
    return mods


class DarknetRef:
    def __init__(self, blocks):
        self.blocks = blocks
        self.mods = _layers(blocks)

    def param_shapes(self):
        """Ordered {name: shape} of parameters, and {name: shape} of buffers (state-dict names)."""
        params, bufs = {}, {}
        for i, m in enumerate(self.mods):
            if m['type'] != 'convolutional':
                continue
            n = m['n']
            params['models.%d.conv%d.weight' % (i, n)] = (m['cout'], m['cin'], m['k'], m['k'])
            if m['bn']:
                params['models.%d.bn%d.weight' % (i, n)] = (m['cout'],)
                params['models.%d.bn%d.bias' % (i, n)] = (m['cout'],)
                bufs['models.%d.bn%d.running_mean' % (i, n)] = (m['cout'],)
                bufs['models.%d.bn%d.running_var' % (i, n)] = (m['cout'],)
            else:
                params['models.%d.conv%d.bias' % (i, n)] = (m['cout'],)
        return params, bufs

    def forward(self, params, x, targets=None, use_giou_loss=True, training=True, bufs=None,
                keep=None, storage_round=None):
        """Returns (outputs[B,N,7+C], loss or None, [metrics per head]).  ``keep`` (a dict) receives
        intermediate activations by module index when given.  ``storage_round`` (a function tensor -> tensor, e.g. a
        float16 round trip) models 16-BIT STORAGE in this float32 arithmetic: it is applied to every convolution's input,
        weight and -- for BatchNorm layers -- output (the tensors a half-precision implementation keeps in memory; the
        head logits stay float32).  The tests use it to tell what 16-bit rounding alone does to the reference's own
        function from what a kernel adds."""
        rnd = storage_round if storage_round is not None else (lambda t: t)
        img_size = x.shape[2]
        outs = {}
        heads, metrics = [], []
        loss = 0.
        for i, m in enumerate(self.mods):
            t = m['type']
            if t == 'convolutional':
                n = m['n']
                w = params['models.%d.conv%d.weight' % (i, n)]
                b = params.get('models.%d.conv%d.bias' % (i, n))
                if storage_round is not None:      # 16-bit weight COPIES of float32 masters: the gradient stays float32
                    w = w + (rnd(w.detach()) - w.detach())
                x = F.conv2d(rnd(x), w, b, m['stride'], m['pad'])
                if m['bn'] and storage_round is not None:
                    x = rnd(x)
                if keep is not None:
                    keep[('raw', i)] = x
                if m['bn']:
                    g = params['models.%d.bn%d.weight' % (i, n)]
                    be = params['models.%d.bn%d.bias' % (i, n)]
                    rm = rv = None
                    if bufs is not None:
                        rm = bufs['models.%d.bn%d.running_mean' % (i, n)]
                        rv = bufs['models.%d.bn%d.running_var' % (i, n)]
                    if training or rm is None:
                        x = F.batch_norm(x, rm, rv, g, be, True, 0.1, 1e-5)
                    else:
                        x = F.batch_norm(x, rm, rv, g, be, False, 0.1, 1e-5)
                if m['act'] == 'mish':
                    x = x * torch.tanh(F.softplus(x))
                elif m['act'] == 'leaky':
                    x = F.leaky_relu(x, 0.1)
            elif t == 'maxpool':
                k, s = m['k'], m['stride']
                if s == 1 and k % 2:
                    x = F.max_pool2d(x, k, s, k // 2)
                elif s == k:
                    x = F.max_pool2d(x, k, s, 0)
                else:
                    # MaxPoolDark (reference darknet2pytorch.py:30-59; complex_yolov3_tiny.cfg's size=2 stride=1 pool)
                    p = k // 2
                    pads = []
                    for n in (x.shape[3], x.shape[2]):          # F.pad order: (left, right, top, bottom)
                        p1 = (k - 1) // 2
                        pads += [p1, p1 + 1 if ((n - 1) // s) != ((n + 2 * p - k) // s) else p1]
                    x = F.max_pool2d(F.pad(x, pads, mode='replicate'), k, stride=s)
            elif t == 'upsample':
                x = x.repeat_interleave(m['stride'], 2).repeat_interleave(m['stride'], 3)
            elif t == 'route':
                parts = [outs[s] for s in m['src']]
                if len(parts) == 1:
                    x = parts[0]
                    if m['groups'] > 1:
                        c = x.shape[1] // m['groups']
                        x = x[:, c * m['group_id']:c * (m['group_id'] + 1)]
                else:
                    x = torch.cat(parts, 1)
            elif t == 'shortcut':
                x = outs[m['src']] + outs[i - 1]
                if m['act'] == 'leaky':
                    x = F.leaky_relu(x, 0.1)
            elif t == 'yolo':
                o, l, met = yolo_layer_ref.head_forward(x, targets, m['anchors'], m['classes'],
                                                        m['ignore_thresh'], img_size, use_giou_loss)
                heads.append(o)
                metrics.append(met)
                loss = loss + l
                if keep is not None:
                    keep[('head_in', i)] = x
                continue
            outs[i] = x
            if keep is not None:
                keep[i] = x
        outputs = torch.cat(heads, 1)
        return outputs, (None if targets is None else loss), metrics
