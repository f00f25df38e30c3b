"""Darknet -- drop-in for reference src/models/darknet2pytorch.py:147-233: ``Darknet(cfgfile, use_giou_loss)``,
``forward(x, targets=None)`` -> ``yolo_outputs`` or ``(loss, yolo_outputs)``, attributes ``blocks``, ``models``
(an ``nn.ModuleList`` with the reference's module / parameter names, so state dicts are interchangeable),
``yolo_layers``, ``width``, ``height``, ``num_classes``, ``print_network()``.

The torch modules in ``self.models`` only OWN the parameters and buffers.  The computation is a plan of HIP
kernels (models/graph.py, models/engine.py): NHWC half-precision (or exact-f32 parity mode) activations,
MFMA implicit-GEMM convolutions with BatchNorm statistics in the epilogue, fused BN+Mish(+shortcut) passes,
routes as channel slices, fused YOLO head loss.  There is no CPU fallback: a CPU input raises.

Differences from the reference that a caller can observe (all deliberate):
  * in training the returned ``yolo_outputs`` stays on the device (the reference copies it to the host every
    step and train.py discards it, #15); with ``targets=None`` it is a CPU tensor as in the reference;
  * parameter gradients are accumulated by the backward kernels straight into one flat fp32 buffer whose
    slices are the parameters' ``.grad`` (this is what the data-parallel wrapper all-reduces);
  * ``dtype='f16'`` (default) / ``'bf16'`` store activations and packed weights in 16 bits and accumulate in fp32
    (bf16 needs no loss scaling); ``dtype='f32'`` is the parity mode;
  * ``deterministic=True`` replaces the fp32-atomic reductions by fixed-order ones: repeats are bit-identical.
"""

import torch
import torch.nn as nn

from .. import ops
from .darknet_utils import load_conv, load_conv_bn, parse_cfg, print_cfg, save_conv, save_conv_bn
from .engine import Arena, Engine
from .graph import Plan, lower_blocks
from .yolo_layer import YoloLayer


class Mish(nn.Module):
    """Parameter-free placeholder keeping the reference's module tree (``mish{n}``); the activation itself
    runs fused in cy_bn_act_fwd (reference darknet2pytorch.py:22-28)."""

    def forward(self, x):  # pragma: no cover - never on the hot path
        raise ops.CyoloError('activation modules are placeholders; run the model through Darknet.forward')


class EmptyModule(nn.Module):
    """Placeholder for route / shortcut / pool / upsample entries of the module list."""

    def __init__(self, kind=''):
        super().__init__()
        self.kind = kind

    def forward(self, x):  # pragma: no cover
        raise ops.CyoloError('structural modules are placeholders; run the model through Darknet.forward')


class _StepFn(torch.autograd.Function):
    """One autograd node for the whole network: forward runs the plan, backward runs the backward plan and
    deposits parameter gradients into the model's flat gradient buffer (so it returns None for them)."""

    @staticmethod
    def forward(ctx, model, x, targets, *params):
        eng = model._engine_for(x)
        model._weights_epoch += 1            # a training step is under way: cached eval packs are stale from here on
        outputs = eng.forward(x, targets, model._param_table(), model.use_giou_loss, x.shape[2])
        model._publish_metrics(eng)
        heads = [m[0] for m in eng.metrics]
        total = heads[0].clone()
        for h in heads[1:]:
            total += h
        ctx.model, ctx.eng, ctx.fwd_serial = model, eng, eng.fwd_serial
        outputs = outputs.clone()            # the engine's buffer is overwritten by its next forward
        ctx.mark_non_differentiable(outputs)
        return (total.reshape(1) if model.use_giou_loss else total), outputs

    @staticmethod
    def backward(ctx, gloss, _gout):
        model, eng = ctx.model, ctx.eng
        if not eng.training:
            raise ops.CyoloError('backward needs model.train(): eval-mode engines keep no activations')
        if eng.fwd_serial != ctx.fwd_serial:
            raise ops.CyoloError('backward() of a forward whose saved activations were overwritten: the engine for this '
                                 'input shape has run another forward since (one forward -> one backward per shape)')
        accumulating = model._plist is not None and any(p.grad is not None for _, p in model._plist)
        for hook in model._pre_backward_hooks:
            hook(model, accumulating)
        grads = model._grad_table()
        eng.backward(grads, gloss.detach().reshape(-1).float().contiguous(), model.loss_scale / model.grad_prescale, act_scale=model.loss_scale,
                     on_module_done=(lambda idx: [h(model, idx) for h in model._module_grad_hooks]) if model._module_grad_hooks else None)
        for hook in model._post_backward_hooks:
            hook(model)
        return (None, None, None) + (None,) * len(model._plist)


class Darknet(nn.Module):
    def __init__(self, cfgfile, use_giou_loss, dtype='f16', loss_scale=None, deterministic=False):
        super(Darknet, self).__init__()
        self.use_giou_loss = use_giou_loss
        self.blocks = parse_cfg(cfgfile)
        self.width = int(self.blocks[0]['width'])
        self.height = int(self.blocks[0]['height'])
        self.dtype_code = ops.dtype_code(dtype)
        self.deterministic = bool(deterministic)     # fixed-order reductions: bit-identical repeats (see Engine)
        # static scale applied to d(logits) before it enters the fp16 backward and removed in the parameter-gradient
        # reductions.  Default 1: at random init the gradients are LARGE (a scale of 1024 overflows fp16 on
        # complex_yolov4.cfg); measured on the mini cfg the gradient error does not depend on it between 1 and 1024.
        self.loss_scale = float(loss_scale if loss_scale is not None else 1.0)
        self.models = self.create_network(self.blocks)
        self.yolo_layers = [layer for layer in self.models if layer.__class__.__name__ == 'YoloLayer']
        self.loss = self.models[len(self.models) - 1]
        self.header = torch.IntTensor([0, 0, 0, 0])
        self.seen = 0
        self.cpu_outputs = True          # targets=None returns a CPU tensor like the reference (:228)
        import collections
        self._plans, self._engines = {}, collections.OrderedDict()      # engines in least-recently-used order (see _engine_for)
        # Byte budget of the engine cache.  The reference changes img_size by +-96 every 10 batches and doubles it under mosaic
        # (kitti_dataset.py:42-43,144,225-230): at batch 16 the seven mosaic geometries are ~7 x 54 GB of engines -- more than
        # the GPU has.  Engines beyond the budget are evicted least-recently-used first, recorded launch lists and tuned
        # choices (process-wide memo, persisted table) surviving in tune.py.  None: 60 % of the device's memory at first use;
        # CY_ENGINE_BUDGET_GB overrides.
        self.engine_budget_bytes = None
        self.engine_evictions = 0
        self._plist = None
        self._grad_flat = None
        self._post_backward_hooks = []
        self._pre_backward_hooks = []    # hook(model, accumulating) before the backward plan touches the flat gradient
        # eval engines keep packed half-precision weights between forwards; they re-pack when this counter moved (every
        # training forward, load_state_dict, load_weights, .to()/.half()-style _apply, mark_weights_dirty) -- and on every
        # eval forward unless static_eval_weights is set (serving: nobody updates parameters behind the model's back)
        self._weights_epoch = 0
        self.static_eval_weights = False
        self._module_grad_hooks = []     # called as hook(model, module_idx) when a module's gradients are final
        self.grad_prescale = 1.0         # folded into every parameter-gradient reduction (1/world under data parallelism)

    # ---- construction (reference create_network :235-401) -----------------------------------------
    def create_network(self, blocks):
        models = nn.ModuleList()
        for m in lower_blocks(blocks):
            t = m['type']
            if t == 'convolutional':
                n = m['n']
                seq = nn.Sequential()
                seq.add_module('conv%d' % n, nn.Conv2d(m['cin'], m['cout'], m['k'], m['stride'], m['pad'], bias=not m['bn']))
                if m['bn']:
                    seq.add_module('bn%d' % n, nn.BatchNorm2d(m['cout']))
                if m['act'] == 'leaky':
                    seq.add_module('leaky%d' % n, nn.LeakyReLU(0.1, inplace=True))
                elif m['act'] == 'mish':
                    seq.add_module('mish%d' % n, Mish())
                elif m['act'] != 'linear':
                    raise ValueError('unsupported activation %r' % m['act'])
                models.append(seq)
            elif t == 'yolo':
                self.num_classes = m['classes']
                models.append(YoloLayer(num_classes=m['classes'], anchors=m['anchors'], stride=0,
                                        scale_x_y=m['scale_x_y'], ignore_thresh=m['ignore_thresh']))
            else:
                models.append(EmptyModule(t))
        return models

    def print_network(self):
        print_cfg(self.blocks)

    def mark_weights_dirty(self):
        """Tell cached eval engines that parameters changed outside Darknet's sight (an in-place edit, an optimizer
        stepping without a training forward of this module).  Only needed with ``static_eval_weights = True``."""
        self._weights_epoch += 1

    def load_state_dict(self, *args, **kwargs):
        self._weights_epoch += 1
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._weights_epoch = getattr(self, '_weights_epoch', 0) + 1
        return super()._apply(fn, *args, **kwargs)

    def load_weights(self, weightfile):
        """Darknet ``.weights`` -> fp32 master parameters and BN running statistics (reference :403-451: int32[5] header,
        ``seen = header[3]``, then per [convolutional] block in cfg order the flat float32 tensors; a short file stops
        the walk at the block where it runs out, as the reference's ``start >= buf.size`` check does)."""
        import numpy as np
        with open(weightfile, 'rb') as fp:
            header = np.fromfile(fp, count=5, dtype=np.int32)
            buf = np.fromfile(fp, dtype=np.float32)
        self.header = torch.from_numpy(header)
        self.seen = self.header[3]
        self._weights_epoch += 1
        start, ind = 0, -2
        for block in self.blocks:
            if start >= buf.size:
                break
            ind += 1
            if block['type'] != 'convolutional':
                continue
            model = self.models[ind]
            if int(block['batch_normalize']):
                start = load_conv_bn(buf, start, model[0], model[1])
            else:
                start = load_conv(buf, start, model[0])
        return start

    def save_weights(self, outfile, cutoff=0):
        """Inverse of load_weights (the reference ships the per-layer writers save_conv_bn / save_conv,
        darknet_utils.py:209-246, but no model-level method): header then every [convolutional] block up to ``cutoff``
        (0 = all) in the reference's tensor order."""
        import numpy as np
        header = np.zeros(5, dtype=np.int32)
        h = np.asarray(self.header).astype(np.int32).ravel()
        header[:min(5, h.size)] = h[:5]
        header[3] = int(self.seen)
        last = len(self.blocks) - 1 if cutoff <= 0 else cutoff
        with open(outfile, 'wb') as fp:
            header.tofile(fp)
            ind = -2
            for block in self.blocks[:last + 1]:
                ind += 1
                if block['type'] != 'convolutional':
                    continue
                model = self.models[ind]
                if int(block['batch_normalize']):
                    save_conv_bn(fp, model[0], model[1])
                else:
                    save_conv(fp, model[0])

    # ---- plumbing ---------------------------------------------------------------------------------
    def _param_table(self):
        t = {k: v.data for k, v in self.named_parameters()}
        t.update({k: v for k, v in self.named_buffers()})
        return t

    def _engine_for(self, x):
        N, _, H, W = x.shape
        pk = (H, W)
        if pk not in self._plans:
            self._plans[pk] = Plan(self.blocks, H, W, ops.chunk(self.dtype_code))
        ek = (N, H, W, self.training, str(x.device), self.deterministic)
        eng = self._engines.get(ek)
        if eng is not None:
            self._engines.move_to_end(ek)
            return eng
        try:
            eng = Engine(self._plans[pk], N, self.dtype_code, x.device, self.training, deterministic=self.deterministic)
        except torch.OutOfMemoryError:
            # the new geometry does not fit beside the cached ones: drop them all and try once more
            self.engine_evictions += len(self._engines)
            self.release_engines()
            eng = Engine(self._plans[pk], N, self.dtype_code, x.device, self.training, deterministic=self.deterministic)
        self._engines[ek] = eng
        self._trim_engines(x.device)
        return eng

    def _engine_budget(self, device):
        if self.engine_budget_bytes is None and getattr(device, 'type', str(device)) == 'cuda':
            import os
            gb = os.environ.get('CY_ENGINE_BUDGET_GB')
            self.engine_budget_bytes = int(float(gb) * (1 << 30)) if gb else int(0.6 * torch.cuda.get_device_properties(device).total_memory)
        return self.engine_budget_bytes

    def engine_bytes(self):
        """Device bytes held by the cached engines."""
        return sum(e.nbytes() for e in self._engines.values())

    def _trim_engines(self, device):
        """Evict least-recently-used engines (never the newest) until the cache fits its byte budget; an engine a captured
        hipGraph points into (graphed.GraphedTrainStep sets pin_retired) is never evicted."""
        budget = self._engine_budget(device)
        if not budget or len(self._engines) < 2:
            return
        sizes = {k: e.nbytes() for k, e in self._engines.items()}
        total, newest, dropped = sum(sizes.values()), next(reversed(self._engines)), False
        for k in list(self._engines):
            if total <= budget:
                break
            if k == newest or self._engines[k].pin_retired:
                continue
            total -= sizes[k]
            del self._engines[k]
            self.engine_evictions += 1
            dropped = True
        if dropped and getattr(device, 'type', str(device)) == 'cuda':
            torch.cuda.empty_cache()      # hand the evicted engines' blocks back: the next geometry's buffers have other sizes

    def release_engines(self):
        """Drop cached device storages (e.g. after multiscale training changed resolution)."""
        self._engines.clear()

    def _publish_metrics(self, eng):
        for layer, met in zip(self.yolo_layers, eng.metrics):
            layer._metrics_dev = met

    def _grad_table(self):
        """{name: fp32 view into the flat gradient buffer}; attaches the views as ``.grad``.  A parameter whose
        ``.grad`` is None (zero_grad(set_to_none=True)) gets its slice zeroed; existing values are accumulated
        into, which is what gradient accumulation over sub-divisions needs (reference train.py:212-221)."""
        named = list(self.named_parameters())
        if self._plist is None or len(self._plist) != len(named):
            self._plist = named
        if self._grad_flat is None or self._grad_flat.device != named[0][1].device:
            total = sum(p.numel() for _, p in named)
            dev = named[0][1].device
            self._grad_arena = Arena(dev, Engine.GUARD_BYTES if dev.type == 'cuda' else 0)      # (red zones: tests/test_gpu_redzone.py)
            self._grad_flat = self._grad_arena.new('flat_grad', total, torch.float32, zero=True)
            self._grad_views, off = {}, 0
            for name, p in named:
                self._grad_views[name] = self._grad_flat[off:off + p.numel()].view_as(p)
                off += p.numel()
            for _, p in named:
                p.grad = None
        if all(p.grad is None for _, p in named):
            self._grad_flat.zero_()
            for name, p in named:
                p.grad = self._grad_views[name]
        else:
            for name, p in named:
                v = self._grad_views[name]
                if p.grad is None:
                    v.zero_()
                    p.grad = v
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    p.grad = v
        return self._grad_views

    @property
    def flat_grad(self):
        return self._grad_flat

    # ---- forward (reference :162-230) -----------------------------------------------------------
    def forward(self, x, targets=None):
        ops.check_device_tensor(x, 'Darknet')
        x = x.float().contiguous()
        if targets is None:
            eng = self._engine_for(x)
            out = eng.forward(x, None, self._param_table(), self.use_giou_loss, x.shape[2],
                              weights_epoch=self._weights_epoch if self.static_eval_weights else None)
            return out.cpu() if self.cpu_outputs else out.clone()
        targets = targets.to(x.device).float().contiguous()
        if self._plist is None:
            self._plist = list(self.named_parameters())
        loss, outputs = _StepFn.apply(self, x, targets, *[p for _, p in self._plist])
        return loss, outputs
