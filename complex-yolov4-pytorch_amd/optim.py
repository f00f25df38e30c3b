"""FusedAdam / FusedSGD: torch.optim.Adam / torch.optim.SGD semantics, one HIP launch per step for the whole model
(cy_adam_multi / cy_sgd_multi).

SURVEY.md section 8f #3 ("next": optimizer step fused).  Same constructor / param_groups / state_dict surface as
torch.optim.Adam for the options the reference uses (reference src/utils/train_utils.py:21-50: Adam(lr) with a
weight-decay group for conv weights); LambdaLR schedulers work because the learning rate is read from the param
groups every step.  Gradients are expected where Darknet puts them: slices of its flat fp32 gradient buffer."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    """``capturable=True`` (the name torch.optim.Adam uses for the same purpose): nothing that changes from step to step is
    passed to the kernel by value -- the step count lives in a device int32 advanced by the launch itself, the per-group
    learning rates / weight decays in a device array refreshed from a pinned host copy -- so ``step()`` can be captured in a
    hipGraph (graphed.GraphedTrainStep) and replayed while LR schedulers keep editing ``param_groups``."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = bool(capturable)
        self._counter = self._grp_dev = self._grp_host = None
        self._table = self._key = None
        self._steps = 0
        self._step_dev = None          # int32 [2] ping-pong step counter, used while a device skip flag is attached
        self._ping = 0

    def load_state_dict(self, state_dict):
        """torch.optim semantics plus what the fused step keeps outside ``state``: the bias-correction step count is
        restored from the per-parameter ``step`` entries (a resumed run continues at t+1, not at t=1) and the kernel
        table is rebuilt around the loaded moment tensors."""
        super().load_state_dict(state_dict)
        self._resync()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._resync()

    def _resync(self):
        steps = [int(st['step']) for st in self.state.values() if 'step' in st]
        self._steps = max(steps) if steps else 0
        self._table = self._key = None
        self._step_dev = None
        self._counter = None

    def note_skipped(self):
        """A step the device skip flag suppressed (DynamicLossScale found a non-finite gradient) did not happen.  The
        KERNEL already knew: with a skip flag attached the step count that feeds the bias corrections lives on the device
        and only advances on applied steps (cy_adam_multi_dev), so no step ever runs with t + 1 in place of t.  This
        only brings the host's copy (the ``step`` entries of ``state_dict()``) back in line, one step late."""
        self._steps = max(0, self._steps - 1)
        for st in self.state.values():
            st['step'] = self._steps

    def refresh_groups(self):
        """param_groups' (lr, weight_decay) -> the pinned host array the captured copy reads (capturable mode)."""
        for i, g in enumerate(self.param_groups):
            self._grp_host[i] = float(g['lr'])
            self._grp_host[8 + i] = float(g['weight_decay'])

    def note_replayed(self):
        """A captured step() was replayed by a graph (graphed.GraphedTrainStep): the device counter advanced by one, the
        host's copies (``_steps`` and the per-parameter ``step`` entries a checkpoint carries) follow."""
        self._steps += 1
        for st in self.state.values():
            st['step'] = self._steps

    def state_dict(self):
        for st in self.state.values():          # (whatever path advanced _steps last: a checkpoint never carries a stale count)
            if 'step' in st:
                st['step'] = self._steps
        return super().state_dict()

    def _build(self):
        items = []
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if 'exp_avg' not in st:
                    st['exp_avg'] = torch.zeros_like(p.data)
                    st['exp_avg_sq'] = torch.zeros_like(p.data)
                    st['step'] = 0
                items.append((p.data, p.grad, st['exp_avg'], st['exp_avg_sq'], gi))
        key = tuple((i[0].data_ptr(), i[1].data_ptr(), i[2].data_ptr(), i[3].data_ptr(), i[4]) for i in items)
        if key != self._key:
            self._table = ops.make_adam_table(items, items[0][0].device) if items else None
            self._key = key

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        loss = closure() if closure is not None else None
        self._build()
        if self._table is None:
            return loss
        # while a hipGraph is being CAPTURED no kernel runs: the step that the capture records happens at each replay
        # (note_replayed), so the host's step count must not move here
        capturing = self.capturable and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self._steps += 1
        b1, b2 = self.param_groups[0]['betas']
        for g in self.param_groups:
            assert tuple(g['betas']) == (b1, b2) and g['eps'] == self.param_groups[0]['eps'], 'betas/eps must be shared'
        if not capturing:
            for st in self.state.values():
                st['step'] = self._steps
        assert len(self.param_groups) <= 8
        skip = getattr(self, 'skip_flag', None)
        lrs, wds = [g['lr'] for g in self.param_groups], [g['weight_decay'] for g in self.param_groups]
        if self.capturable:
            dev = self._table[0].device
            if self._counter is None:
                if capturing:
                    raise ops.CyoloError('FusedAdam(capturable=True): run one eager step() before capturing it (the device step '
                                         'counter would be created, and reset at every replay, inside the graph)')
                self._counter = torch.full((1,), self._steps - 1, dtype=torch.int32, device=dev)
                self._grp_dev = torch.zeros(16, dtype=torch.float32, device=dev)
                self._grp_host = torch.zeros(16, dtype=torch.float32).pin_memory()
            self.refresh_groups()
            self._grp_dev.copy_(self._grp_host, non_blocking=True)     # (captured: re-read from the pinned copy at every replay)
            ops.adam_multi_graph(self._table[0], self._table[1], b1, b2, self.param_groups[0]['eps'], self._counter, self._grp_dev,
                                 zero_grad=zero_grad, skip_flag=skip)
            return loss
        if skip is not None:
            if self._step_dev is None:       # first step under a skip flag (or after a resume): seed the device counter
                self._step_dev = torch.full((2,), self._steps - 1, dtype=torch.int32, device=skip.device)
                self._ping = 0
            i = self._ping
            ops.adam_multi_dev(self._table[0], self._table[1], b1, b2, self.param_groups[0]['eps'], self._step_dev[i:i + 1],
                               self._step_dev[1 - i:2 - i], lrs, wds, zero_grad=zero_grad, skip_flag=skip)
            self._ping = 1 - i
            return loss
        ops.adam_multi(self._table[0], self._table[1], b1, b2, self.param_groups[0]['eps'], 1 - b1 ** self._steps,
                       1 - b2 ** self._steps, lrs, wds, zero_grad=zero_grad)
        return loss


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum, nesterov, weight_decay) with dampening 0 in one launch (cy_sgd_multi); the
    reference's ``optimizer_type == 'sgd'`` choice (train_utils.py:35-37)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, nesterov=False, weight_decay=0.0):
        if nesterov and momentum <= 0:
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        super().__init__(params, dict(lr=lr, momentum=momentum, nesterov=nesterov, weight_decay=weight_decay, dampening=0))
        self._table = self._key = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = self._key = None      # the loaded momentum buffers are new tensors

    def __setstate__(self, state):
        super().__setstate__(state)
        self._table = self._key = None

    def note_skipped(self):
        pass

    def _build(self):
        items, fresh = [], False
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if 'momentum_buffer' not in st:
                    st['momentum_buffer'] = torch.zeros_like(p.data)
                    fresh = True
                items.append((p.data, p.grad, st['momentum_buffer'], st['momentum_buffer'], gi))
        key = tuple((i[0].data_ptr(), i[1].data_ptr(), i[2].data_ptr(), i[4]) for i in items)
        if key != self._key:
            self._table = ops.make_adam_table(items, items[0][0].device) if items else None
            self._key = key
        return fresh

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        loss = closure() if closure is not None else None
        self._build()
        if self._table is None:
            return loss
        g0 = self.param_groups[0]
        for g in self.param_groups:
            assert g['momentum'] == g0['momentum'] and g['nesterov'] == g0['nesterov'], 'momentum/nesterov must be shared'
        assert len(self.param_groups) <= 8
        # torch's first step sets buf = g; with the zero-initialised buffers of _build that IS buf = momentum * buf + g,
        # so no step is special -- and a momentum buffer restored by load_state_dict is never overwritten
        ops.sgd_multi(self._table[0], self._table[1], g0['momentum'], g0['nesterov'], False,
                      [g['lr'] for g in self.param_groups], [g['weight_decay'] for g in self.param_groups], zero_grad=zero_grad,
                      skip_flag=getattr(self, 'skip_flag', None))
        return loss


class DynamicLossScale:
    """Dynamic loss scaling for the fp16 backward (the role torch.cuda.amp.GradScaler plays for autocast models).

    ``Darknet.loss_scale`` multiplies d(logits) before the half-precision backward pass and is divided out again in the
    fp32 parameter-gradient reductions, so only the fp16 activation gradients see it.  Per step:

        loss.backward();  scaler.check();  optimizer.step();  scaler.update()

    ``check`` runs cy_grad_nonfinite over the model's flat gradient; the fused optimizers read the resulting device flag
    and skip the whole step when it is set -- no host synchronisation on the step path.  ``update`` starts an
    asynchronous copy of the flag and acts on the copy started one step earlier: after an overflow the scale is multiplied by
    ``backoff_factor``, after ``growth_interval`` clean steps by ``growth_factor`` (one step late, which only delays
    the adjustment)."""

    def __init__(self, model, optimizer, init_scale=1.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 max_scale=65536.0):
        self.model = model.module if hasattr(model, 'module') else model
        self.optimizer = optimizer
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval, self.max_scale = int(growth_interval), float(max_scale)
        self.model.loss_scale = float(init_scale)
        self._flag = self._host = self._event = None
        self._clean = 0
        self.skipped = 0

    @property
    def scale(self):
        return self.model.loss_scale

    def check(self):
        flat = self.model.flat_grad
        if flat is None:
            return
        if self._flag is None:
            self._flag = torch.zeros(1, dtype=torch.int32, device=flat.device)
            self._host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.optimizer.skip_flag = self._flag
        ops.grad_nonfinite(flat, self._flag)

    def update(self):
        if self._flag is None:
            return
        if self._event is not None:               # result of the PREVIOUS step's check
            self._event.synchronize()
            if int(self._host[0]):
                self.model.loss_scale = max(self.model.loss_scale * self.backoff_factor, 2.0 ** -14)
                self._clean = 0
                self.skipped += 1
                if hasattr(self.optimizer, 'note_skipped'):
                    self.optimizer.note_skipped()
            else:
                self._clean += 1
                if self._clean >= self.growth_interval:
                    self.model.loss_scale = min(self.model.loss_scale * self.growth_factor, self.max_scale)
                    self._clean = 0
        self._host.copy_(self._flag, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
