"""FusedAdam / FusedSGD: torch.optim.Adam / torch.optim.SGD semantics, one HIP launch per step for the whole model
(cy_adam_multi / cy_sgd_multi).

SURVEY.md section 8f #3 ("next": optimizer step fused).  Same constructor / param_groups / state_dict surface as
torch.optim.Adam for the options the reference uses (reference src/utils/train_utils.py:21-50: Adam(lr) with a
weight-decay group for conv weights); LambdaLR schedulers work because the learning rate is read from the param
groups every step.  Gradients are expected where Darknet puts them: slices of its flat fp32 gradient buffer."""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._table = self._key = None
        self._steps = 0

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
        key = tuple((i[0].data_ptr(), i[1].data_ptr(), i[4]) for i in items)
        if key != self._key:
            self._table = ops.make_adam_table(items, items[0][0].device) if items else None
            self._key = key

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        loss = closure() if closure is not None else None
        self._build()
        if self._table is None:
            return loss
        self._steps += 1
        b1, b2 = self.param_groups[0]['betas']
        for g in self.param_groups:
            assert tuple(g['betas']) == (b1, b2) and g['eps'] == self.param_groups[0]['eps'], 'betas/eps must be shared'
        for st in self.state.values():
            st['step'] = self._steps
        assert len(self.param_groups) <= 8
        ops.adam_multi(self._table[0], self._table[1], b1, b2, self.param_groups[0]['eps'], 1 - b1 ** self._steps,
                       1 - b2 ** self._steps, [g['lr'] for g in self.param_groups],
                       [g['weight_decay'] for g in self.param_groups], zero_grad=zero_grad)
        return loss


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(lr, momentum, nesterov, weight_decay) with dampening 0 in one launch (cy_sgd_multi); the
    reference's ``optimizer_type == 'sgd'`` choice (train_utils.py:35-37)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, nesterov=False, weight_decay=0.0):
        if nesterov and momentum <= 0:
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        super().__init__(params, dict(lr=lr, momentum=momentum, nesterov=nesterov, weight_decay=weight_decay, dampening=0))
        self._table = self._key = None
        self._steps = 0

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
        key = tuple((i[0].data_ptr(), i[1].data_ptr(), i[4]) for i in items)
        if key != self._key:
            self._table = ops.make_adam_table(items, items[0][0].device) if items else None
            self._key = key
        return fresh

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        loss = closure() if closure is not None else None
        fresh = self._build()
        if self._table is None:
            return loss
        g0 = self.param_groups[0]
        for g in self.param_groups:
            assert g['momentum'] == g0['momentum'] and g['nesterov'] == g0['nesterov'], 'momentum/nesterov must be shared'
        assert len(self.param_groups) <= 8
        first = self._steps == 0 or fresh
        self._steps += 1
        ops.sgd_multi(self._table[0], self._table[1], g0['momentum'], g0['nesterov'], first,
                      [g['lr'] for g in self.param_groups], [g['weight_decay'] for g in self.param_groups], zero_grad=zero_grad)
        return loss
