"""Shared test helpers: a small, almost smooth (Mish, two leaky layers) cfg that exercises every block type of the hot path, and a
gradient-agreement statistic that is honest about leaky-ReLU kinks."""
import os
import tempfile

import numpy as np

V4_ANCHORS = "11, 15, 0, 10, 24, 0, 11, 25, 0, 23, 49, 0, 23, 55, 0, 24, 53, 0, 24, 60, 0, 27, 63, 0, 29, 74, 0"


def _conv(f, k, s=1, act='mish', bn=1):
    return ('[convolutional]\n' + ('batch_normalize=1\n' if bn else '') +
            'filters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n' % (f, k, s, act))


def _yolo(mask):
    return '[yolo]\nmask=%s\nanchors=%s\nclasses=3\nnum=9\nignore_thresh=.7\nscale_x_y=1.05\n' % (mask, V4_ANCHORS)


# module index in the comment; grids are S/4 (head 1) and S/2 (head 2)
MINI_BLOCKS = [
    '[net]\nwidth=64\nheight=64\nchannels=3\n',
    _conv(32, 3, 1),                                   # 0
    _conv(32, 3, 2),                                   # 1   S/2
    _conv(64, 3, 2),                                   # 2   S/4
    '[route]\nlayers=-1\ngroups=2\ngroup_id=1\n',      # 3   second half of 2 (view)
    _conv(32, 3, 1),                                   # 4
    '[route]\nlayers=-1,-2\n',                         # 5   cat(4, view 3): the view has to be copied
    _conv(64, 1),                                      # 6
    '[route]\nlayers=-5\n',                            # 7   alias of 2
    _conv(64, 1),                                      # 8
    _conv(32, 1),                                      # 9
    _conv(64, 3),                                      # 10
    '[shortcut]\nfrom=-3\nactivation=linear\n',        # 11  10 + 8 (fused into 10)
    _conv(64, 1),                                      # 12
    '[route]\nlayers=-1,-7\n',                         # 13  cat(12, 6)
    _conv(64, 1),                                      # 14
    '[maxpool]\nstride=1\nsize=5\n',                   # 15
    '[route]\nlayers=-2\n',                            # 16
    '[maxpool]\nstride=1\nsize=9\n',                   # 17
    '[route]\nlayers=-4\n',                            # 18
    '[maxpool]\nstride=1\nsize=13\n',                  # 19
    '[route]\nlayers=-1,-3,-5,-6\n',                   # 20  SPP cat(19, 17, 15, 14)
    _conv(64, 1, act='leaky'),                         # 21
    _conv(30, 1, act='linear', bn=0),                  # 22
    _yolo('3,4,5'),                                    # 23
    '[route]\nlayers=-3\n',                            # 24  -> 21
    _conv(32, 1, act='leaky'),                         # 25
    '[upsample]\nstride=2\n',                          # 26  S/2
    '[route]\nlayers=-1,1\n',                          # 27  cat(26, 1)
    _conv(64, 3),                                      # 28
    _conv(30, 1, act='linear', bn=0),                  # 29
    _yolo('0,1,2'),                                    # 30
]


def mini_cfg_path():
    path = os.path.join(tempfile.gettempdir(), 'cyolo_mini_%d.cfg' % os.getpid())
    with open(path, 'w') as f:
        f.write('\n'.join(MINI_BLOCKS))
    return path


def grad_rel_errors(named_grads, ref_grads):
    """{name: max|g - ref| / max|ref|} over parameter tensors (numpy / torch CPU tensors)."""
    out = {}
    for n, g in named_grads:
        r = ref_grads[n]
        out[n] = float((g - r).abs().max()) / (float(r.abs().max()) + 1e-12)
    return out


def assert_grads_agree(errs, tight=5e-3, frac_tight=0.8, loose=1.0):
    """Gradient parity statistic for nets with leaky-ReLU: a pre-activation that lands within float32 round-off of
    zero takes slope 1 in one evaluation order and 0.1 in another; where that element also carries one of the few
    large head gradients, a whole channel's bias gradient moves by tens of percent (measured on the oracle itself:
    tests/test_plan_sim.py).  So: the median and most tensors must be tight, none may be wrong by more than `loose`."""
    v = np.asarray(list(errs.values()))
    bad = {k: e for k, e in errs.items() if e > loose}
    assert not bad, bad
    assert np.median(v) <= tight, ('median', float(np.median(v)))
    assert (v <= tight).mean() >= frac_tight, ('fraction tight', float((v <= tight).mean()))


def storage_round(dtype):
    """tensor -> tensor: a round trip through ``dtype`` (torch.float16 / torch.bfloat16) whose backward rounds the
    gradient the same way -- 16-bit storage of an activation AND of its gradient, in float32 arithmetic
    (oracle.darknet_ref.DarknetRef.forward(storage_round=...))."""
    import torch

    class _Round(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.to(dtype).to(t.dtype)

        @staticmethod
        def backward(ctx, g):
            return g.to(dtype).to(g.dtype)

    return lambda t: _Round.apply(t) if t.requires_grad else t.to(dtype).to(t.dtype)
