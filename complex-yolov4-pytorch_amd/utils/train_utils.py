"""reduce_tensor / create_optimizer -- the two pieces of reference src/utils/train_utils.py the hot path touches."""
import torch

from ..parallel import reduce_tensor, subdivisions_for  # noqa: F401


def create_optimizer(configs, model):
    """Same three parameter groups as reference train_utils.py:21-50 (biases / conv weights with decay / rest)."""
    m = model.module if hasattr(model, 'module') else model
    pg0, pg1, pg2 = [], [], []
    for k, v in m.named_parameters():
        if '.bias' in k:
            pg2.append(v)
        elif 'conv' in k and '.weight' in k:
            pg1.append(v)
        else:
            pg0.append(v)
    if configs.optimizer_type == 'sgd':
        opt = torch.optim.SGD(pg0, lr=configs.lr, momentum=configs.momentum, nesterov=True)
    elif configs.optimizer_type == 'adam':
        if getattr(configs, 'fused_optimizer', True) and pg0 and pg0[0].is_cuda:
            from ..optim import FusedAdam
            opt = FusedAdam(pg0, lr=configs.lr)
        else:
            opt = torch.optim.Adam(pg0, lr=configs.lr)
    else:
        assert False, "Unknown optimizer type"
    opt.add_param_group({'params': pg1, 'weight_decay': configs.weight_decay})
    opt.add_param_group({'params': pg2})
    return opt
