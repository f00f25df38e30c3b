"""reduce_tensor / create_optimizer / checkpoint helpers of reference src/utils/train_utils.py."""
import copy
import os

import torch

from ..parallel import reduce_tensor, subdivisions_for  # noqa: F401


def create_optimizer(configs, model):
    """The reference's three parameter groups (train_utils.py:21-50): everything that is neither a bias nor a conv weight
    first (BatchNorm scales), then conv weights with ``configs.weight_decay``, then biases -- as the fused multi-tensor
    optimizers (one HIP launch per step).  There is no silent fallback: parameters that are not on the HIP device raise,
    unless the caller asks for the stock ``torch.optim`` classes with ``configs.fused_optimizer = False`` (what the CPU
    plumbing tests do; torch.optim then also serves device parameters, at 327 launches per step)."""
    from .. import ops
    net = getattr(model, 'module', model)
    groups = {'other': [], 'conv_weight': [], 'bias': []}
    for name, param in net.named_parameters():
        kind = 'bias' if '.bias' in name else ('conv_weight' if ('conv' in name and '.weight' in name) else 'other')
        groups[kind].append(param)
    first = groups['other']
    fused = bool(getattr(configs, 'fused_optimizer', True))
    if fused and not (first and first[0].is_cuda):
        raise ops.CyoloError('create_optimizer: the fused optimizers run on the HIP device only -- move the model to the '
                             'device first (model.to(device)) or set configs.fused_optimizer = False for torch.optim')
    if configs.optimizer_type == 'adam':
        from ..optim import FusedAdam
        opt = (FusedAdam if fused else torch.optim.Adam)(first, lr=configs.lr)
    elif configs.optimizer_type == 'sgd':
        from ..optim import FusedSGD
        opt = (FusedSGD if fused else torch.optim.SGD)(first, lr=configs.lr, momentum=configs.momentum, nesterov=True)
    else:
        raise AssertionError('Unknown optimizer type')
    opt.add_param_group({'params': groups['conv_weight'], 'weight_decay': configs.weight_decay})
    opt.add_param_group({'params': groups['bias']})
    return opt


def get_saved_state(model, optimizer, lr_scheduler, epoch, configs):
    """(model state dict, {'epoch', 'configs', 'optimizer', 'lr_scheduler'}) as reference train_utils.py:80-93.  The model
    state dict holds the fp32 master parameters under the reference's keys, so the checkpoint loads into the reference."""
    m = model.module if hasattr(model, 'module') else model
    utils_state_dict = {'epoch': epoch, 'configs': configs, 'optimizer': copy.deepcopy(optimizer.state_dict()),
                        'lr_scheduler': copy.deepcopy(lr_scheduler.state_dict())}
    return m.state_dict(), utils_state_dict


def save_checkpoint(checkpoints_dir, saved_fn, model_state_dict, utils_state_dict, epoch):
    """Model_<fn>_epoch_<e>.pth and Utils_<fn>_epoch_<e>.pth, as reference train_utils.py:96-104."""
    model_save_path = os.path.join(checkpoints_dir, 'Model_{}_epoch_{}.pth'.format(saved_fn, epoch))
    utils_save_path = os.path.join(checkpoints_dir, 'Utils_{}_epoch_{}.pth'.format(saved_fn, epoch))
    torch.save(model_state_dict, model_save_path)
    torch.save(utils_state_dict, utils_save_path)
    print('save a checkpoint at {}'.format(model_save_path))
