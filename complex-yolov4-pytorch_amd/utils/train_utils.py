"""reduce_tensor / create_optimizer / checkpoint helpers of reference src/utils/train_utils.py."""
import copy
import os

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
        if getattr(configs, 'fused_optimizer', True) and pg0 and pg0[0].is_cuda:
            from ..optim import FusedSGD
            opt = FusedSGD(pg0, lr=configs.lr, momentum=configs.momentum, nesterov=True)
        else:
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
