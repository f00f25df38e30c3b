"""create_model / get_num_parameters / make_data_parallel -- drop-ins for reference src/models/model_utils.py.
``make_data_parallel`` swaps torch DDP for the RCCL flat-buffer wrapper in ``parallel.py``."""
import torch

from .darknet2pytorch import Darknet


def create_model(configs):
    """reference model_utils.py:20-28"""
    if (configs.arch == 'darknet') and (configs.cfgfile is not None):
        print('using darknet')
        model = Darknet(cfgfile=configs.cfgfile, use_giou_loss=configs.use_giou_loss,
                        dtype=getattr(configs, 'dtype', 'f16'))
    else:
        assert False, 'Undefined model backbone'
    return model


def get_num_parameters(model):
    """reference model_utils.py:31-38"""
    m = model.module if hasattr(model, 'module') else model
    return sum(p.numel() for p in m.parameters() if p.requires_grad)


def make_data_parallel(model, configs):
    """reference model_utils.py:41-67.  One process per GPU; ``nn.DataParallel`` (the reference's last branch)
    is refused: it scatters target rows across devices and breaks the sample-index column (App. A #20)."""
    from ..parallel import RcclDataParallel
    if configs.distributed:
        if configs.gpu_idx is None:
            raise ValueError('distributed training uses one process per GPU: pass --gpu_idx / launch with torchrun')
        torch.cuda.set_device(configs.gpu_idx)
        model.cuda(configs.gpu_idx)
        configs.batch_size = int(configs.batch_size / configs.ngpus_per_node)
        configs.num_workers = int((configs.num_workers + configs.ngpus_per_node - 1) / configs.ngpus_per_node)
        return RcclDataParallel(model)
    if configs.gpu_idx is not None:
        torch.cuda.set_device(configs.gpu_idx)
        return model.cuda(configs.gpu_idx)
    raise ValueError('nn.DataParallel is not supported (it corrupts the targets tensor); pass --gpu_idx')
