"""create_model / get_num_parameters / make_data_parallel -- drop-ins for reference src/models/model_utils.py
(:20-28, :31-38, :41-67): same names, same ``configs`` attributes read, same in-place edits of ``configs``.
``make_data_parallel`` hands the model to the RCCL flat-buffer wrapper of ``parallel.py`` instead of torch DDP
(SURVEY section 8a row K)."""
import torch

from .darknet2pytorch import Darknet


def create_model(configs):
    """Darknet built from ``configs.cfgfile``; ``configs.dtype`` ('f16' default / 'bf16' / 'f32') picks the compute mode,
    ``configs.deterministic`` (default False) the bit-reproducible reductions."""
    if configs.arch != 'darknet' or configs.cfgfile is None:
        raise AssertionError('Undefined model backbone')       # the reference asserts False here
    print('using darknet')
    return Darknet(cfgfile=configs.cfgfile, use_giou_loss=configs.use_giou_loss, dtype=getattr(configs, 'dtype', 'f16'),
                   deterministic=bool(getattr(configs, 'deterministic', False)))


def get_num_parameters(model):
    """Number of trainable scalars (of the wrapped module when the model is wrapped)."""
    net = getattr(model, 'module', model)
    return sum(w.numel() for w in net.parameters() if w.requires_grad)


def _per_process(value, parts):
    return int((value + parts - 1) / parts)


def make_data_parallel(model, configs):
    """One process per GPU.  Distributed: move the model to ``configs.gpu_idx``, split ``batch_size`` / ``num_workers``
    over the node's GPUs as the reference does and wrap the model; single GPU: just move it.  The reference's last
    branch (``nn.DataParallel`` over all visible GPUs) is refused: it scatters the rows of ``targets`` across devices and
    breaks their sample-index column."""
    from ..parallel import RcclDataParallel
    device = configs.gpu_idx
    if device is None:
        raise ValueError('nn.DataParallel is not supported (it corrupts the targets tensor); pass --gpu_idx '
                         '(distributed runs: one process per GPU, e.g. torchrun)')
    torch.cuda.set_device(device)
    model = model.cuda(device)
    if not configs.distributed:
        return model
    n = configs.ngpus_per_node
    configs.batch_size = int(configs.batch_size / n)
    configs.num_workers = _per_process(configs.num_workers, n)
    return RcclDataParallel(model)
