"""C-ABI surface (CPU): the built library loads and exports every symbol include/cyolo_hip.h declares; host-only
entry points work without a device; the product layer refuses to run without one."""
import ctypes
import os

import pytest
import torch

from complex_yolov4_pytorch_amd import _lib, ops

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__
    return __graft_entry__.build()


def test_header_symbols_are_exported(built):
    protos = _lib.parse_header()
    assert len(protos) >= 46
    dll = ctypes.CDLL(built)
    missing = [n for n in protos if not hasattr(dll, n)]
    assert not missing, missing
    for required in ('cy_conv_igemm', 'cy_conv_wgrad', 'cy_bn_act_fwd', 'cy_yolo_loss', 'cy_riou_pairs',
                     'cy_rnms_greedy', 'cy_pp2_merge'):
        assert required in protos


def test_host_only_entry_points(built):
    lib = _lib.lib()
    assert lib.raw('cy_version')() >= 100
    assert lib.raw('cy_conv_stats_rows')(16 * 76 * 76, 128) > 0
    assert lib.raw('cy_conv_wgrad_split')(16 * 76 * 76, 128, 128, 3) >= 1
    assert lib.raw('cy_yolo_loss_workspace')(16, 76, 3, 3, 96) > 16 * 3 * 76 * 76 * 8
    assert lib.raw('cy_rnms_workspace')(32, 256) > 0
    assert lib.raw('cy_bn_scratch_rows')() >= 0 and lib.raw('cy_bn_bwd_rows')(1000, 64, 0) == lib.raw('cy_conv_stats_rows')(1000, 64) == 16


def test_argument_validation_without_launch(built):
    lib = _lib.lib()
    # NULL pointers / bad dtype are rejected before any launch: CY_ERR_ARG = -1
    assert lib.raw('cy_conv_igemm')(None, 1, 8, 8, 8, 8, None, 8, None, 8, 8, 8, 8, 3, 1, 1, 0, 0, None, None, None, None) == -1
    assert lib.raw('cy_riou_pairs')(None, None, 4, 1, None, None, None, None) == -1
    assert lib.raw('cy_bn_act_fwd')(None, 8, None, 8, None, 0, 10, 8, None, None, 0, 0, None) == -1
    # the entry points added around the path (rasteriser, pools, optimizers, loss scaling)
    assert lib.raw('cy_bev_rasterize')(None, 10, 0., 50., -25., 25., -2.73, 1.27, -2.73, 4.0, 0.08, 608, 608, None, None, None) == -1
    assert lib.raw('cy_maxpool_fwd')(None, 1, 8, 8, 8, 8, None, 8, 8, 8, 5, 1, 2, None, None, 0, None) == -1
    assert lib.raw('cy_maxpool_argmax_bytes')(2, 19, 19, 19, 512) == 2 * (19 + 19) * 19 * 512
    assert lib.raw('cy_grad_nonfinite')(None, 10, None, None) == -1
    assert lib.raw('cy_sgd_multi')(None, None, 0, 0.9, 1, 1, 0, None, None, 0, None, None) == -1
    assert lib.raw('cy_adam_multi')(None, None, 0, 0.9, 0.999, 1e-8, 0.1, 0.001, 0, None, None, 0, None, None) == -1
    assert lib.raw('cy_bev_workspace')(608, 608) == 608 * 608 * 12
    assert lib.raw('cy_pipe_launches')() >= 0
    assert lib.raw('cy_bev_mosaic')(None, None, None, None, 3, 8, 8, 8, None, 0.5, None, None) == -1
    assert lib.raw('cy_bev_mosaic_targets')(None, 0, None, 8, 8, None, 8, None) == -1
    assert lib.raw('cy_bev_flip_cutout')(None, 3, 8, 8, 1, None, 0, 0.0, None, None, 0, None, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-device behaviour')
def test_no_cpu_fallback():
    with pytest.raises(_lib.CyoloError):
        ops.riou_pairs(torch.zeros(2, 6), torch.zeros(2, 6), True)
    with pytest.raises(_lib.CyoloError):
        ops.nchw_to_nhwc(torch.zeros(1, 3, 8, 8), 8, ops.CY_F16)
