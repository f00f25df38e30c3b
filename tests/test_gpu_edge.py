"""Edge cases and size-independent properties at the BASELINE sizes (GPU): empty / ragged targets, the reference's NaN
on an empty batch, maximum NMS sizes (idempotence, sortedness), batch-permutation invariance of the full-size train step,
mosaic (1216^2) and 1024^2 geometries, degenerate boxes."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import complex_yolov4_pytorch_amd.ops as ops  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet  # noqa: E402
from complex_yolov4_pytorch_amd.utils.evaluation_utils import nms_cpu, post_processing_v2  # noqa: E402
from complex_yolov4_pytorch_amd.utils.iou_rotated_boxes_utils import iou_pred_vs_target_boxes  # noqa: E402
from oracle import nms_ref, yolo_layer_ref  # noqa: E402
from tests.golden.make_golden import V4_ANCH, head_input  # noqa: E402
from tests.test_gpu_head import _anchors, _run_head  # noqa: E402

CFG = os.path.join(os.path.dirname(__file__), '..', 'complex-yolov4-pytorch_amd', 'config', 'cfg')
DEV = 'cuda'


def test_empty_target_batch_is_nan_like_the_reference():
    """nT == 0 -> masked means over empty sets -> NaN total loss (the reference never guards it)."""
    x = head_input(2, 19, 0)
    tg = torch.zeros(0, 8)
    _, l_ref, m_ref = yolo_layer_ref.head_forward(x, tg, _anchors((6, 7, 8)), 3, 0.7, 608, True)
    assert math.isnan(float(l_ref.reshape(-1)[0]))
    out, met, dl = _run_head(x, tg, _anchors((6, 7, 8)), True)
    assert math.isnan(met[0]) and met[18] == 0
    # the no-object objectness term is still well defined and matches
    np.testing.assert_allclose(met[17], m_ref['conf_noobj'], rtol=1e-5)
    assert torch.isfinite(dl).all()


def test_ragged_targets_one_image_without_objects():
    x = head_input(3, 19, 2)
    tg = syn.targets(3, 4, 608, seed=3)
    tg = tg[tg[:, 0] != 1]                       # image 1 has no target rows
    xr = x.clone().requires_grad_(True)
    _, l_ref, m_ref = yolo_layer_ref.head_forward(xr, tg, _anchors((6, 7, 8)), 3, 0.7, 608, True)
    l_ref.sum().backward()
    _, met, dl = _run_head(x, tg, _anchors((6, 7, 8)), True)
    np.testing.assert_allclose(met[0], float(l_ref.detach()), rtol=1e-4)
    np.testing.assert_allclose(dl.numpy(), xr.grad.numpy(), rtol=1e-3, atol=5e-6)


def test_identical_and_degenerate_boxes():
    # identical boxes: the reference's float32 clip of a polygon against itself is decided by the last bits of the corner
    # coordinates.  Where they are exact (axis-aligned) it returns IoU 1.0 / loss 0.0 and so must we; at
    # yaw 0.3 the reference ITSELF returns IoU 0.8355 / loss 0.0749 on this host (vertices on an edge evaluate to +-1 ulp and
    # spurious crossing points appear) -- a value that moves with sinf / cosf / atan2f ulps, so only its range is asserted.
    for yaw in (0.0, math.pi / 2):
        b = torch.tensor([[10., 10., 4., 2., math.sin(yaw), math.cos(yaw)]], device=DEV)
        ious, loss = iou_pred_vs_target_boxes(b, b.clone(), GIoU=True)
        assert abs(float(ious[0]) - 1.0) < 1e-5 and abs(float(loss)) < 1e-5
    b = torch.tensor([[10., 10., 4., 2., math.sin(0.3), math.cos(0.3)]], device=DEV)
    ious, loss = iou_pred_vs_target_boxes(b, b.clone(), GIoU=True)
    assert 0.8 < float(ious[0]) <= 1.0 + 1e-6 and 0.0 <= float(loss) < 0.2
    i64, _ = iou_pred_vs_target_boxes(b, b.clone(), GIoU=False)                # the float64 convex clip is exact here
    assert abs(float(i64[0]) - 1.0) < 1e-6
    far = b.clone(); far[0, 0] += 100
    i64, l64 = iou_pred_vs_target_boxes(b, far, GIoU=False)
    assert float(i64[0]) == 0.0 and abs(float(l64) - 1.0) < 1e-6
    # zero-area candidate in NMS: self-IoU is 0 in the reference (infinite loop there); here it is emitted once
    pred = syn.nms_predictions(1, 64, 8, seed=2)
    rows = torch.nonzero(pred[0, :, 6] >= 0.5).reshape(-1)
    pred[0, rows[0], 2] = 0.0
    out = post_processing_v2(pred, 0.5, 0.5)[0]
    assert out is not None and torch.isfinite(out[:, 6:]).all()


def test_greedy_nms_idempotent_and_sorted_at_max_size():
    """All 22743 rows of one image as candidates: kept set is sorted by confidence and NMS of the kept set keeps all."""
    pred = syn.nms_predictions(1, 22743, 22743, seed=11, n_centres=400)[0]
    boxes, confs = pred[:, :6].to(DEV), (pred[:, 6] * pred[:, 7:].max(1)[0]).to(DEV)
    keep = nms_cpu(boxes, confs, 0.5)
    assert 0 < len(keep) < 22743 and len(set(keep.tolist())) == len(keep)
    kc = confs.cpu().numpy()[keep]
    assert np.all(np.diff(kc) <= 0)
    keep2 = nms_cpu(boxes[torch.from_numpy(keep).to(DEV)], confs[torch.from_numpy(keep).to(DEV)], 0.5)
    np.testing.assert_array_equal(keep2, np.arange(len(keep)))
    # spot-check the pairwise decision on the kept set against the float64 oracle
    sub = keep[:300]
    iou = nms_ref.iou_matrix(pred[sub, :6].numpy(), pred[sub, :6].numpy())
    assert np.all(iou[np.triu_indices(len(sub), 1)] <= 0.5)


def test_post_processing_v2_max_size_properties():
    pred = syn.nms_predictions(2, 22743, 6000, seed=12, n_centres=300)
    outs = post_processing_v2(pred, 0.5, 0.5)
    for o in outs:
        assert o is not None and o.shape[1] == 9 and torch.isfinite(o).all()
        score = (o[:, 6] * o[:, 7]).numpy()
        assert np.all(np.diff(score) <= 1e-7)                 # emitted in rank order
        assert set(np.unique(o[:, 8].numpy())) <= {0.0, 1.0, 2.0}


def _model(cfg, dtype, giou=True):
    torch.manual_seed(0)
    return Darknet(os.path.join(CFG, cfg), use_giou_loss=giou, dtype=dtype).to(DEV)


def _perm_step(cfg, dtype, B, S):
    model = _model(cfg, dtype)
    model.train()
    x, tg = syn.bev_images(B, S, seed=21), syn.targets(B, 6, S, seed=21)
    loss, _ = model(x.to(DEV), tg.to(DEV))
    loss.backward()
    g0 = model.flat_grad.clone()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    inv = torch.empty_like(perm); inv[perm] = torch.arange(B)
    tg2 = tg.clone(); tg2[:, 0] = inv[tg[:, 0].long()].float()
    for p in model.parameters():
        p.grad = None
    loss2, _ = model(x[perm].to(DEV), tg2.to(DEV))
    loss2.backward()
    assert torch.isfinite(loss).all() and torch.isfinite(g0).all() and torch.isfinite(model.flat_grad).all()
    rel = float((model.flat_grad - g0).norm() / g0.norm())
    # kink-free statistics of the same two gradients: the ratio of their norms, and the element-wise relative
    # difference at the MEDIAN element -- a pre-activation that changes side of the leaky-ReLU kink moves the gradient along
    # ONE path (a few elements by a lot: the norm of the difference), not the typical element
    ratio = float(model.flat_grad.norm() / g0.norm())
    d = (model.flat_grad - g0).abs() / (g0.abs() + 1e-3 * float(g0.abs().mean()))
    med = float(d.median())
    print('%s %s permutation: loss %.4f vs %.4f, grad rel diff %.2e, norm ratio %.6f, median element-wise rel diff %.2e'
          % (cfg, dtype, float(loss.detach()), float(loss2.detach()), rel, ratio, med))
    return float(loss.detach()), float(loss2.detach()), rel, ratio, med


def test_train_step_is_batch_permutation_invariant_fp32():
    """Permuting the images (and the sample index of the targets) leaves loss and gradients unchanged: BN statistics,
    target assignment and the reductions do not depend on sample order (tiny cfg, fp32: tight)."""
    l1, l2, rel, ratio, med = _perm_step('complex_yolov4_tiny.cfg', 'f32', 4, 608)
    # The loss is tight.  The bound on the NORM OF THE DIFFERENCE is not a kernel tolerance: a permutation changes the summation
    # order of the BatchNorm statistics, a pre-activation next to the leaky-ReLU kink may change side, and the gradient then moves
    # discontinuously along that path -- measured over ten runs on the MI355X (round 5): 1.1e-3 ... 2.1e-3 nine times, 8.2e-3 once.
    # What a sample-order dependence of the statistics, the target assignment or a reduction WOULD move -- also at the 1e-2 level,
    # which that bound alone would let pass -- are the kink-free statistics: the gradient's norm and its typical element.
    assert abs(l1 - l2) / l1 < 1e-5 and rel < 3e-2
    assert abs(ratio - 1.0) < 2e-3 and med < 1e-3, (ratio, med)


def test_full_size_train_step_permutation_invariant_loss():
    """BASELINE configs[1] size (complex_yolov4.cfg, batch 16, 608^2, fp16).  The LOSS is invariant to 1e-2; the
    gradient of this random-init 110-layer net is not a stable quantity at fp16 resolution -- two REPEATS of the
    identical step already differ by O(1) in direction (tools/perm_probe.py: fp32 4e-2, fp16 0.9; the fp32 atomics of
    the BN statistic bins reorder a 1e-7 perturbation that the net amplifies; the reference shows the same
    sensitivity, tests/test_plan_sim.py), so only finiteness is asserted for it here."""
    l1, l2, rel, _, _ = _perm_step('complex_yolov4.cfg', 'f16', 16, 608)
    assert abs(l1 - l2) / l1 < 1e-2


@pytest.mark.parametrize('size,batch', [(1216, 1), (1024, 2)])
def test_large_geometries_tiny_cfg_fp32_vs_oracle(size, batch):
    """Mosaic inputs are 1216x1216 in the reference (App. A #19); BASELINE configs[4] is 1024^2."""
    from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
    from oracle import darknet_ref
    model = _model('complex_yolov4_tiny.cfg', 'f32')
    sd = model.state_dict()
    sd.update({k: syn.fill_tensor(k, tuple(v.shape)).to(DEV) for k, v in sd.items() if v.dtype.is_floating_point})
    model.load_state_dict(sd)
    model.train()
    x, tg = syn.bev_images(batch, size, seed=31), syn.targets(batch, 6, size, seed=31)
    loss, out = model(x.to(DEV), tg.to(DEV))
    loss.backward()
    net = darknet_ref.DarknetRef(parse_cfg(os.path.join(CFG, 'complex_yolov4_tiny.cfg')))
    ps, bs = net.param_shapes()
    params = {k: v.requires_grad_(True) for k, v in syn.fill_state_dict(ps).items()}
    o_ref, l_ref, _ = net.forward(params, x, tg, True, True, syn.fill_state_dict(bs))
    assert tuple(out.shape) == tuple(o_ref.shape) == (batch, 3 * ((size // 32) ** 2 + (size // 16) ** 2), 10)
    np.testing.assert_allclose(float(loss.detach()), float(l_ref.detach()), rtol=1e-4)
    np.testing.assert_allclose(out.cpu().numpy(), o_ref.detach().numpy(), rtol=2e-3, atol=2e-3)


def test_v4_fp16_1024_step_is_finite():
    model = _model('complex_yolov4.cfg', 'f16')
    model.train()
    loss, out = model(syn.bev_images(2, 1024, seed=41).to(DEV), syn.targets(2, 6, 1024, seed=41).to(DEV))
    loss.backward()
    assert tuple(out.shape) == (2, 64512, 10)                 # 64,512 boxes per image at 1024^2
    assert torch.isfinite(loss).all() and torch.isfinite(model.flat_grad).all()


def test_bev_rasteriser_golden_and_edges(golden):
    """cy_bev_rasterize through the kitti_bev_utils drop-ins against the reference's maps: occupied pixels, intensity and
    height bit-exact (including the equal-height tie rule and the box faces), density to 1 ulp of float32."""
    import numpy as np
    import complex_yolov4_pytorch_amd.config.kitti_config as cnf
    from complex_yolov4_pytorch_amd.data_process import kitti_bev_utils as bev
    from oracle import bev_ref
    from tests.golden.make_golden import lidar_points
    g = golden('bev')
    pts = lidar_points(30000)
    ref = np.zeros((3, 608 * 608), dtype=np.float32)
    ref[:, g['pixels']] = g['values']
    ref = ref.reshape(3, 608, 608)
    # raw points on the device, one fused pass
    got = bev.makeBVFeature(torch.from_numpy(pts).cuda(), cnf.DISCRETIZATION, cnf.boundary, raw=True)
    assert got.is_cuda and got.dtype == torch.float32 and got.shape == (3, 608, 608)
    got = got.cpu().numpy()
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_allclose(got[2], ref[2], rtol=2e-7, atol=0)
    np.testing.assert_array_equal(got[2] != 0, ref[2] != 0)
    # the reference's two-call form on numpy arrays: numpy float64 out
    b = bev.removePoints(pts.copy(), cnf.boundary)
    assert b.shape[0] == int(g['kept'][0])
    got2 = bev.makeBVFeature(b, cnf.DISCRETIZATION, cnf.boundary)
    assert isinstance(got2, np.ndarray) and got2.dtype == np.float64
    np.testing.assert_array_equal(got2.astype(np.float32)[:2], ref[:2])
    # the filtered points survive any container change (np.copy / torch tensor / device tensor): no hidden tag
    got3 = bev.makeBVFeature(torch.from_numpy(np.array(b, dtype=np.float32)).cuda().float(), cnf.DISCRETIZATION, cnf.boundary)
    np.testing.assert_array_equal(got3.cpu().numpy()[:2], ref[:2])
    # the workspace is left clean: a second frame is not polluted by the first; empty cloud -> zeros
    again = bev.makeBVFeature(torch.from_numpy(pts).cuda(), cnf.DISCRETIZATION, cnf.boundary, raw=True).cpu().numpy()
    np.testing.assert_array_equal(again, got)
    empty = bev.makeBVFeature(torch.zeros(0, 4).cuda(), cnf.DISCRETIZATION, cnf.boundary)
    assert float(empty.abs().sum()) == 0.0
    # a full-size scan (120k points, KITTI-like) against the oracle
    big = lidar_points(120000, seed=5)
    o = bev_ref.make_bv_feature(bev_ref.remove_points(big.copy(), cnf.boundary), cnf.DISCRETIZATION, cnf.boundary).astype(np.float32)
    gb = bev.makeBVFeature(torch.from_numpy(big).cuda(), cnf.DISCRETIZATION, cnf.boundary, raw=True).cpu().numpy()
    np.testing.assert_array_equal(gb[:2], o[:2])
    np.testing.assert_allclose(gb[2], o[2], rtol=2e-7, atol=0)
