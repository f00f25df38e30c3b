"""CPU stand-in for bench.py's per-rank body, started by bench.self_launch in tests/test_bench_launch.py: the rank plumbing of
the benchmark (launch through torch.distributed.run, barrier, max-over-ranks timing, per-rank line, ONE JSON line from rank 0) on
CPU ranks over gloo, operator layer = tests/opsim.py, mini cfg.  Nothing here is a measurement; bench.py holds no simulator code."""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from bench import emit  # noqa: E402
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet  # noqa: E402
from complex_yolov4_pytorch_amd.parallel import RcclDataParallel  # noqa: E402


def sim_main(a, rank, world):
    from tests import opsim
    from tests.util import mini_cfg_path

    class _MP:
        def setattr(self, o, n, v):
            setattr(o, n, v)
    opsim.install(_MP())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29512')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    torch.set_num_threads(2)
    model = Darknet(mini_cfg_path(), use_giou_loss=True, dtype='f32')
    model.train()
    net = RcclDataParallel(model, bucket_bytes=64 << 10)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, tg = syn.bev_images(2, 64, seed=rank, sparsity=0.5), syn.targets(2, 3, 64, seed=rank)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = net(x, tg)
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    every = [torch.zeros_like(elapsed) for _ in range(world)]
    dist.all_gather(every, elapsed)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    w0 = next(model.parameters()).detach().double().sum().reshape(1)
    ws = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(ws, w0)
    dist.destroy_process_group()
    if rank == 0:
        emit({'metric': 'SIMULATED ranks (CPU, gloo, mini cfg): launch plumbing only', 'value': round(world * 2 * a.steps / float(elapsed), 3),
              'unit': 'images/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * float(elapsed) / a.steps, 3),
              'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
              'config': {'workload': 'simulated', 'global_batch': 2 * world, 'parallelism': 'dp%d' % world,
                         'loss_final': round(float(loss.detach().reshape(-1)[0]), 4)},
              'per_rank_ms_per_step': [round(1e3 * float(t) / a.steps, 3) for t in every],
              'params_equal_across_ranks': bool(all(float(w) == float(ws[0]) for w in ws))})


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    a = ap.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus != world:
        sys.exit('bench_sim.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (a.gpus, world))
    sim_main(a, rank, world)
