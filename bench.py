#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): BEV images/s of a 608x608 Complex-YOLOv4 TRAIN step.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One step = forward + GIoU loss + backward + Adam update of complex_yolov4.cfg on a fixed synthetic batch of 16 BEV
images per GPU (BASELINE configs[1]); inputs are resident in HBM before the timed region.  Weak scaling: every rank
runs its own 16 images, gradients are averaged over RCCL (parallel.RcclDataParallel).  Rank 0 prints ONE JSON line with
`roofline` (the implicit-GEMM conv kernel, timed with HIP events on its launch stream) and `cpu_baseline` (the oracle --
a CPU restatement of the reference -- timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import complex_yolov4_pytorch_amd.ops as ops  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet  # noqa: E402
from complex_yolov4_pytorch_amd.parallel import RcclDataParallel  # noqa: E402
from complex_yolov4_pytorch_amd.utils.train_utils import create_optimizer  # noqa: E402

CFG = os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
MFMA_PEAK_TFLOPS = {'f16': 2500.0, 'f32': 157.3}     # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
HBM_PEAK_BPS = HBM_PEAK_GBS * 1e9


class _OptCfg:
    optimizer_type, lr, momentum, weight_decay = 'adam', 1e-3, 0.949, 5e-4     # reference train_config.py:82-94


def usable_cores(cap=32):
    """Cores this process may really use: affinity mask and cgroup CPU quota, capped (oversubscribed OpenMP teams on
    a 256-thread host made a single oracle step take minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
            if quota != 'max':
                n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def pmc_traffic(kernel, a):
    """HBM bytes per launch of the kernel family from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs of this same command; FETCH doubled per the gfx950 correction of MI355X_MICROARCH.md).  PMC counters
    cannot be read from inside the process, so this is the recorded figure for the default workload, or None."""
    if (a.batch, a.size, a.dtype) != (16, 608, 'f16'):
        return None
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')
    try:
        with open(path) as f:
            d = json.load(f)[kernel]
        return dict(bytes_per_launch=round(d['fetch_bytes_per_launch_corrected'] + d['write_bytes_per_launch']),
                    fetch=round(d['fetch_bytes_per_launch_corrected']), write=round(d['write_bytes_per_launch']),
                    unit='bytes', source='profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)')
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(batch, size, seconds_budget=20.0):
    """Oracle train step (forward + GIoU loss + backward) on the host cores, bounded sample."""
    from complex_yolov4_pytorch_amd.models.darknet_utils import parse_cfg
    from oracle import darknet_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    net = darknet_ref.DarknetRef(parse_cfg(CFG))
    ps, bs = net.param_shapes()
    params = {k: v.requires_grad_(True) for k, v in syn.fill_state_dict(ps).items()}
    bufs = syn.fill_state_dict(bs)
    x, tg = syn.bev_images(batch, size, seed=0), syn.targets(batch, 6, size, seed=0)

    def step():
        for p in params.values():
            p.grad = None
        _, loss, _ = net.forward(params, x, tg, True, True, bufs)
        loss.sum().backward()

    t0 = time.time()
    step()                                  # warm-up (also sizes the sample: a slow host gets one timed step)
    warm = time.time() - t0
    t0, n = time.time(), 0
    while n < 1 or (n < 8 and (time.time() - t0) + warm < seconds_budget):
        step()
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=round(batch / dt, 4), unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample='%d timed train steps (fwd+GIoU loss+bwd) of complex_yolov4.cfg, batch %d, %dx%d, fp32, torch-CPU oracle'
                       % (n, batch, size, size))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU')
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus > 1 and world != a.gpus:
        sys.exit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (a.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_ddp = os.environ.get('CY_DDP_FORCE') == '1'
    if world > 1 or force_ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)

    torch.manual_seed(0)
    model = Darknet(CFG, use_giou_loss=True, dtype=a.dtype).to(dev)
    model.train()
    net = RcclDataParallel(model) if (world > 1 or force_ddp) else model
    opt = create_optimizer(_OptCfg, model)          # FusedAdam (cy_adam_multi) on the device
    x = syn.bev_images(a.batch, a.size, seed=rank).to(dev)
    tg = syn.targets(a.batch, 6, a.size, seed=rank).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = net(x, tg)
        loss.backward()
        opt.step()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    sync()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed)
    final_loss = float(loss.detach().reshape(-1)[0])

    roofline = None
    if not a.no_roofline:
        # exclusive kernel durations: the probe steps issue the weight-gradient kernels on the main stream (in the timed
        # region they overlap the dgrad/BN kernels from a side stream, which stretches every kernel's own duration).
        # EVERY rank runs the probe steps (a step contains the gradient all-reduce); only rank 0 brackets its launches.
        sides = [(e, e.side) for e in model._engines.values()]
        for e, _ in sides:
            e.side = None
        if rank == 0:
            ops.PROFILER = ops.LaunchProfiler()
        for _ in range(2):
            step()
        summ = ops.PROFILER.summary() if rank == 0 else {}
        by_bound = None
        if rank == 0:
            # the same launches split by which roof bounds them: arithmetic intensity above / below the ridge point
            over = ops.PROFILER.bracket_overhead_ms()
            ridge = MFMA_PEAK_TFLOPS[a.dtype] * 1e12 / HBM_PEAK_BPS
            acc = {'mfma': [0.0, 0.0, 0.0, 0], 'hbm': [0.0, 0.0, 0.0, 0]}
            for kind, fl, nb, ev0, ev1 in ops.PROFILER.records:
                if kind != 'igemm' or nb <= 0:
                    continue
                k = 'mfma' if fl / nb >= ridge else 'hbm'
                t = max(ev0.elapsed_time(ev1) - over, 1e-6)
                acc[k][0] += fl; acc[k][1] += nb; acc[k][2] += t; acc[k][3] += 1
            by_bound = {}
            for k, (fl, nb, ms, n) in acc.items():
                if n:
                    by_bound[k] = dict(launches_per_step=n // 2, ms_per_step=round(ms / 2, 3),
                                       tflops=round(fl / (ms * 1e-3) / 1e12, 1), gbs_algorithmic=round(nb / (ms * 1e-3) / 1e9, 1),
                                       frac=round((fl / (ms * 1e-3)) / (MFMA_PEAK_TFLOPS[a.dtype] * 1e12), 4) if k == 'mfma'
                                       else round((nb / (ms * 1e-3)) / HBM_PEAK_BPS, 4))
        ops.PROFILER = None
        for e, sd in sides:
            e.side = sd
        sync()
    if roofline is None and rank == 0 and not a.no_roofline:
        ig = summ.get('igemm')
        if ig:
            ach = ig['flops'] / (ig['ms'] * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[a.dtype]
            roofline = dict(bound='mfma', kernel='igemm_kernel (implicit-GEMM conv: forward + dgrad launches, all tile variants)',
                            achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4),
                            traffic=(pmc_traffic('igemm', a) or {}).get('bytes_per_launch'),
                            traffic_detail=pmc_traffic('igemm', a),
                            launches_per_step=ig['launches'] // 2, avg_launch_us=round(1e3 * ig['ms'] / ig['launches'], 2),
                            hbm_gbs_algorithmic=round(ig['bytes'] / (ig['ms'] * 1e-3) / 1e9, 1),
                            algorithmic_bytes_per_launch=round(ig['bytes'] / ig['launches']),
                            measured='HIP events around every launch of the kernel on its launch stream, minus the duration of an empty event bracket; 2 extra single-stream steps after the timed region',
                            avg_launch_us_with_event_overhead=round(1e3 * ig['ms_raw'] / ig['launches'], 2))
            wg = summ.get('wgrad')
            if wg:
                roofline['wgrad_kernel'] = dict(achieved=round(wg['flops'] / (wg['ms'] * 1e-3) / 1e12, 2), unit='TFLOP/s',
                                                launches_per_step=wg['launches'] // 2,
                                                avg_launch_us=round(1e3 * wg['ms'] / wg['launches'], 2))
            roofline['conv_ms_per_step'] = round((ig['ms'] + (wg['ms'] if wg else 0)) / 2, 3)
            roofline['by_bound'] = by_bound   # launches above the ridge point against the MFMA peak, the rest against HBM
    if world > 1:
        dist.barrier()

    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(2, a.size)
        imgs = world * a.batch * a.steps
        line = {
            'metric': 'BEV images/s (608x608) train step', 'value': round(imgs / elapsed, 3), 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(1e3 * elapsed / a.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f16' if a.dtype == 'f16' else 'f32', 'data': 'synthetic',
            'config': {'workload': 'complex_yolov4.cfg train step (fwd + rotated-GIoU loss + bwd + Adam), batch %d per GPU, %dx%dx3 synthetic BEV, 6 targets/image'
                                   % (a.batch, a.size, a.size),
                       'global_batch': world * a.batch, 'parallelism': 'dp%d' % world, 'loss_final': round(final_loss, 4)},
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
