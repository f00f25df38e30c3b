#!/usr/bin/env python3
"""Refine the split-K factors of the persisted tune table (complex-yolov4-pytorch_amd/tune.py) INSIDE a real train step.

    python tools/refine_wgrad_splits.py gpurun_out/tune_refined.json [--configs f16:16:608,bf16:16:608,...]

models/engine.py picks a layer's split from back-to-back launches of that one kernel.  For some layers that disagrees with what
the launch costs inside the step (256->256 3x3 @38x38 in round 3: the table's split 7 ran 45 us in the step, split 14 ran 33 us):
in the step the kernel starts on cold slabs, after a different producer, and is followed by a different consumer.  This tool
starts from the current table and, for one configuration at a time, scales EVERY layer's split by a few factors (layers are
independent launches, so all can be varied at once), runs whole single-stream steps with HIP-event brackets around each
weight-gradient launch (ops.LaunchProfiler, the same brackets bench.py uses), adds the fold's share for the slabs
(bytes / measured fold rate), and keeps the best split per layer shape.  Tile choice (64 / 128) is left as tuned.

RESULT (round 3, one box, sources 2c867027822c79f1): 66 of 110 f16 layers changed; the exclusive weight-gradient time of a
single-stream step fell 4.02 -> 3.71 ms -- and the benchmarked two-stream step got SLOWER, 855.8 -> 849.9 images/s (three
alternations).  The refined splits are mostly deeper (more, shorter blocks): alone that fills the chip better, beside the trunk's
dgrad / BN kernels it takes CUs from the critical path.  The shipped table therefore stays the engine's own; this tool is kept
as the measurement."""
import os
import sys

os.environ['CY_WGRAD_SIDE_STREAM'] = '0'        # exclusive durations
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import complex_yolov4_pytorch_amd.ops as ops  # noqa: E402
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd import tune  # noqa: E402
from complex_yolov4_pytorch_amd.models import engine as eng_mod  # noqa: E402
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet  # noqa: E402

CFG = os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')
FACTORS = (0.5, 0.7, 1.0, 1.4, 2.0)
FOLD_BYTES_PER_US = 4.2e6       # wgrad_reduce_multi: 2.49 GB in 0.58 ms (profiles/r03_pmc_hbm_traffic.json, r03_per_step_kernels_single_stream.txt)
STEPS = 3


class Prof(ops.LaunchProfiler):
    cur = None

    def bracket(self, kind, flops, nbytes):
        return ops._Bracket(self, (kind, Prof.cur['idx'] if Prof.cur else -1), flops, nbytes)


_orig_work = eng_mod.Engine._conv_work


def _tagged(self, rec):
    Prof.cur = rec
    return _orig_work(self, rec)


eng_mod.Engine._conv_work = _tagged


def wgrad_key(eng, rec, heads):
    dy = eng.head_tmp[heads[id(rec)]] if id(rec) in heads else eng.view(rec['out'], grad=True)
    xv = eng.view(rec['x'])
    return ('wgrad', eng.dt, dy.N, dy.H, dy.W, dy.C, dy.ld, xv.H, xv.W, xv.C, xv.ld, rec['ks'], rec['stride'], rec['pad']), dy.M


def refine(dtype, B, S, nt=6):
    torch.manual_seed(0)
    model = Darknet(CFG, use_giou_loss=True, dtype=dtype).to('cuda').train()
    x, tg = syn.bev_images(B, S, seed=0).to('cuda'), syn.targets(B, nt, S, seed=0).to('cuda')

    def step():
        model.zero_grad(set_to_none=True)
        loss, _ = model(x, tg)
        loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    eng = model._engine_for(x)
    heads = {id(h['conv']): i for i, h in enumerate(eng.plan.heads)}
    base = dict(eng.wsplit)
    info = {}
    for rec in eng.plan.convs:
        idx = rec['idx']
        if idx in eng.watomic:
            continue
        key, M = wgrad_key(eng, rec, heads)
        cop, cip, kk = eng_mod._pad32(rec['cout']), rec['cin_pad'], rec['ks'] * rec['ks']
        info[idx] = dict(key=key, M=M, slab_us=cop * kk * cip * 4 / FOLD_BYTES_PER_US, cap=eng.wsplit_cap[idx], t64=idx in eng.wtile64)
    results = {}        # key -> {split: [cost_us, ...]}
    for f in FACTORS:
        for idx, inf in info.items():
            c = max(1, min(inf['cap'], int(round(base[idx] * f))))
            while c > 1 and (c - 1) * 512 >= inf['M']:
                c -= 1
            eng.wsplit[idx] = c
        eng._reduce_groups = None
        step()
        torch.cuda.synchronize()
        p = Prof()
        ops.PROFILER = p
        for _ in range(STEPS):
            step()
        ops.PROFILER = None
        torch.cuda.synchronize()
        over = p.bracket_overhead_ms()
        per = {}
        for (kind, idx), flops, nbytes, s, e in p.records:
            if kind == 'wgrad' and idx in info:
                per.setdefault(idx, []).append(max(s.elapsed_time(e) - over, 0.0) * 1e3)
        for idx, ts in per.items():
            inf = info[idx]
            c = eng.wsplit[idx]
            cost = sorted(ts)[len(ts) // 2] + c * inf['slab_us']
            results.setdefault(inf['key'], {}).setdefault(c, []).append(cost)
    changed = 0
    for idx, inf in info.items():
        r = results.get(inf['key'])
        if not r:
            continue
        best = min(r, key=lambda c: sum(r[c]) / len(r[c]))
        cost = sum(r[best]) / len(r[best])
        old = base[idx]
        old_cost = sum(r[old]) / len(r[old]) if old in r else None
        if old_cost is not None and cost > 0.97 * old_cost:
            best, cost = old, old_cost          # keep the tuned split unless the step says >= 3 % better
        if best != old:
            changed += 1
        tune.put(inf['key'], best + (1000 if inf['t64'] else 0), cost * 1e-3)
        eng.wsplit[idx] = best
    print('%s B%d %dx%d: %d of %d layers changed their split' % (dtype, B, S, S, changed, len(info)), flush=True)
    del model, eng
    torch.cuda.empty_cache()


def main():
    out = sys.argv[1]
    cfgs = 'f16:16:608,bf16:16:608,f16:8:1024,f16:16:1216'
    for a in sys.argv[2:]:
        if a.startswith('--configs='):
            cfgs = a.split('=', 1)[1]
    if not tune.valid():
        sys.exit('the persisted table does not match the kernel sources: run tools/make_tune_cache.py first')
    import shutil
    shutil.copyfile(os.environ.get('CY_TUNE_CACHE_PATH', tune.CACHE_PATH), out)     # tune.save merges into it: every other entry stays
    for c in cfgs.split(','):
        dt, B, S = c.split(':')
        refine(dt, int(B), int(S), nt=24 if int(S) == 1216 else 6)
    tune.save(out)
    print('wrote', out)


if __name__ == '__main__':
    main()
