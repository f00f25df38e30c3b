#!/usr/bin/env python3
"""Produce the persisted tile / split-K table (complex-yolov4-pytorch_amd/tune.py) on an MI355X.

    CY_TUNE_REPS=8 python tools/make_tune_cache.py gpurun_out/tune_gfx950.json [--quick]

Runs one training step (forward + backward: that is where the engine times its candidates) of every configuration bench.py
measures -- complex_yolov4.cfg 608x608 batch 16 in f16 and bf16, default and deterministic mode; 1024x1024 batch 8;
1216x1216 batch 16; the six other multiscale resolutions 512...704 at batch 16 -- and one eval forward of the batch-32
inference configuration, with the existing table ignored
(CY_TUNE_CACHE=0) so that everything is re-timed, and writes every choice with its time.  Copy the result to
complex-yolov4-pytorch_amd/tune_cache/gfx950.json and commit it: it is valid for exactly the kernel sources it was
measured on (sha inside)."""
import os
import sys
import time

os.environ['CY_TUNE_CACHE'] = '0'
os.environ['CY_TUNE_DET_TIMING'] = '1'      # this process may time deterministic engines too (models/engine.py)
os.environ.setdefault('CY_TUNE_REPS', '8')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd import tune  # noqa: E402
from complex_yolov4_pytorch_amd.models.darknet2pytorch import Darknet  # noqa: E402

CFG = os.path.join(ROOT, 'complex-yolov4-pytorch_amd', 'config', 'cfg', 'complex_yolov4.cfg')


def train_case(dtype, B, S, det, nt=6):
    torch.manual_seed(0)
    model = Darknet(CFG, use_giou_loss=True, dtype=dtype, deterministic=det).to('cuda').train()
    x, tg = syn.bev_images(B, S, seed=0).to('cuda'), syn.targets(B, nt, S, seed=0).to('cuda')
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        loss, _ = model(x, tg)
        loss.backward()
    torch.cuda.synchronize()
    model.release_engines()
    del model
    torch.cuda.empty_cache()


def eval_case(dtype, B, S):
    torch.manual_seed(0)
    model = Darknet(CFG, use_giou_loss=True, dtype=dtype).to('cuda').eval()
    model.cpu_outputs = False
    with torch.no_grad():
        model(syn.bev_images(B, S, seed=0).to('cuda'))
    torch.cuda.synchronize()
    model.release_engines()
    del model
    torch.cuda.empty_cache()


def main():
    out = sys.argv[1]
    quick = '--quick' in sys.argv
    cases = [('train', 'f16', 16, 608, False), ('train', 'f16', 16, 608, True)]
    if not quick:
        cases += [('train', 'bf16', 16, 608, False), ('train', 'bf16', 16, 608, True), ('eval', 'f16', 32, 608, False),
                  ('train', 'f16', 8, 1024, False), ('train', 'f16', 16, 1216, False), ('train', 'f32', 16, 608, True),
                  ('train', 'f32', 16, 608, False)]      # (f32 default mode: bench.py's other_configs.train608_f32)
        # multiscale training (reference kitti_dataset.py:42-43,225-230: img_size +- 3 x 32 every 10 batches): without these a
        # new resolution times its candidates at first sight
        cases += [('train', 'f16', 16, S, False) for S in (512, 544, 576, 640, 672, 704)]
    for kind, dtype, B, S, det in cases:
        t0 = time.time()
        if kind == 'train':
            train_case(dtype, B, S, det, nt=24 if S == 1216 else 6)
        else:
            eval_case(dtype, B, S)
        print('%s %s B%d %dx%d det=%s: %.1f s, %d entries' % (kind, dtype, B, S, S, det, time.time() - t0, len(tune._recorded)), flush=True)
    n = tune.save(out, merge=False)
    print('wrote', out, n, 'entries for kernel sources', tune.sources_sha())


if __name__ == '__main__':
    main()
