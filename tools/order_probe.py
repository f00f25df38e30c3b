"""Why is train1216 15 % slower as the last entry of the default bench run than alone (VERDICT r2 weak #6b)?  Runs, in ONE
process: train608, train1216, train608 again, train1216 again (fresh engines each time), and prints images/s plus the
shader clock rocm-smi reports right after each leg.  A clock / power effect shows up on BOTH configurations' second legs;
an allocator or tuning-state effect only where the state differs.   usage: python tools/order_probe.py"""
import os
import subprocess
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'Power', 'Temperature (Sensor junction)', 'Temperature (Sensor memory)'))]
        return ' | '.join(keep)[:400]
    except Exception as e:      # noqa: BLE001
        return 'rocm-smi: %r' % (e,)


dev = torch.device('cuda', 0)
print('idle:', smi(), flush=True)
for leg, (cfg, steps) in enumerate([('train608', 40), ('train1216', 10), ('train608', 40), ('train1216', 10), ('train1216', 40)]):
    c = bench.CONFIGS[cfg]
    t0 = time.time()
    r = bench.measure_train(dev, c['batch'], c['size'], 'f16', steps, 3, mosaic=bool(c.get('mosaic')))
    print('leg %d %-9s %4d steps: %8.1f images/s  %7.2f ms/step  (leg wall %.1f s, allocated %.1f GB, reserved %.1f GB)  %s'
          % (leg, cfg, steps, r['value'], r['ms_per_step'], time.time() - t0, torch.cuda.memory_allocated() / 1e9,
             torch.cuda.memory_reserved() / 1e9, smi()), flush=True)
