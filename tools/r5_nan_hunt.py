"""Round 5: which op of the f16 EVAL forward first produces a non-finite value when every engine buffer starts as 0xFF bytes
(NaN) between red zones (tests/test_gpu_redzone.py::test_eval_forward_between_red_zones failed with NaN outputs)?
Usage: python tools/r5_nan_hunt.py [B] [S] [dtype] [train]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import complex_yolov4_pytorch_amd.synthetic as syn  # noqa: E402
from complex_yolov4_pytorch_amd import ops  # noqa: E402
from complex_yolov4_pytorch_amd.models import engine as E  # noqa: E402
from tests.test_gpu_r2 import DEV, _model  # noqa: E402

B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 608
dtype = sys.argv[3] if len(sys.argv) > 3 else 'f16'
E.Engine.GUARD_BYTES = 64 << 10
model = _model('complex_yolov4.cfg', dtype)
model.eval()
model.cpu_outputs = False
model.static_eval_weights = True
os.environ['CY_PLAN_REPLAY'] = '0'
x = syn.bev_images(B, S, seed=33).to(DEV)
eng = model._engine_for(x)
eng.replay = False
found = []
for op in ('conv', 'pool', 'upsample', 'copy', 'add'):
    orig = getattr(E.Engine, '_f_' + op)

    def wrap(self, rec, *a, _orig=orig, _op=op):
        _orig(self, rec, *a)
        torch.cuda.synchronize()
        out = self.view(rec['out'])
        t = out.to_nchw()
        bad = int((~torch.isfinite(t)).sum())
        if bad and len(found) < 6:
            # which input of this op was already bad?
            ins = {}
            for k in ('x', 'res', 'a', 'b'):
                if rec.get(k) is not None:
                    v = self.view(rec[k]).to_nchw()
                    ins[k] = int((~torch.isfinite(v)).sum())
            nz = torch.nonzero(~torch.isfinite(t))
            found.append(1)
            print('op %s idx %s: %d non-finite of %d in its output %r (view C=%d ld=%d); inputs non-finite: %s; first at (n,c,h,w)=%s last=%s; rec: %s'
                  % (_op, rec.get('idx'), bad, t.numel(), rec['out'], out.C, out.ld, ins, nz[0].tolist(), nz[-1].tolist(),
                     {k: rec[k] for k in ('ks', 'stride', 'pad', 'cin', 'cout', 'cin_pad', 'bn', 'act', 'H', 'W') if k in rec}), flush=True)
    setattr(E.Engine, '_f_' + op, wrap)
with torch.no_grad():
    out = model(x)
print('output finite:', bool(torch.isfinite(out).all()), 'violations:', eng.arena.violations())
