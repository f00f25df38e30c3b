"""Which kernel of cy_yolo_loss changes its result when other kernels run beside it?  Like head_race_probe.py, but after every
repeat the whole workspace (assignment table ti, pair results tf, ownership maps, accumulators) is compared with the first
run's, component by component.   usage: python tools/head_race_probe2.py [iters=600] [giou=1]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from tests import probes
import complex_yolov4_pytorch_amd.synthetic as syn
from complex_yolov4_pytorch_amd.ops import CY_F16, View

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
giou = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
load = sys.argv[3] if len(sys.argv) > 3 else 'conv'      # what runs beside the loss: our conv kernel | torch GEMM | torch elementwise
B, G, A, C, S = 16, 76, 3, 3, 608
anchors = [(11, 14, 0, 1), (11, 14, -3.14, 1), (11, 14, 0.5, 0.8)]
torch.manual_seed(0)
logits = (torch.randn(B * G * G * A * (7 + C), device='cuda') * 0.5).contiguous()
tg = syn.targets(B, 6, S, seed=5).cuda()
nT = tg.shape[0]
need = ops.yolo_loss_workspace(B, G, A, C, nT)
ws = torch.empty(need, dtype=torch.uint8, device='cuda')
met = torch.zeros(20, device='cuda')
dl = torch.empty_like(logits)
side = torch.cuda.Stream()
x = View.alloc(16, 152, 152, 128, CY_F16); x.buf.normal_()
y = View.alloc(16, 152, 152, 128, CY_F16)
w = torch.randn(128, 128, 3, 3, device='cuda') * 0.03
wf, _ = ops.pack_weights(w, 128, 128, CY_F16)
cells = B * A * G * G
al = lambda v: (v + 255) // 256 * 256      # noqa: E731
o_acc, o_cnt = 0, 256
o_owner = 512
o_flags = o_owner + al(4 * cells)
o_ti = o_flags + al(4 * cells)
o_tf = o_ti + al(16 * nT)
parts = dict(acc=(o_acc, 144), cnt=(o_cnt, 12), owner=(o_owner, 4 * cells), flags=(o_flags, 4 * cells), ti=(o_ti, 16 * nT), tf=(o_tf, 32 * nT))


ma = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
mb = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
big = torch.randn(64 << 20, device='cuda')


xf = View.alloc(16, 152, 152, 128, 2); xf.buf.normal_()
yf = View.alloc(16, 152, 152, 128, 2)
wff, _ = ops.pack_weights(w, 128, 128, 2)
sc = torch.ones(128, device='cuda'); sh = torch.zeros(128, device='cuda')
part = torch.empty(8 * 128 * 9 * 128, device='cuda')


def busy_load():
    if load.startswith('conv'):      # conv = library default; convN = tile hint N (1: 4-wave, 2-6: pipelined, 7-9: + loader waves)
        hint = int(load[4:] or 0)
        for _ in range(3):
            ops.conv_igemm(x, wf, 128, y, 3, 1, 1, tile=hint)
    elif load == 'f32conv':          # the general kernel (per-lane tap), fp32
        ops.conv_igemm(xf, wff, 128, yf, 3, 1, 1)
    elif load == 'wgrad':
        for _ in range(2):
            ops.conv_wgrad(y, x, 3, 1, 1, part, 8)
    elif load.startswith('dirty'):   # dirtyNaN / dirtyBig / dirtyZero: leave a pattern in every VGPR and LDS word of every CU
        pat = {'dirtyNaN': 0x7FC00001, 'dirtyBig': 0x4B800000, 'dirtyZero': 0, 'dirtyNeg': 0xBF800000, 'dirtyInt': 0x00000005}[load]
        probes.probe_dirty(pat, blocks=int(os.environ.get('CY_PROBE_DIRTY_BLOCKS', '2048')), lds_bytes=int(os.environ.get('CY_PROBE_DIRTY_LDS', str(64 * 1024))))
    elif load == 'bn':
        for _ in range(6):
            ops.bn_act_fwd(x, y, None, sc, sh, 2)
    elif load == 'mm':
        for _ in range(2):
            torch.mm(ma, mb)
    else:
        for _ in range(3):
            big.mul_(1.0000001)


def loss_on_side():
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    side.wait_event(ev)
    with torch.cuda.stream(side), ops.stream_scope(side):
        ops.yolo_loss(logits, B, G, A, C, tg, anchors, S, 0.7, giou, ws, met, dl)


for busy in (False, True):
    ref, bad = None, 0
    for it in range(iters):
        if busy:
            busy_load()
        loss_on_side()
        if busy:
            busy_load()
        torch.cuda.synchronize()
        cur = (met.clone(), dl.clone(), ws.clone())
        if ref is None:
            ref = cur
            print('   reference of this mode: loss %.9g, sum|dlogits| %.9g, tf checksum %.9g' % (float(cur[0][0]), float(cur[1].double().abs().sum()),
                  float(cur[2][o_tf:o_tf + 32 * nT].view(torch.float32).double().abs().sum())), flush=True)
            continue
        if torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]) and torch.equal(cur[2][:o_tf + 32 * nT], ref[2][:o_tf + 32 * nT]):
            continue
        bad += 1
        if bad <= 4:
            msg = []
            for name, (off, n) in parts.items():
                a, b = cur[2][off:off + n], ref[2][off:off + n]
                if not torch.equal(a, b):
                    if name in ('ti',):
                        ai, bi = a.view(torch.int32).view(-1, 4).cpu(), b.view(torch.int32).view(-1, 4).cpu()
                        rows = (ai != bi).any(1).nonzero().flatten().tolist()
                        msg.append('ti rows %s now %s ref %s' % (rows[:6], ai[rows[:3]].tolist(), bi[rows[:3]].tolist()))
                    elif name == 'tf':
                        af, bf = a.view(torch.float32).view(-1, 8).cpu(), b.view(torch.float32).view(-1, 8).cpu()
                        rows = (af != bf).any(1).nonzero().flatten().tolist()
                        msg.append('tf rows %s now %s ref %s' % (rows[:8], [round(v, 4) for v in af[rows[0]].tolist()], [round(v, 4) for v in bf[rows[0]].tolist()]))
                    elif name == 'acc':
                        ad, bd = a.view(torch.float64).cpu(), b.view(torch.float64).cpu()
                        msg.append('acc %s' % [(i, float(ad[i]), float(bd[i])) for i in range(18) if ad[i] != bd[i]])
                    else:
                        msg.append('%s: %d words differ' % (name, int((a.view(torch.int32) != b.view(torch.int32)).sum())))
            print('   repeat %d: metrics equal %s, dlogits equal %s; %s' % (it, torch.equal(cur[0], ref[0]), torch.equal(cur[1], ref[1]), ' | '.join(msg) or 'workspace identical'), flush=True)
    print('giou=%s, %s kernels in flight on the main stream: %s -> %d of %d repeats differ' % (giou, load, busy, bad, iters - 1), flush=True)
