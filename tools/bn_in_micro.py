"""consumer-side BatchNorm, MEASURED.  The producer's BatchNorm + Mish pass followed by the 1x1 conv that reads its
output (what the engine runs: cy_bn_act_fwd_fused + cy_conv_igemm on the streaming kernel) against ONE launch of the same conv
kernel that reads the producer's pre-BN tensor, transforms the rows on their way into LDS and writes the activated tensor as a side
output (cy_conv1x1_bn_in + the 3 us cy_bn_finalize it then needs), interleaved rounds, batch 16 of complex_yolov4.cfg's 304 x 304
stage: 64 -> 128 (the sibling pair as one conv) and 64 -> 64.  usage: python tools/bn_in_micro.py [rounds] [reps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import complex_yolov4_pytorch_amd.ops as ops
from complex_yolov4_pytorch_amd.ops import CY_F16, View

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dt = CY_F16
for (N, H, Ci, Co) in [(16, 304, 64, 128), (16, 304, 64, 64), (32, 304, 64, 128)]:
    raw = View.alloc(N, H, H, Ci, dt); raw.buf.normal_()
    act = View.alloc(N, H, H, Ci, dt)
    act2 = View.alloc(N, H, H, Ci, dt)
    out = View.alloc(N, H, H, Co, dt)
    out2 = View.alloc(N, H, H, Co, dt)
    w = torch.randn(Co, Ci, 1, 1, device='cuda') * 0.1
    wf, _ = ops.pack_weights(w, Co, Ci, dt)
    M = raw.M
    rows = ops.conv_stats_rows(M, Ci)
    bins = torch.zeros(rows * 2 * Ci, device='cuda')
    # the producer's statistics bins as its conv epilogue would have left them (sum, sum of squares per channel, spread over the bins)
    x32 = raw.buf.float().view(-1, Ci)
    bins.view(rows, 2, Ci)[0, 0] = x32.sum(0)
    bins.view(rows, 2, Ci)[0, 1] = (x32 * x32).sum(0)
    gamma = torch.rand(Ci, device='cuda') + 0.5
    beta = torch.randn(Ci, device='cuda') * 0.1
    rmean, rvar = torch.zeros(Ci, device='cuda'), torch.ones(Ci, device='cuda')
    nbt = torch.zeros(1, dtype=torch.int64, device='cuda')
    vec = torch.zeros(4, Ci, device='cuda')
    other = torch.zeros(rows * 2 * 128, device='cuda')
    st1 = torch.zeros((ops.conv_stats_rows(M, Co) + ops.bn_scratch_rows()) * 2 * Co, device='cuda')
    st2 = torch.zeros_like(st1)
    MISH = ops.ACT['mish']

    bins0 = bins.clone()

    def two_launches():
        bins.copy_(bins0)      # (cy_bn_finalize leaves its table zeroed: both candidates restore it, the same 8 KB copy in each)
        ops.bn_act_fwd_fused(raw, act, None, bins, rows, gamma, beta, rmean, rvar, nbt, 0.03, 1e-5, vec, other, MISH)
        ops.conv_igemm(act, wf, Co, out, 1, 1, 0, flags=ops.CONV_STATS, stats=st1)

    def consumer_side():
        bins.copy_(bins0)
        ops.bn_finalize(bins, rows, Ci, int(M), gamma, beta, rmean, rvar, nbt, 0.03, 1e-5, vec[0], vec[1], vec[2], vec[3])
        ops.conv1x1_bn_in(raw, vec[2], vec[3], MISH, act2, wf, Co, out2, flags=ops.CONV_STATS, stats=st2)

    def conv_only():
        ops.conv_igemm(act, wf, Co, out, 1, 1, 0, flags=ops.CONV_STATS, stats=st1)

    def bn_only():
        bins.copy_(bins0)
        ops.bn_act_fwd_fused(raw, act, None, bins, rows, gamma, beta, rmean, rvar, nbt, 0.03, 1e-5, vec, other, MISH)

    cands = {'bn pass + conv': two_launches, 'consumer-side': consumer_side, '(conv alone)': conv_only, '(bn pass alone)': bn_only}
    for f in cands.values():
        f()
    torch.cuda.synchronize()
    st1.zero_(); st2.zero_()
    two_launches(); consumer_side()
    torch.cuda.synchronize()
    d_act = (act.buf.float() - act2.buf.float()).abs().max().item()
    d_out = (out.buf.float() - out2.buf.float()).abs().max().item()
    # (bins are picked by block index and the two conv launches may tile differently: compare the sums over the bins)
    r1 = ops.conv_stats_rows(M, Co)
    b1, b2 = st1[:r1 * 2 * Co].view(r1, 2, Co).sum(0), st2[:r1 * 2 * Co].view(r1, 2, Co).sum(0)
    d_st = ((b1 - b2).abs().max() / b1.abs().max()).item()
    xr = raw.buf.float().view(-1, Ci)[:4096]
    ar = torch.nn.functional.mish(xr * vec[2] + vec[3])
    orf = ar.half().float() @ w.view(Co, Ci).half().float().t()
    d_ref = (out2.buf.float().view(-1, Co)[:4096] - orf).abs().max().item()
    print('   vs torch on 4096 rows: max |out - ref| %.2e, max |ref| %.2f, scale %.2f..%.2f' % (d_ref, orf.abs().max().item(), vec[2].min().item(), vec[2].max().item()))
    best = {k: [] for k in cands}
    for _ in range(rounds):
        for k, f in cands.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                f()
            e.record(); e.synchronize()
            best[k].append(s.elapsed_time(e) * 1e3 / reps)
    med = {k: sorted(v)[len(v) // 2] for k, v in best.items()}
    mb = M * 2 / 1e6
    print('batch %d, %d -> %d 1x1 @%d (pre-BN %.0f MB, out %.0f MB): ' % (N, Ci, Co, H, mb * Ci, mb * Co) +
          ', '.join('%s %.1f us' % (k, v) for k, v in med.items()) +
          ' | consumer-side saves %.1f us (%.1f %%); max |d act| %.2e, |d out| %.2e of max |out| %.1f, rel d stats %.1e'
          % (med['bn pass + conv'] - med['consumer-side'], 100 * (1 - med['consumer-side'] / med['bn pass + conv']), d_act, d_out,
             out.buf.float().abs().max().item(), d_st), flush=True)
