"""Timeline of one steady-state train step from a rocprofv3 --kernel-trace CSV: wall time between two Adam launches, busy
time per HIP queue, idle gaps on the main queue, the part of the side stream (weight gradients + fold) that is exposed
after the main stream's last backward kernel, and the time per kernel family on the main queue.
usage: python tools/timeline.py <kernel_trace.csv> [out.txt]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if 'adam_multi' in r['Kernel_Name']]
assert len(adam) >= 3, 'need at least three steps'
lo, hi = adam[-2], adam[-1]          # one full step: after the second-to-last Adam up to and including the last
step = rows[lo + 1:hi + 1]
t0, t1 = rows[lo]['e'], rows[hi]['e']
wall = (t1 - t0) / 1e6
out = []
out.append('step wall (end of Adam to end of next Adam): %.3f ms, %d kernels' % (wall, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append(r)
main_q = max(byq, key=lambda q: len(byq[q]))


def fam(n):
    for f, pat in (('conv fwd/dgrad', 'igemm'), ('conv fwd/dgrad', 'conv3x3_slab'), ('conv fwd/dgrad', 'direct_s2dgrad'), ('conv fwd/dgrad', 'direct3x3'), ('conv fwd/dgrad', 'direct1x1'), ('wgrad', 'wgrad_dma'), ('wgrad fold', 'wgrad_reduce'), ('bn_act_fwd', 'bn_act_fwd'),
                   ('bn_bwd_reduce', 'bn_bwd_reduce'), ('bn_bwd_apply', 'bn_bwd_apply'), ('adam', 'adam_multi'), ('pack', 'pack_weights')):
        if pat in n:
            return f
    return 'other (head, pools, copies)'


for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r['e'] - r['s'] for r in rs) / 1e6
    out.append('queue %s%s: %d kernels, busy %.3f ms (%.0f %% of the step), first start +%.3f ms, last end +%.3f ms' % (
        q, ' (main)' if q == main_q else '', len(rs), busy, 100 * busy / wall, (rs[0]['s'] - t0) / 1e6, (max(r['e'] for r in rs) - t0) / 1e6))
m = byq[main_q]
gaps = [(b['s'] - a['e']) / 1e3 for a, b in zip(m, m[1:])]
gaps.insert(0, (m[0]['s'] - t0) / 1e3)
pos = [g for g in gaps if g > 0]
out.append('main queue: idle between consecutive kernels %.3f ms in total (%d gaps, median %.2f us); gaps > 10 us: %d, together %.3f ms' % (
    sum(pos) / 1e3, len(pos), sorted(pos)[len(pos) // 2] if pos else 0, sum(g > 10 for g in pos), sum(g for g in pos if g > 10) / 1e3))
big = sorted(((g, i) for i, g in enumerate(gaps) if g > 10), reverse=True)[:8]
others = [r for q, rs in byq.items() if q != main_q for r in rs]
for g, i in big:
    prev = m[i - 1]['Kernel_Name'][:60] if i else '(step start)'
    a, b = (m[i - 1]['e'] if i else t0), m[i]['s']
    ov = sum(max(0, min(b, r['e']) - max(a, r['s'])) for r in others)
    out.append('   gap %7.1f us at +%.3f ms before %-44s after %-44s (other queues busy %.0f %% of it)' % (
        g, (a - t0) / 1e6, m[i]['Kernel_Name'][:44], prev[:44], 100.0 * ov / max(1, b - a)))
# what the other queues run after the main queue's last backward kernel (the exposed tail before Adam)
ad = next(i for i, r in enumerate(m) if 'adam_multi' in r['Kernel_Name'])
a, b = m[ad - 1]['e'], m[ad]['s']
out.append('tail between the last backward kernel on the main queue and Adam: %.1f us; kernels of the other queues in it:' % ((b - a) / 1e3))
for r in others:
    if r['e'] > a and r['s'] < b:
        out.append('   +%8.1f us .. +%8.1f us  %s' % ((r['s'] - a) / 1e3, (r['e'] - a) / 1e3, r['Kernel_Name'][:90]))
f = collections.OrderedDict()
for r in m:
    k = fam(r['Kernel_Name'])
    f.setdefault(k, [0, 0.0])
    f[k][0] += 1
    f[k][1] += (r['e'] - r['s']) / 1e6
out.append('main queue by family:')
for k, (n, ms) in sorted(f.items(), key=lambda kv: -kv[1][1]):
    out.append('   %-28s %4d launches  %7.3f ms' % (k, n, ms))
for q, rs in byq.items():
    if q == main_q:
        continue
    # overlap of this queue's kernels with main-queue kernels
    ov = 0
    j = 0
    for r in rs:
        for mm in m:
            if mm['e'] <= r['s']:
                continue
            if mm['s'] >= r['e']:
                break
            ov += min(r['e'], mm['e']) - max(r['s'], mm['s'])
    busy = sum(r['e'] - r['s'] for r in rs)
    out.append('queue %s: %.3f of its %.3f busy ms run while a main-queue kernel is running' % (q, ov / 1e6, busy / 1e6))
txt = '\n'.join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + '\n')
