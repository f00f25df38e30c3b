#!/bin/bash
# BASELINE's own metric, "MFMA utilisation % (rocprof)", on the default step: SQ counters of
# `python bench.py --worker --steps 3 --warmup 2` (complex_yolov4.cfg 608x608 batch 16 f16, as benchmarked: two streams) in separate
# rocprofv3 --pmc passes (<= 8 SQ counters each, MI355X_MICROARCH.md "rocprofv3 PMC slots"), only the dispatches after the
# second optimizer launch counted (steady state), aggregated per kernel family and for the largest kernels individually.
#   MFMA pipe busy   = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)   (SQ_BUSY_CYCLES sums 32 shader engines; 1024 SIMDs)
#   LDS array active = SQ_LDS_IDX_ACTIVE / (8 x SQ_BUSY_CYCLES)            (256 CUs = 8 per engine)
# Writes gpurun_out/<tag>_sq_counters.json (+ .txt); copy to profiles/ to commit.   usage: bash tools/pmc_sq.sh [tag]
tag=${1:-r05}
export TMPDIR=/tmp
root=$(pwd)
mops=$(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU_MFMA_MOPS_F16" | head -1)
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM $mops"; do
  i=$((i+1))
  out=$root/gpurun_out/sq_${tag}_$i
  rm -rf $out
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $out -- python $root/bench.py --worker --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $root/gpurun_out/sq_${tag}_$i.log 2>&1)
  tail -n 2 $root/gpurun_out/sq_${tag}_$i.log | cut -c1-300
done
python - "$root/gpurun_out" "$root" "$tag" <<'PY'
import collections, csv, glob, json, os, sys
root, repo, tag = sys.argv[1:4]
sys.path.insert(0, repo)
import bench
FAM = [('conv fwd/dgrad', ('igemm_fast_kernel', 'igemm_kernel', 'igemm_pipe_kernel', 'direct3x3_kernel', 'direct1x1_kernel', 'direct_s2dgrad_kernel', 'conv3x3_slab')),
       ('wgrad', ('wgrad_dma_kernel', 'wgrad_kernel')), ('wgrad fold', ('wgrad_reduce',)), ('bn_act_fwd', ('bn_act_fwd',)),
       ('bn_bwd_reduce', ('bn_bwd_reduce',)), ('bn_bwd_apply', ('bn_bwd_apply',)), ('pack', ('pack_weights',)), ('adam', ('adam_multi',))]
def fam(name):
    for f, keys in FAM:
        if any(k in name for k in keys):
            return f
    return 'other'
STEADY = 3
per = collections.defaultdict(lambda: collections.defaultdict(float))     # kernel name -> counter -> sum over the steady steps
launches = collections.Counter()
for p in (1, 2):
    rows = []
    for f in glob.glob('%s/sq_%s_%d/**/*counter_collection.csv' % (root, tag, p), recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print('pass %d: no counter rows (see gpurun_out/sq_%s_%d.log)' % (p, tag, p)); continue
    disp = {}
    for r in rows:
        disp.setdefault(int(r['Dispatch_Id']), r['Kernel_Name'])
    adam = sorted(d for d, k in disp.items() if 'adam_multi' in k)
    assert len(adam) == 2 + STEADY, adam
    seen = set()
    for r in rows:
        d = int(r['Dispatch_Id'])
        if d <= adam[1]:
            continue
        c = r['Counter_Name']
        if p == 2 and c == 'SQ_BUSY_CYCLES':
            c = 'SQ_BUSY_CYCLES#2'
        per[r['Kernel_Name']][c] += float(r['Counter_Value'])
        if p == 1 and d not in seen:
            seen.add(d); launches[r['Kernel_Name']] += 1
def derive(c):
    busy, busy2 = c.get('SQ_BUSY_CYCLES', 0.0), c.get('SQ_BUSY_CYCLES#2', 0.0)
    out = {}
    if busy:
        out['mfma_busy'] = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (32.0 * busy)
        wc = c.get('SQ_WAVE_CYCLES', 0.0)
        if wc:
            out['wave_cycles_share'] = dict(active=c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, wait_inst=c.get('SQ_WAIT_INST_ANY', 0.0) / wc,
                                            wait_inst_lds=c.get('SQ_WAIT_INST_LDS', 0.0) / wc, wait_any=c.get('SQ_WAIT_ANY', 0.0) / wc)
    if busy2:
        out['lds_array_active'] = c.get('SQ_LDS_IDX_ACTIVE', 0.0) / (8.0 * busy2)
        out['lds_bank_conflict_share_of_lds_active'] = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0.0))
        mf = c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0.0)
        if mf and c.get('SQ_INSTS_LDS'):
            out['lds_insts_per_mfma_mop'] = c['SQ_INSTS_LDS'] / mf
    return out
fams = collections.defaultdict(lambda: collections.defaultdict(float))
fl = collections.Counter()
for k, c in per.items():
    for n, v in c.items():
        fams[fam(k)][n] += v
        fams['whole step'][n] += v
    fl[fam(k)] += launches[k]; fl['whole step'] += launches[k]
doc = {'kernel_sources_sha': bench.kernel_sources_sha(), 'git_head': os.environ.get('GIT_HEAD', 'unknown'), 'steps_counted': STEADY,
       'command': 'rocprofv3 --pmc <SQ set> -- python bench.py --worker --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra (two passes)',
       'definitions': {'mfma_busy': 'SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): share of the SIMD-cycles of the kernels\' busy time in which the MFMA pipe is busy',
                       'lds_array_active': 'SQ_LDS_IDX_ACTIVE / (8 x SQ_BUSY_CYCLES)'},
       'families': {}, 'kernels': {}}
lines = []
for f, c in sorted(fams.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0)):
    d = derive(c); d['launches_per_step'] = fl[f] / STEADY
    d['busy_cycles_per_step'] = c.get('SQ_BUSY_CYCLES', 0.0) / STEADY / 32.0
    doc['families'][f] = d
    lines.append('%-16s launches/step %6.1f  MFMA busy %5.1f %%  LDS array active %5.1f %%  bank-conflict share %4.1f %%' % (
        f, d['launches_per_step'], 100 * d.get('mfma_busy', 0), 100 * d.get('lds_array_active', 0), 100 * d.get('lds_bank_conflict_share_of_lds_active', 0)))
top = sorted(per.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))[:14]
for k, c in top:
    d = derive(c); d['launches_per_step'] = launches[k] / STEADY
    short = k.split('(')[0][-90:]
    doc['kernels'][short] = d
    ws = d.get('wave_cycles_share', {})
    lines.append('  %-90s x%5.1f  MFMA %5.1f %%  LDS %5.1f %% (conflicts %4.1f %%)  waves: active %4.1f %% wait_inst %4.1f %% (lds %4.1f %%) wait_any %4.1f %%' % (
        short, d['launches_per_step'], 100 * d.get('mfma_busy', 0), 100 * d.get('lds_array_active', 0), 100 * d.get('lds_bank_conflict_share_of_lds_active', 0),
        100 * ws.get('active', 0), 100 * ws.get('wait_inst', 0), 100 * ws.get('wait_inst_lds', 0), 100 * ws.get('wait_any', 0)))
json.dump(doc, open('%s/%s_sq_counters.json' % (root, tag), 'w'), indent=1)
open('%s/%s_sq_counters.txt' % (root, tag), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
find $root/gpurun_out/sq_${tag}_1 $root/gpurun_out/sq_${tag}_2 -name "*.csv" -size +1M -delete 2>/dev/null
