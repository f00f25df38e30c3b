#!/bin/bash
# HBM traffic and SQ counters of ANY bench configuration (configs[3] inference and configs[4] 1024 x 1024 had a
# wall clock and nothing else): three rocprofv3 --pmc passes of `bench.py --worker <args>` (FETCH_SIZE; WRITE_SIZE; the SQ busy /
# MFMA / wave-cycle set -- separate passes as MI355X_MICROARCH.md prescribes, FETCH x 2 per its gfx950 correction), aggregated per
# kernel family over the steady-state steps: a step ends with the marker kernel (adam_multi for the train configurations,
# pp2_sweep for inference), the dispatches up to the WARM-th marker are dropped.
# usage: bash tools/pmc_config.sh <tag> <marker> <warm> <bench args...>   -> gpurun_out/<tag>_pmc.json / .txt
tag=$1; marker=$2; warm=$3; shift 3
export TMPDIR=/tmp
root=$(pwd)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  (cd /tmp && rocprofv3 --pmc $set --output-format csv -d $out -- python $root/bench.py --worker --no-cpu-baseline --no-roofline --no-extra "$@" > $root/gpurun_out/pmc_${tag}_$i.log 2>&1)
  tail -n 1 $root/gpurun_out/pmc_${tag}_$i.log | cut -c1-200
done
python - "$root/gpurun_out" "$root" "$tag" "$marker" "$warm" "$*" <<'PY'
import collections, csv, glob, json, os, sys
root, repo, tag, marker, warm, args = sys.argv[1:7]
warm = int(warm)
sys.path.insert(0, repo)
FAM = [('conv fwd/dgrad', ('igemm_fast_kernel', 'igemm_kernel', 'igemm_pipe_kernel', 'conv3x3_slab', 'direct3x3_kernel', 'direct1x1_kernel', 'direct_s2dgrad_kernel')),
       ('wgrad', ('wgrad_dma_kernel', 'wgrad_kernel')), ('wgrad fold', ('wgrad_reduce',)), ('bn_act_fwd', ('bn_act_fwd',)),
       ('bn_bwd_reduce', ('bn_bwd_reduce',)), ('bn_bwd_apply', ('bn_bwd_apply',)), ('pack', ('pack_weights',)), ('adam', ('adam_multi',)),
       ('nms', ('pp2_', 'mask_kernel', 'rank_kernel', 'greedy')), ('pools/upsample', ('maxpool', 'upsample'))]
def fam(name):
    for f, keys in FAM:
        if any(k in name for k in keys):
            return f
    return 'other'
per = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
steps = None
for p in (1, 2, 3):
    rows = []
    for f in glob.glob('%s/pmc_%s_%d/**/*counter_collection.csv' % (root, tag, p), recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print('pass %d: no counter rows (see gpurun_out/pmc_%s_%d.log)' % (p, tag, p)); continue
    disp = {}
    for r in rows:
        disp.setdefault(int(r['Dispatch_Id']), r['Kernel_Name'])
    marks = sorted(d for d, k in disp.items() if marker in k)
    assert len(marks) > warm, (marker, len(marks))
    steps = len(marks) - warm
    lo, hi = marks[warm - 1], marks[-1]
    seen = set()
    for r in rows:
        d = int(r['Dispatch_Id'])
        if d <= lo or d > hi:
            continue
        per[fam(r['Kernel_Name'])][r['Counter_Name']] += float(r['Counter_Value'])
        if p == 1 and d not in seen:
            seen.add(d); launches[fam(r['Kernel_Name'])] += 1
from complex_yolov4_pytorch_amd import tune
doc = dict(config=args, steps_counted=steps, kernel_sources_sha=tune.sources_sha(), git_head=os.environ.get('GIT_HEAD', 'unknown'),
           definitions=dict(fetch='FETCH_SIZE (KB) x 1024 x 2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md; as tools/pmc_traffic.sh)', write='WRITE_SIZE (KB) x 1024',
                            mfma_busy='SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)', lds_array_active='SQ_LDS_IDX_ACTIVE / (8 x SQ_BUSY_CYCLES)',
                            wait_any='SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave-cycles parked at s_waitcnt / s_barrier)'), families={})
lines = ['%s: %s, %d steady steps (per step below)' % (tag, args, steps), '%-18s %9s %10s %10s %9s %9s %9s' % ('family', 'launches', 'fetch GB', 'write GB', 'mfma busy', 'lds act', 'wait_any')]
tot_f = tot_w = 0.0
allc = collections.defaultdict(float)
for f, c in sorted(per.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE', 0) + kv[1].get('WRITE_SIZE', 0))):
    fetch = c.get('FETCH_SIZE', 0.0) * 1024 * 2 / steps / 1e9 if 'FETCH_SIZE' in c else None
    write = c.get('WRITE_SIZE', 0.0) * 1024 / steps / 1e9 if 'WRITE_SIZE' in c else None
    busy = c.get('SQ_BUSY_CYCLES', 0.0)
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (32 * busy) if busy else None
    la = c.get('SQ_LDS_IDX_ACTIVE', 0.0) / (8 * busy) if busy else None
    wa = c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else None
    for k, v in c.items():
        allc[k] += v
    doc['families'][f] = dict(launches_per_step=launches[f] / steps, fetch_counter_per_step=c.get('FETCH_SIZE', 0.0) / steps,
                              write_counter_per_step=c.get('WRITE_SIZE', 0.0) / steps, mfma_busy=mf, lds_array_active=la, wait_any=wa)
    lines.append('%-18s %9.1f %10s %10s %9s %9s %9s' % (f, launches[f] / steps, '%.3f' % fetch if fetch is not None else '-', '%.3f' % write if write is not None else '-',
                                                     '%.3f' % mf if mf is not None else '-', '%.3f' % la if la is not None else '-', '%.3f' % wa if wa is not None else '-'))
busy = allc.get('SQ_BUSY_CYCLES', 0.0)
doc['whole'] = dict(fetch_counter_per_step=allc.get('FETCH_SIZE', 0.0) / steps, write_counter_per_step=allc.get('WRITE_SIZE', 0.0) / steps,
                    mfma_busy=allc.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (32 * busy) if busy else None)
lines.append('raw counters per step: FETCH_SIZE %.4g KB, WRITE_SIZE %.4g KB (fetch GB = x 1024 x 2: the gfx950 correction; as tools/pmc_traffic.sh)'
             % (allc.get('FETCH_SIZE', 0.0) / steps, allc.get('WRITE_SIZE', 0.0) / steps))
if busy:
    lines.append('whole step: MFMA pipe busy %.3f' % (allc.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (32 * busy)))
open('%s/%s_pmc.json' % (root, tag), 'w').write(json.dumps(doc, indent=1))
open('%s/%s_pmc.txt' % (root, tag), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
PY
rm -rf $root/gpurun_out/pmc_${tag}_1 $root/gpurun_out/pmc_${tag}_2 $root/gpurun_out/pmc_${tag}_3
