#!/bin/bash
# HBM traffic per kernel family for the bench workload: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
# passes as MI355X_MICROARCH.md prescribes), aggregated per launch (x launches / steps_counted = per step).  FETCH_SIZE is doubled (gfx950
# reports 64 B per 128-B request for wide coalesced reads).  Only dispatches after the second optimizer launch are counted, so
# one-time tuning launches (none when the persisted tune table is valid) stay out.  Writes gpurun_out/pmc_hbm_traffic.json; copy it to profiles/ to commit.
export TMPDIR=/tmp
root=$(pwd)
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_$c
  rm -rf $out
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $out -- python $root/bench.py --worker --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra > $root/gpurun_out/pmc_$c.log 2>&1)
done
python - "$root/gpurun_out" "$root" <<'PY'
import csv, glob, json, os, sys, collections
root = sys.argv[1]
sys.path.insert(0, sys.argv[2])
import bench
FAM = [('igemm', ('igemm_fast_kernel', 'igemm_kernel', 'igemm_pipe_kernel', 'direct3x3_kernel', 'direct1x1_kernel', 'direct_s2dgrad_kernel', 'conv3x3_slab')), ('wgrad', ('wgrad_dma_kernel', 'wgrad_kernel')),
       ('wgrad_reduce', ('wgrad_reduce',)), ('bn_act_fwd', ('bn_act_fwd',)), ('bn_bwd_reduce', ('bn_bwd_reduce',)),
       ('bn_bwd_apply', ('bn_bwd_apply',)), ('bn_bwd_finalize', ('bn_bwd_finalize',)), ('bn_finalize', ('bn_finalize',)),
       ('pack_weights', ('pack_weights',)), ('adam', ('adam_multi',))]
def fam(name):
    for f, keys in FAM:
        if any(k in name for k in keys):
            return f
    return 'other'
agg = collections.defaultdict(lambda: dict(launches=0, FETCH_SIZE=0.0, WRITE_SIZE=0.0))
STEADY = 3     # the run is 2 warm-up + 3 timed steps; only dispatches after the 2nd Adam launch are counted, so the one-time
               # tile-tuning launches of the first step do not enter the per-launch averages
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = []
    for f in glob.glob('%s/pmc_%s/**/*counter_collection.csv' % (root, c), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == c]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    adam = [int(r['Dispatch_Id']) for r in rows if 'adam_multi' in r['Kernel_Name']]
    assert len(adam) == 2 + STEADY, adam
    n = collections.Counter()
    for r in rows:
        if int(r['Dispatch_Id']) <= adam[1]:
            continue
        k = fam(r['Kernel_Name'])
        agg[k][c] += float(r['Counter_Value'])
        n[k] += 1
    for k, v in n.items():
        agg[k]['launches'] = max(agg[k]['launches'], v)
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-units of 1024? -> the counters are in KB (1 KB = 1024 B) per the tool's derived metric
out = {'steps_counted': STEADY, 'kernel_sources_sha': bench.kernel_sources_sha(), 'git_head': os.environ.get('GIT_HEAD', 'unknown'),
       'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --worker --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra'}
for k, v in sorted(agg.items()):
    L = max(1, v['launches'])
    out[k] = dict(launches=v['launches'], fetch_bytes_per_launch_corrected=2.0 * 1024.0 * v['FETCH_SIZE'] / L,
                  write_bytes_per_launch=1024.0 * v['WRITE_SIZE'] / L)
json.dump(out, open(root + '/pmc_hbm_traffic.json', 'w'), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print('%-16s launches %5d  fetch %8.2f MB  write %8.2f MB per launch' % (k, v['launches'], v['fetch_bytes_per_launch_corrected'] / 1e6, v['write_bytes_per_launch'] / 1e6))
PY
find $root/gpurun_out/pmc_FETCH_SIZE $root/gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +1M -delete
