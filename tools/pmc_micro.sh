#!/bin/bash
# usage: tools/pmc_micro.sh <tag> <env assignments...> -- <conv_micro args>
# Collects a few SQ counter sets (one rocprofv3 --pmc pass each) for the conv micro-benchmark and prints per-kernel sums.
tag=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
export TMPDIR=/tmp
root=$(pwd)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  (cd /tmp && env "${envs[@]}" rocprofv3 --pmc $set --output-format csv -d $out -- python $root/tools/conv_micro.py "$@" > /dev/null 2>&1)
done
python - "$root/gpurun_out" "$tag" <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob('%s/pmc_%s_*/**/*counter_collection.csv' % (root, tag), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'igemm' not in k and 'wgrad' not in k:
            continue
        k = k.split('(')[0][-60:]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    print(tag, k)
    for c, v in sorted(d.items()):
        print('   %-28s %14.0f per launch' % (c, v / max(1, cnt[(k, c)])))
    if 'SQ_BUSY_CYCLES' in d and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        # SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles; 1024 SIMDs = 32 per engine
        busy = d['SQ_BUSY_CYCLES'] / cnt[(k, 'SQ_BUSY_CYCLES')]
        mfma = d['SQ_VALU_MFMA_BUSY_CYCLES'] / cnt[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')]
        print('   => MFMA pipe busy %.1f %% of the kernel\'s SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)); '
              'LDS array active %.1f %% of CU-cycles' % (100 * mfma / (32 * busy),
              100 * d.get('SQ_LDS_IDX_ACTIVE', 0) / max(1, cnt[(k, 'SQ_LDS_IDX_ACTIVE')]) / (8 * busy)))
PY
