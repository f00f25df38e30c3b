#!/bin/bash
# How much of a single-stream step is NOT kernel time: untraced wall of the step (CY_WGRAD_SIDE_STREAM=0 CY_HEADS_SIDE=0)
# against the sum of exclusive kernel durations of the same configuration under rocprofv3 --kernel-trace --stats.
export TMPDIR=/tmp CY_WGRAD_SIDE_STREAM=0 CY_HEADS_SIDE=0
root=$(pwd)
python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20 > /tmp/gp.json
python -c "import json; d=json.load(open('/tmp/gp.json')); print('single-stream untraced wall: %.3f ms/step' % d['ms_per_step'])"
bash tools/rocprof_bench.sh gap > /dev/null 2>&1
head -1 gpurun_out/gap_per_step.txt
python - <<'PY'
import csv
a = list(csv.DictReader(open('gpurun_out/gap_kernel_stats_a.csv'))); b = list(csv.DictReader(open('gpurun_out/gap_kernel_stats_b.csv')))
ca = sum(float(r['Calls']) for r in a); cb = sum(float(r['Calls']) for r in b)
print('kernels per step: %.0f' % ((cb - ca) / 20))
PY
