#!/bin/bash
# Same-box A/B of two source trees (each with its own built library): tools/ab_tree.sh <treeA> <treeB> [rounds=2] [bench args...]
# e.g. tools/ab_tree.sh gpurun_ab/r2 . 2      (gpurun_ab/ travels to the GPU box but stays out of git)
a=$1; b=$2; rounds=${3:-2}; shift 3
for i in $(seq $rounds); do
  for t in $a $b; do
    (cd $t && python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20 --warmup 6 "$@" 2>/tmp/abt_$$.err | tail -1 > /tmp/abt_$$.json) || { tail -5 /tmp/abt_$$.err; continue; }
    python -c "import json; d=json.load(open('/tmp/abt_$$.json')); print('$t', d['value'], d['ms_per_step'])"
  done
done
