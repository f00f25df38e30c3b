#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_env.sh VAR "val1 val2 ..." [rounds=2] [bench args...]
var=$1; vals=$2; rounds=${3:-2}; shift 3
for i in $(seq $rounds); do
  for v in $vals; do
    env $var=$v python bench.py --no-extra --no-cpu-baseline --steps 20 "$@" > /tmp/ab_$$.json 2>/tmp/ab_$$.err || { tail -5 /tmp/ab_$$.err; continue; }
    python -c "import json; d=json.load(open('/tmp/ab_$$.json')); print('$var=$v', d['value'], d['ms_per_step'])"
  done
done
