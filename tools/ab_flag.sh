#!/bin/bash
# A/B of bench.py argument sets on the same box: tools/ab_flag.sh rounds "args A" "args B" ...
rounds=$1; shift
for i in $(seq $rounds); do
  for args in "$@"; do
    python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20 $args > /tmp/abf_$$.json 2>/tmp/abf_$$.err || { tail -5 /tmp/abf_$$.err; continue; }
    python -c "import json; d=json.load(open('/tmp/abf_$$.json')); print('[%s]' % '$args', d['value'], d['ms_per_step'], d['config'].get('dgrad_bn_sums_layers'))"
  done
done
