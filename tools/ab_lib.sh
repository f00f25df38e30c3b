#!/bin/bash
# A/B of two builds of libcyolo_hip.so on the same box: tools/ab_lib.sh <a.so> <b.so> [rounds=2] [bench args...]
a=$1; b=$2; rounds=${3:-2}; shift 3
lib=complex-yolov4-pytorch_amd/csrc/libcyolo_hip.so
cp $lib /tmp/keep_$$.so
for i in $(seq $rounds); do
  for v in $a $b; do
    cp $v $lib
    python bench.py --no-extra --no-cpu-baseline --no-roofline --steps 20 "$@" > /tmp/abl_$$.json 2>/tmp/abl_$$.err || { tail -5 /tmp/abl_$$.err; continue; }
    python -c "import json; d=json.load(open('/tmp/abl_$$.json')); print('$v', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/keep_$$.so $lib
