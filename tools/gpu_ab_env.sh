#!/bin/bash
# same-box A/B of the default bench step under environment settings: tools/gpu_ab_env.sh "VAR=a" "VAR=b" ... (3 alternations)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
python -c "import torch; print(torch.cuda.Stream(priority=-5).priority, torch.cuda.Stream(priority=5).priority)"
for i in 1 2 3; do
  for e in "$@"; do
    echo -n "$e: "; env $e python bench.py --steps 30 --warmup 6 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/ab/ab_$(date +%s).txt
