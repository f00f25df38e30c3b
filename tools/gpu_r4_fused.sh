#!/bin/bash
# two-phase conv+BN+act: operator tests, then the tune table for the new sources, then same-box A/B of the step with / without
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_r4.py -q -s -m gpu -k "two_phase or refuses or replay or inference_b32" > gpurun_out/r4e/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4e/tests.log
grep -E "^.?(two-phase|f16 eval|  images|  rows|  end|FAILED|ERROR|tests rc|[0-9]+ (passed|failed))" gpurun_out/r4e/tests.log | cut -c1-400
CY_TUNE_REPS=8 python tools/make_tune_cache.py gpurun_out/r4e/tune_gfx950.json > gpurun_out/r4e/tune.log 2>&1; tail -2 gpurun_out/r4e/tune.log
cp gpurun_out/r4e/tune_gfx950.json complex-yolov4-pytorch_amd/tune_cache/gfx950.json
CY_TUNE_CACHE=0 CY_TUNE_VERBOSE=1 python bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | grep "conv+bn" > gpurun_out/r4e/decisions.txt; wc -l gpurun_out/r4e/decisions.txt
for i in 1 2 3; do
  for m in 0 1 2; do
    echo -n "CY_CONV_BN_FUSED=$m: "; CY_CONV_BN_FUSED=$m python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r4e/ab.txt
