#!/bin/bash
# two-phase conv+BN+act: operator tests, then same-box A/B of the step with / without, per-layer timings
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_r4.py -q -s -m gpu -k "two_phase or refuses or replayed_inference" > gpurun_out/r4e/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r4e/tests.log
grep -E "^.?(two-phase|FAILED|ERROR|tests rc|[0-9]+ (passed|failed))" gpurun_out/r4e/tests.log | cut -c1-400
CY_TUNE_CACHE=0 CY_TUNE_VERBOSE=1 python bench.py --steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | grep "conv+bn" > gpurun_out/r4e/decisions.txt; wc -l gpurun_out/r4e/decisions.txt
for i in 1 2; do
  for m in 0 1 2; do
    echo -n "CY_CONV_BN_FUSED=$m: "; CY_TUNE_CACHE=0 CY_CONV_BN_FUSED=$m python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r4e/ab.txt
