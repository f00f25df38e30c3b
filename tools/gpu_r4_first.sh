#!/bin/bash
# first GPU session of round 4: the new parity tests, the tune table for the current sources, a quick bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_r4.py -x -q -s -m gpu > gpurun_out/r4a/tests_r4.log 2>&1; echo "tests rc $?" >> gpurun_out/r4a/tests_r4.log
CY_TUNE_REPS=8 python tools/make_tune_cache.py gpurun_out/r4a/tune_gfx950.json > gpurun_out/r4a/tune.log 2>&1
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r4a/bench.log 2>&1
tail -5 gpurun_out/r4a/tests_r4.log; tail -3 gpurun_out/r4a/tune.log; tail -c 600 gpurun_out/r4a/bench.log
