#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4m
python -m pytest tests/test_gpu_r4.py -q -m gpu -k "two_phase_equals" 2>&1 | tail -3
CY_TUNE_REPS=8 python tools/make_tune_cache.py gpurun_out/r4m/tune_gfx950.json > gpurun_out/r4m/tune.log 2>&1; tail -1 gpurun_out/r4m/tune.log
cp gpurun_out/r4m/tune_gfx950.json complex-yolov4-pytorch_amd/tune_cache/gfx950.json
bash tools/gpu_ab_env.sh CY_X=0 CY_WGRAD_BATCH=2 CY_WGRAD_BATCH=4 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0
