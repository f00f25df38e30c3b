#!/bin/bash
# the round-4 GPU tests with their printed measurements (no -x: every test reports), the v3-tiny debug tool, the host enqueue probe
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4b
python tools/v3tiny_debug.py > gpurun_out/r4b/v3tiny.log 2>&1
python -m pytest tests/test_gpu_r4.py -q -s -m gpu "$@" > gpurun_out/r4b/tests_r4.log 2>&1; echo "tests rc $?" >> gpurun_out/r4b/tests_r4.log
grep -E "^.?(conditioned|v4 608|f32 eval|f16 eval|  rows|  end to end|  images|bench --gpus|v3-tiny|FAILED|ERROR|tests rc|[0-9]+ (passed|failed))" gpurun_out/r4b/tests_r4.log | cut -c1-450
tail -4 gpurun_out/r4b/v3tiny.log
CY_PLAN_REPLAY=0 python tools/enqueue_probe.py > gpurun_out/r4b/enqueue_eager.txt 2>&1
python tools/enqueue_probe.py > gpurun_out/r4b/enqueue_replay.txt 2>&1
tail -3 gpurun_out/r4b/enqueue_eager.txt gpurun_out/r4b/enqueue_replay.txt
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-300
CY_PLAN_REPLAY=0 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-300
