#!/bin/bash
# Everything profiles/ holds for round 3, in one GPU session: per-step kernel tables (rocprofv3 --kernel-trace --stats,
# differential; as benchmarked = two streams, and single stream = exclusive durations), step timeline, HBM traffic (PMC, separate
# passes), per-layer conv profile, host enqueue time.  Copy gpurun_out/r03_* into profiles/.
cd "$(dirname "$0")/.."
export CY_WGRAD_ATOMIC=0
bash tools/rocprof_bench.sh r03 > gpurun_out/r03_rocprof.out 2>&1
cp gpurun_out/r03_per_step.txt gpurun_out/r03_per_step_kernels.txt
CY_WGRAD_SIDE_STREAM=0 bash tools/rocprof_bench.sh r03ss > gpurun_out/r03ss_rocprof.out 2>&1
cp gpurun_out/r03ss_per_step.txt gpurun_out/r03_per_step_kernels_single_stream.txt
bash tools/trace_step.sh r03 > /dev/null 2>&1
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_traffic.sh > gpurun_out/r03_pmc.out 2>&1
cp gpurun_out/pmc_hbm_traffic.json gpurun_out/r03_pmc_hbm_traffic.json
CY_WGRAD_SIDE_STREAM=0 python tools/layer_profile.py 16 608 > gpurun_out/r03_layer_profile.txt 2>&1
python tools/enqueue_probe.py > gpurun_out/r03_enqueue.txt 2>&1
tail -3 gpurun_out/r03_pmc.out; tail -4 gpurun_out/r03_layer_profile.txt; tail -5 gpurun_out/r03_enqueue.txt
