#!/bin/bash
# Everything profiles/ holds for round 4, in one GPU session: tune table for the current sources, per-step kernel tables
# (rocprofv3 --kernel-trace --stats, differential; two streams as benchmarked, and single stream = exclusive durations), HBM
# traffic (PMC, separate passes), per-layer conv profile, host enqueue time eager / replayed.  Copy gpurun_out/r04/* into profiles/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
CY_TUNE_REPS=8 python tools/make_tune_cache.py gpurun_out/r04/tune_gfx950.json > gpurun_out/r04/tune.log 2>&1; tail -1 gpurun_out/r04/tune.log
cp gpurun_out/r04/tune_gfx950.json complex-yolov4-pytorch_amd/tune_cache/gfx950.json
bash tools/rocprof_bench.sh r04 > gpurun_out/r04/rocprof.out 2>&1
cp gpurun_out/r04_per_step.txt gpurun_out/r04/r04_per_step_kernels.txt
cp gpurun_out/r04_kernel_stats_a.csv gpurun_out/r04/r04_bench_b16_f16_kernel_stats_4steps.csv; cp gpurun_out/r04_kernel_stats_b.csv gpurun_out/r04/r04_bench_b16_f16_kernel_stats_24steps.csv
CY_WGRAD_SIDE_STREAM=0 bash tools/rocprof_bench.sh r04ss > gpurun_out/r04/rocprof_ss.out 2>&1
cp gpurun_out/r04ss_per_step.txt gpurun_out/r04/r04_per_step_kernels_single_stream.txt
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_traffic.sh > gpurun_out/r04/pmc.out 2>&1
cp gpurun_out/pmc_hbm_traffic.json gpurun_out/r04/r04_pmc_hbm_traffic.json
CY_WGRAD_SIDE_STREAM=0 python tools/layer_profile.py 16 608 > gpurun_out/r04/r04_layer_profile.txt 2>&1
(echo "== CY_PLAN_REPLAY=0 (every call from Python)"; CY_PLAN_REPLAY=0 python tools/enqueue_probe.py 2>&1 | grep -v amdgpu; echo "== default (recorded launch lists, cy_run_plan)"; python tools/enqueue_probe.py 2>&1 | grep -v amdgpu) > gpurun_out/r04/r04_enqueue.txt
rm -rf gpurun_out/prof_r04_a gpurun_out/prof_r04_b gpurun_out/prof_r04ss_a gpurun_out/prof_r04ss_b gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -14 gpurun_out/r04/r04_per_step_kernels_single_stream.txt; tail -3 gpurun_out/r04/pmc.out; tail -4 gpurun_out/r04/r04_layer_profile.txt; cat gpurun_out/r04/r04_enqueue.txt
