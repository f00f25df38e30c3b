#!/bin/bash
# Everything profiles/ holds for round 5, in one GPU session on the FINAL sources: per-step kernel tables (rocprofv3 --kernel-trace
# --stats, differential; two streams as benchmarked, and single stream = exclusive durations), HBM traffic (PMC, separate
# passes), SQ counters (MFMA busy, LDS, wave-cycle shares), per-layer conv profile, the default bench line.  The measuring
# process under the tracer is `bench.py --worker` (the plain command is a supervisor without a GPU context).
# Copy gpurun_out/r05/* into profiles/.   usage: GIT_HEAD=<sha> bash tools/r05_profiles.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
bash tools/rocprof_bench.sh r05 > gpurun_out/r05/rocprof.out 2>&1
cp gpurun_out/r05_per_step.txt gpurun_out/r05/r05_per_step_kernels.txt
cp gpurun_out/r05_kernel_stats_a.csv gpurun_out/r05/r05_bench_b16_f16_kernel_stats_4steps.csv; cp gpurun_out/r05_kernel_stats_b.csv gpurun_out/r05/r05_bench_b16_f16_kernel_stats_24steps.csv
CY_WGRAD_SIDE_STREAM=0 bash tools/rocprof_bench.sh r05ss > gpurun_out/r05/rocprof_ss.out 2>&1
cp gpurun_out/r05ss_per_step.txt gpurun_out/r05/r05_per_step_kernels_single_stream.txt
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_traffic.sh > gpurun_out/r05/pmc.out 2>&1
cp gpurun_out/pmc_hbm_traffic.json gpurun_out/r05/r05_pmc_hbm_traffic.json
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_sq.sh r05 > gpurun_out/r05/sq.out 2>&1
cp gpurun_out/r05_sq_counters.json gpurun_out/r05_sq_counters.txt gpurun_out/r05/
CY_WGRAD_SIDE_STREAM=0 python tools/layer_profile.py 16 608 > gpurun_out/r05/r05_layer_profile.txt 2>&1
python bench.py > gpurun_out/r05/r05_bench_default.json 2> gpurun_out/r05/r05_bench_default.err
rm -rf gpurun_out/prof_r05_a gpurun_out/prof_r05_b gpurun_out/prof_r05ss_a gpurun_out/prof_r05ss_b gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/sq_r05_1 gpurun_out/sq_r05_2
head -14 gpurun_out/r05/r05_per_step_kernels_single_stream.txt; tail -3 gpurun_out/r05/pmc.out; head -12 gpurun_out/r05/r05_sq_counters.txt; tail -4 gpurun_out/r05/r05_layer_profile.txt; cut -c1-400 gpurun_out/r05/r05_bench_default.json
