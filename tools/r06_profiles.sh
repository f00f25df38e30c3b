#!/bin/bash
# Everything profiles/ holds for round 6, in one GPU session on the FINAL sources: per-step kernel tables (rocprofv3 --kernel-trace
# --stats, differential; two streams as benchmarked, and single stream = exclusive durations), HBM traffic (PMC, separate
# passes), SQ counters (MFMA busy, LDS, wave-cycle shares), per-layer conv profile, the default bench line -- and (new) the kernel
# tables and counters of BASELINE configs[3] (inference batch 32 + NMS) and configs[4] (1024 x 1024, batch 8).
# Copy gpurun_out/r06/* into profiles/.   usage: GIT_HEAD=<sha> bash tools/r06_profiles.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
bash tools/rocprof_bench.sh r06 > gpurun_out/r06/rocprof.out 2>&1
cp gpurun_out/r06_per_step.txt gpurun_out/r06/r06_per_step_kernels.txt
cp gpurun_out/r06_kernel_stats_a.csv gpurun_out/r06/r06_bench_b16_f16_kernel_stats_4steps.csv; cp gpurun_out/r06_kernel_stats_b.csv gpurun_out/r06/r06_bench_b16_f16_kernel_stats_24steps.csv
CY_WGRAD_SIDE_STREAM=0 bash tools/rocprof_bench.sh r06ss > gpurun_out/r06/rocprof_ss.out 2>&1
cp gpurun_out/r06ss_per_step.txt gpurun_out/r06/r06_per_step_kernels_single_stream.txt
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_traffic.sh > gpurun_out/r06/pmc.out 2>&1
cp gpurun_out/pmc_hbm_traffic.json gpurun_out/r06/r06_pmc_hbm_traffic.json
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_sq.sh r06 > gpurun_out/r06/sq.out 2>&1
cp gpurun_out/r06_sq_counters.json gpurun_out/r06_sq_counters.txt gpurun_out/r06/
CY_WGRAD_SIDE_STREAM=0 python tools/layer_profile.py 16 608 > gpurun_out/r06/r06_layer_profile.txt 2>&1
for c in infer32 train1024; do
  bash tools/rocprof_bench.sh r06$c --config $c > gpurun_out/r06/rocprof_$c.out 2>&1
  cp gpurun_out/r06${c}_per_step.txt gpurun_out/r06/r06_per_step_kernels_$c.txt
done
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_config.sh r06_infer32 pp2_sweep 2 --config infer32 --steps 3 --warmup 2 > gpurun_out/r06/pmc_infer32.out 2>&1
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_config.sh r06_train1024 adam_multi 2 --config train1024 --steps 3 --warmup 2 > gpurun_out/r06/pmc_train1024.out 2>&1
cp gpurun_out/r06_infer32_pmc.json gpurun_out/r06_infer32_pmc.txt gpurun_out/r06_train1024_pmc.json gpurun_out/r06_train1024_pmc.txt gpurun_out/r06/
python bench.py > gpurun_out/r06/r06_bench_default.json 2> gpurun_out/r06/r06_bench_default.err
rm -rf gpurun_out/prof_r06* gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/sq_r06_1 gpurun_out/sq_r06_2
head -14 gpurun_out/r06/r06_per_step_kernels_single_stream.txt; tail -3 gpurun_out/r06/pmc.out; head -12 gpurun_out/r06/r06_sq_counters.txt; tail -4 gpurun_out/r06/r06_layer_profile.txt; cat gpurun_out/r06/r06_infer32_pmc.txt gpurun_out/r06/r06_train1024_pmc.txt; cut -c1-400 gpurun_out/r06/r06_bench_default.json
