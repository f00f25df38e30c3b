#!/bin/bash
# rocprofv3 kernel table of BASELINE configs[3] (inference batch 32 + rotated NMS) and configs[4] (1024 x 1024 batch 8 train step):
# per-step kernel times by difference of two runs, like tools/rocprof_bench.sh does for the headline configuration.
# Writes gpurun_out/r06/r06_per_step_kernels_{infer32,train1024}.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
for c in infer32 train1024; do
  bash tools/rocprof_bench.sh r06$c --config $c > gpurun_out/r06/rocprof_$c.out 2>&1
  cp gpurun_out/r06${c}_per_step.txt gpurun_out/r06/r06_per_step_kernels_$c.txt
  rm -rf gpurun_out/prof_r06${c}_a gpurun_out/prof_r06${c}_b
done
head -50 gpurun_out/r06/r06_per_step_kernels_infer32.txt
