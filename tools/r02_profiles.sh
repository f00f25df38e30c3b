#!/bin/bash
# Everything profiles/ holds for round 2, in one GPU session: per-step kernel table (rocprofv3 --kernel-trace --stats,
# differential), step timeline, HBM traffic (PMC, separate passes), SQ counters of the conv kernels on two layer shapes.
cd "$(dirname "$0")/.."
bash tools/rocprof_bench.sh r02 > gpurun_out/r02_rocprof.out 2>&1
bash tools/trace_step.sh r02 > /dev/null 2>&1
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/pmc_traffic.sh > gpurun_out/r02_pmc.out 2>&1
{
  echo "# SQ counters (rocprofv3 --pmc, four passes per configuration; tools/pmc_micro.sh) of the conv kernels, batch 16, fp16"
  echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_* in cycles."
  for shape in "16 128 128 3 1 76" "16 512 1024 3 1 19"; do
    echo "## shape (N Cin Cout ks stride H) = $shape : forward (with BN statistics), dgrad, wgrad"
    echo "### untraced timing"
    PIPE_CFG=0,0,0,0,0 python tools/conv_micro.py $shape 20 2>&1 | sed 's/^/4-wave      /'
    PIPE_CFG=2,0,0,0,0 python tools/conv_micro.py $shape 20 fwd,dgrad 2>&1 | sed 's/^/8-wave      /'
    PIPE_CFG=2,0,0,3,0 python tools/conv_micro.py $shape 20 fwd,dgrad 2>&1 | sed 's/^/8+4-wave    /'
    echo "### counters: 4-wave kernels (conv_igemm.hip) + wgrad"
    bash tools/pmc_micro.sh s4 PIPE_CFG=0,0,0,0,0 -- $shape 5
    echo "### counters: 8-wave pipelined kernel, policy tile (conv_pipe.hip)"
    bash tools/pmc_micro.sh s8 PIPE_CFG=2,0,0,0,0 -- $shape 5 fwd,dgrad
    echo "### counters: 8 compute + 4 loader waves"
    bash tools/pmc_micro.sh s12 PIPE_CFG=2,0,0,3,0 -- $shape 5 fwd,dgrad
  done
} > gpurun_out/r02_sq_counters_conv.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
tail -5 gpurun_out/r02_sq_counters_conv.txt
