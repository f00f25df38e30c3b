# SQ counters of the conv kernel variants on one long-K shape (A/B of the two kernel families and of the loader/compute split)
cd /root/repo
: > gpurun_out/r2_pmc_b.log
bash tools/pmc_micro.sh c256 PIPE_CFG=2,256,128,0,0 -- 16 512 1024 3 1 19 5 fwd >> gpurun_out/r2_pmc_b.log 2>&1
bash tools/pmc_micro.sh c256lc PIPE_CFG=2,256,128,3,0 -- 16 512 1024 3 1 19 5 fwd >> gpurun_out/r2_pmc_b.log 2>&1
bash tools/pmc_micro.sh c128lc PIPE_CFG=2,128,128,3,0 -- 16 512 512 3 1 19 5 fwd >> gpurun_out/r2_pmc_b.log 2>&1
