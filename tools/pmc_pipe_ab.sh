cd /root/repo
bash tools/pmc_micro.sh old PIPE_CFG=0,0,0,0,0 -- 16 512 1024 3 1 19 5 fwd > gpurun_out/r2_pmc_a.log 2>&1
bash tools/pmc_micro.sh c192 PIPE_CFG=2,192,128,4,0 -- 16 512 1024 3 1 19 5 fwd >> gpurun_out/r2_pmc_a.log 2>&1
bash tools/pmc_micro.sh c256 PIPE_CFG=2,256,128,4,0 -- 16 512 1024 3 1 19 5 fwd >> gpurun_out/r2_pmc_a.log 2>&1
bash tools/pmc_micro.sh c256mid PIPE_CFG=2,256,128,0,0 -- 16 512 1024 3 1 19 5 fwd >> gpurun_out/r2_pmc_a.log 2>&1
