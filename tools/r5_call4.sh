#!/bin/bash
# full GPU suite on the current tree, then the bench loop WITH a concurrent rocm-smi poller (what the driver does around its run)
out=gpurun_out/c4; mkdir -p $out
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25 > $out/pytest_full.log
tail -n 3 $out/pytest_full.log
( while true; do /opt/rocm/bin/rocm-smi --showuse --showmemuse --showpower --json > $out/smi.last 2>&1; sleep 0.3; done ) &
SMI=$!
bash tools/r5_fault_loop.sh c4loop ${1:-30} 1 > $out/loop.log 2>&1
kill $SMI
grep -c "retries 0" gpurun_out/c4loop/summary.txt; grep -v "retries 0" gpurun_out/c4loop/summary.txt | cut -c1-800
head -c 600 $out/smi.last
