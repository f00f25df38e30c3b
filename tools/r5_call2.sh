#!/bin/bash
out=gpurun_out/c2; mkdir -p $out
# (1) round 3's reproducible fault: the captured step on ONE stream -- now under the supervisor, which maps the address
CY_WGRAD_SIDE_STREAM=0 timeout 300 python3 bench.py --graph 1 --steps 4 --warmup 2 --no-extra --no-cpu-baseline --no-roofline > $out/graph1.out 2> $out/graph1.err
echo "graph1 rc=$?" > $out/summary.txt
# (2) red-zone suite
timeout 900 python -m pytest tests/test_gpu_redzone.py -q -m gpu -s 2>&1 | tail -60 > $out/redzone.log
# (3) configs[3] f16 end-to-end numbers
timeout 600 python -m pytest tests/test_gpu_r4.py -q -m gpu -s -k "inference_b32" 2>&1 | tail -40 > $out/infer.log
# (4) dynamics
timeout 900 python -m pytest tests/test_zz_gpu_dynamics.py -q -m gpu -s 2>&1 | tail -60 > $out/dynamics.log
tail -n 5 $out/graph1.err >> $out/summary.txt; tail -c 1500 $out/graph1.out >> $out/summary.txt
tail -n 3 $out/redzone.log $out/infer.log $out/dynamics.log >> $out/summary.txt
cat $out/summary.txt
