#!/bin/bash
# What the driver runs at round end, on one box: pytest -m gpu -x, smoke(), the bench command; then the bench command in a loop.
out=gpurun_out/${1:-v1}; mkdir -p $out; n=${2:-20}
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > $out/pytest.log; tail -n 2 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-260 $out/bench.json
bash tools/r5_fault_loop.sh $(basename $out)loop $n 0 > /dev/null 2>&1; echo "loop: $(grep -c 'retries 0' gpurun_out/$(basename $out)loop/summary.txt) of $n clean"; grep -v "retries 0" gpurun_out/$(basename $out)loop/summary.txt | cut -c1-600
