#!/bin/bash
# the whole -m gpu suite (no -x), smoke(), and the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4t
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r4t/full.log 2>&1; echo "rc $?" >> gpurun_out/r4t/full.log
tail -16 gpurun_out/r4t/full.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r4t/bench_default.json 2> gpurun_out/r4t/bench_default.err; tail -c 3000 gpurun_out/r4t/bench_default.json
