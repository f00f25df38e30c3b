#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r4t
python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r4t/full.log 2>&1; echo "rc $?" >> gpurun_out/r4t/full.log
tail -30 gpurun_out/r4t/full.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
