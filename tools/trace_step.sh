#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> tools/timeline.py summary in gpurun_out/<tag>_timeline.txt
tag=${1:-r02}; shift
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/trace_$tag
rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -- python $root/bench.py --worker --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-extra "$@" > $root/gpurun_out/trace_$tag.log 2>&1)
python tools/timeline.py "$(find $out -name '*kernel_trace.csv' | head -1)" $root/gpurun_out/${tag}_timeline.txt
find $out -name "*.csv" -size +2M -delete
