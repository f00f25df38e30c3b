#!/bin/bash
# The kernels around every idle gap > 100 us of one steady-state step (rocprofv3 kernel trace of a short bench run).
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/trace_gap
rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out -- python $root/bench.py --worker --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-extra "$@" > $root/gpurun_out/trace_gap.log 2>&1)
python - "$(find $out -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
lo, hi = adam[-2], adam[-1]
t0 = rows[lo]["e"]
step = rows[lo+1:hi+1]
end = step[0]["e"]
for i in range(1, len(step)):
    gap = step[i]["s"] - end
    if gap > 100000:
        print("GAP %.1f us at +%.3f ms" % (gap/1e3, (step[i]["s"]-t0)/1e6))
        for r in step[max(0,i-6):i+6]:
            print("  +%.3f..%.3f ms q%s %s" % ((r["s"]-t0)/1e6, (r["e"]-t0)/1e6, r["Queue_Id"], r["Kernel_Name"][:100]))
    end = max(end, step[i]["e"])
PY
rm -rf $out
