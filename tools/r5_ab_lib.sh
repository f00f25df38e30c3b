#!/bin/bash
# Same-box A/B of library builds: alternates `python bench.py --worker` (headline leg only) between the listed libcyolo_hip.so
# builds (CY_LIBPATH), N rounds.   usage: bash tools/r5_ab_lib.sh <rounds> <tag=path> ...   ("default" = the in-tree build)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    tag=${spec%%=*}; path=${spec#*=}
    if [ "$path" = "default" ]; then unset CY_LIBPATH; else export CY_LIBPATH=$PWD/$path; fi
    line=$(python3 bench.py --worker --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1)
    echo "$tag round $r: $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print(d['value'], 'images/s', d['ms_per_step'], 'ms')" "$line")"
  done
done
