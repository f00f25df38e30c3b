#!/bin/bash
# Snapshot a commit's tree (with its library built here by hipcc) under gpurun_ab/<name> for a same-box A/B against the
# working tree (tools/ab_tree.sh): tools/ab_snapshot.sh <commit> <name>
set -e
c=$1; n=$2
rm -rf gpurun_ab/$n; mkdir -p gpurun_ab/$n
git archive $c | tar -x -C gpurun_ab/$n
rm -rf gpurun_ab/$n/tests/golden gpurun_ab/$n/profiles gpurun_ab/$n/gpurun_ab
(cd gpurun_ab/$n && python -c "import __graft_entry__ as g; print(g.build())" | tail -1)
