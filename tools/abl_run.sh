#!/bin/bash
# runs tools/slab_micro.py on the ablation builds of tools/abl_build.sh (one process per build): shape indices in $1
cd "$(dirname "$0")/.."
for n in ${ABL_SET:-0 1 2 4 8 16 32 3 6 7}; do
  lib=tools/_abl/lib_$n.so
  [ $n = 0 ] && lib=complex-yolov4-pytorch_amd/csrc/libcyolo_hip.so
  [ -f $lib ] || continue
  echo "== CY_ABL=$n"
  CY_LIBPATH=$lib SLAB_HINTS=${SLAB_HINTS:-312,313} timeout 300 python tools/slab_micro.py 3 10 ${1:-0} 2>&1 | grep -v amdgpu.ids
done
