#!/bin/bash
# Ablation builds of conv_pipe.hip (CY_ABL bits: 1 no MFMAs, 2 no fragment reads, 4 no DMA, 8 no barriers in the K loop, 16 no
# epilogue, 32 no K loop) linked against the other objects of the in-tree build into tools/_abl/lib_<n>.so; run a tool with
# CY_LIBPATH=tools/_abl/lib_<n>.so.  Timing instruments only: every variant but 0 computes garbage.
set -e
cd "$(dirname "$0")/.."
C=complex-yolov4-pytorch_amd/csrc
mkdir -p tools/_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$C -Wno-unused-value -DCY_ABL=$n -c $C/conv_pipe.hip -o tools/_abl/conv_pipe_$n.o &
done
wait
for n in "$@"; do
  objs=$(ls $C/*.o | grep -v conv_pipe.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_abl/lib_$n.so $objs tools/_abl/conv_pipe_$n.o
done
ls -la tools/_abl/*.so
