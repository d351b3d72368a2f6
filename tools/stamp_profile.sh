#!/bin/bash
# Builds a PROFILING variant of the family-2 H=64 kernels with per-phase cycle stamps (PINN_STAMP) into
# neuralpde.jl_amd/csrc/abl/libpinn_stamp.so; tools/stamp_report.py prints the per-phase breakdown on a GPU box.
set -e
cd "$(dirname "$0")/../neuralpde.jl_amd/csrc"
mkdir -p abl build/abl
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -DPINN_STAMP $EXTRA"
for f in engine.cpp descriptor.cpp program.cpp plan.cpp ${INST:-inst2_h64_d2.hip inst2_lapc.hip}; do
  x=""; [ "${f##*.}" = cpp ] && x="-x hip"
  $HIPCC $FLAGS $x -c $f -o build/abl/stamp_$(basename $f).o &
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o abl/libpinn_stamp${TAG}.so build/abl/stamp_*.o
ls -la abl/libpinn_stamp${TAG}.so
