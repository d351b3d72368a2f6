#!/bin/bash
# Builds ablation variants of the two cfg2 hot kernels (one phase stubbed out each) for timing studies.
# Output: neuralpde.jl_amd/csrc/abl/libpinn_abl_<NAME>.so  (results are WRONG by construction; timing only)
set -e
cd "$(dirname "$0")/../neuralpde.jl_amd/csrc"
mkdir -p abl build/abl
HIPCC=/opt/rocm/bin/hipcc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable"
build_one() {
  name=$1; shift
  for f in engine.cpp inst_h64_d2_poisson.hip inst_h64_d2_value.hip; do
    x=""; [ "$f" = engine.cpp ] && x="-x hip"
    $HIPCC $FLAGS "$@" $x -c $f -o build/abl/${name}_$(basename $f).o &
  done
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o abl/libpinn_abl_${name}.so build/abl/${name}_*.o
}
build_one BASE
build_one NOACT -DPINN_ABL_NOACT
build_one NOSCR -DPINN_ABL_NOSCR
build_one NOFWD -DPINN_ABL_NOFWD
build_one NODA -DPINN_ABL_NODA
build_one NODW -DPINN_ABL_NODW
build_one NOMFMA -DPINN_ABL_NOFWD -DPINN_ABL_NODA -DPINN_ABL_NODW
build_one NOTAPE -DPINN_ABL_NOTAPE
build_one NOMEM -DPINN_ABL_NOSCR -DPINN_ABL_NOTAPE
ls -la abl/
