#!/bin/bash
# Builds a PROFILING variant of the family-2 H=64 kernels with per-phase cycle stamps (PINN_STAMP) into
# neuralpde.jl_amd/csrc/abl/libpinn_stamp${TAG}.so (Makefile target `variant`); tools/stamp_report.py prints the per-phase breakdown on a GPU box.
set -e
cd "$(dirname "$0")/../neuralpde.jl_amd/csrc"
make -j8 variant NAME=stamp${TAG} VFLAGS="-DPINN_STAMP $EXTRA" VINST="${INST:-inst2_h64_d2 inst2_lapc}" > /dev/null
ls -la abl/libpinn_stamp${TAG}.so
