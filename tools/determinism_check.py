#!/usr/bin/env python3
"""Bit-reproducibility probe on a GPU box: repeated loss+gradient evaluations of a workload must be bit-identical (fixed-order
reductions, no float atomics).  Usage: [FULL=1] python tools/determinism_check.py [cfg2|cfg3|cfg4|cfg5] [reps]; PINN_AB_LIB selects another build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads
if os.environ.get("PINN_AB_LIB"):
    npde._lib.set_library(npde.Library(os.environ["PINN_AB_LIB"]))
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
full = os.environ.get("FULL") is not None             # FULL=1: the BASELINE sizes (every workgroup runs many tiles)
wl = {"cfg4": lambda: workloads.cfg4_cavity(points=262144, bcs_points=32768) if full else workloads.cfg4_cavity(points=3000, bcs_points=400, width=128, hidden=5),
      "cfg5": lambda: workloads.cfg5_heat_inverse(points=1000000 if full else 4000, bcs_points=65536 if full else 500),
      "cfg3": lambda: workloads.cfg3_burgers(points=262144 if full else 8192),
      "cfg2": lambda: workloads.cfg2_poisson2d(points=65536 if full else 8192)}[which]()
rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
eng = rep.engine
print(eng.describe().split("term")[0])
th = rep.flat_init_params
l0, g0 = eng.loss_grad(th)
bad = 0
for i in range(reps):
    l, g = eng.loss_grad(th)
    nd = int(np.sum(g != g0))
    if nd or not np.array_equal(l, l0):
        bad += 1
        idx = np.nonzero(g != g0)[0]
        print(f"rep {i}: {nd} gradient entries differ (first at {idx[:6]}, max abs diff {np.max(np.abs(g - g0)):.3e}); losses equal: {np.array_equal(l, l0)}")
        # which parameter blocks: walk the chains' [W | b] layout
        off = 0
        for ci, ch in enumerate(wl.chains):
            for li in range(len(ch.sizes) - 1):
                for nm, n in (("W", ch.sizes[li] * ch.sizes[li + 1]), ("b", ch.sizes[li + 1])):
                    k = int(np.sum((idx >= off) & (idx < off + n)))
                    if k:
                        print(f"      net {ci} layer {li} {nm}: {k} of {n}")
                    off += n
print("deterministic" if bad == 0 else f"NON-DETERMINISTIC in {bad}/{reps} repetitions")
