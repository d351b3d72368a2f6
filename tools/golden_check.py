#!/usr/bin/env python3
"""Full-size goldens (tests/golden/*_full.npz) through a given build of the library: PINN_LIB=<path> python tools/golden_check.py cfg2_full cfg3_full
Prints the relative errors of the term losses and of the gradient (L2, Linf) against the float64 oracle outputs; exit code 1 above 1e-5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pinn_import
npde = pinn_import.load()
if os.environ.get("PINN_LIB"):
    npde._lib.set_library(npde.Library(os.environ["PINN_LIB"]))
from neuralpde_jl_amd import workloads
makers = {"cfg2_full": lambda: workloads.cfg2_poisson2d(points=65536), "cfg3_full": lambda: workloads.cfg3_burgers(points=262144),
          "cfg4_full": lambda: workloads.cfg4_cavity(points=262144, bcs_points=32768), "cfg5_full": lambda: workloads.cfg5_heat_inverse(points=1000000, bcs_points=65536)}
bad = 0
for name in sys.argv[1:] or ["cfg2_full", "cfg3_full"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    wl = makers[name]()
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    losses, grad = rep.engine.loss_grad(g["theta"], g["weights"])
    gref = g["grad_stencil"]
    le = np.abs(losses - g["losses_stencil"]) / np.abs(g["losses_stencil"])
    g2 = np.linalg.norm(grad - gref) / np.linalg.norm(gref)
    gi = np.max(np.abs(grad - gref)) / np.max(np.abs(gref))
    l2, gr2 = rep.engine.loss_grad(g["theta"], g["weights"])
    det = np.array_equal(l2, losses) and np.array_equal(gr2, grad)
    print(f"{name}: loss rel err {le.max():.2e}, grad rel L2 {g2:.2e}, Linf {gi:.2e}, bit-reproducible {det}   [{rep.engine.L.path.split('/')[-1]}]", flush=True)
    bad += not (le.max() < 1e-5 and g2 < 1e-5 and gi < 1e-5 and det)
sys.exit(1 if bad else 0)
