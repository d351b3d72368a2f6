#!/usr/bin/env python3
"""Time the fused cfg2 kernels for each ablation library (tools/build_ablations.sh)."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
libs = sorted(glob.glob(os.path.join(ROOT, "neuralpde.jl_amd", "csrc", "abl", "libpinn_abl_*.so")))
wl = workloads.cfg2_poisson2d(points=65536)
for path in libs:
    name = os.path.basename(path)[len("libpinn_abl_"):-3]
    m._lib.set_library(m.Library(path))
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    ts = []
    for i in range(12):
        rep.engine.loss_grad(wl.theta)
        ts.append([g["ms"] for g in rep.engine.group_timings()])
    ts = np.array(ts[2:])
    print(f"{name:8s} interior {np.median(ts[:,0])*1e3:8.1f} us   bc {np.median(ts[:,1])*1e3:8.1f} us", flush=True)
    del rep
