"""The persistent training kernel alone (BASELINE config 1, 50 + 2,000 resident Adam iterations = two launches of k_train) for a
rocprofv3 --pmc pass:   cd /tmp && rocprofv3 --kernel-trace --pmc <counters> -d <out> -o p -- python <repo>/tools/train_kernel_pmc.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads
wl = workloads.cfg1_poisson1d()
prob = npde.discretize(wl.pde_system, wl.discretization())
res = npde.solve(prob, npde.Adam(1e-3), maxiters=50)
res = npde.solve(npde.remake(prob, u0=res.u), npde.Adam(1e-3), maxiters=2000)
print(prob.pinnrep.engine.get_option("adam_path"), res.losses[-1])
