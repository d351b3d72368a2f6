"""A/B timing of two builds of libpinn_hip.so on the same GPU box: the fused kernels of the bench workload, alternating between
neuralpde.jl_amd/csrc/abl/libpinn_other.so (copy the reference build there) and the current library.  Usage: python tools/ab_compare.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
for tag, path in (("other", "neuralpde.jl_amd/csrc/abl/libpinn_other.so"), ("head", None), ("other", "neuralpde.jl_amd/csrc/abl/libpinn_other.so"), ("head", None)):
    m._lib.set_library(m.Library(path) if path else None)
    wl = workloads.cfg2_poisson2d(points=65536)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    th = torch.tensor(wl.theta, dtype=torch.float32, device="cuda"); out = torch.zeros(eng.P + eng.K, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    eng.set_timing(1, -1)
    ts = []
    for i in range(60):
        eng.loss_grad_device(th.data_ptr(), out.data_ptr(), None, st.cuda_stream); torch.cuda.synchronize()
        ts.append([g["ms"] for g in eng.group_timings()])
    ts = np.array(ts[10:]) * 1e3
    print(tag, "interior %.1f us  bc %.1f us" % (np.median(ts[:, 0]), np.median(ts[:, 1])), flush=True)
    del rep, eng
