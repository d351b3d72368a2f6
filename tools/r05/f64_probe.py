"""r05: wall time of the float64 evaluation of the bench workload per library variant (timing probes of csrc/pinn_kernels5.hpp: PINN_F64M_PROBE)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
for rnd in range(2):
    for name in sys.argv[1:] or ["head"]:
        m._lib.set_library(None if name == "head" else m.Library(os.path.join(ROOT, "neuralpde.jl_amd", "csrc", "abl", f"libpinn_{name}.so")))
        wl = workloads.cfg2_poisson2d(points=65536)
        rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
        eng = rep.engine
        eng.set_option("precision", "f64")
        th = np.asarray(rep.flat_init_params, dtype=np.float64)
        for _ in range(3): eng.loss_grad_f64(th)
        t0 = time.perf_counter()
        for _ in range(8): l, g = eng.loss_grad_f64(th)
        print(f"{name:>8s} round {rnd}: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per float64 evaluation ({eng.get_option('f64_path')}); loss[0] {l[0]:.6e}", flush=True)
        del rep, eng
