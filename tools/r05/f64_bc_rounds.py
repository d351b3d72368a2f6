"""r05 experiment: do the value-only float64 tile kernels gain from more than ONE round of waves per launch?  The bench workload's boundary terms
are 65,536 points each = 2,048 waves = exactly one round at two waves per SIMD (all waves start together and stay in step: GEMM phases and
activation phases of the two waves of a SIMD coincide).  Times the float64 evaluation of the 2-D Poisson problem with a small interior set and
boundary sets of 65,536 / 131,072 / 262,144 points: if the time per boundary point falls with the set size, merging the four terms' launches pays."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import pinn_import
m = pinn_import.load()
from neuralpde_jl_amd import workloads
for nb in (65536, 131072, 262144):
    wl = workloads.cfg2_poisson2d(points=1024, bcs_points=nb)
    rep = m.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    eng.set_option("precision", "f64")
    th = np.asarray(rep.flat_init_params, dtype=np.float64)
    wall = []
    for i in range(25):
        t0 = time.perf_counter()
        eng.loss_grad_f64(th)
        wall.append(time.perf_counter() - t0)
    t = np.median(wall[5:]) * 1e3
    print(f"boundary sets of {nb:7d} points: {t:.3f} ms per evaluation ({eng.get_option('f64_path')}), {t * 1e6 / (4 * nb):.2f} ns per boundary point", flush=True)
    del rep, eng
