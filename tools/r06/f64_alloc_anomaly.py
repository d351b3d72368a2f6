"""Probe for an intermittent slowdown seen in tools/time_f64.py (profiles/r06_f64_anomaly.txt): does a handle with multi-GB float64 scratch (the bench
workload), created — and destroyed or kept — earlier in the process, slow the next handle's float64 evaluation down?  (Measured: no, 10 of 10 runs clean.)
usage: python tools/r06/f64_alloc_anomaly.py keep|free|lanes     (keep: the first handle stays alive while the second is timed; lanes: the first handle also runs the lane-per-point kernels)"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads


def make(wl):
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization(precision="f64"))
    th = np.asarray(rep.flat_init_params, dtype=np.float64)
    for _ in range(4):
        rep.engine.loss_grad_f64(th)
    return rep, th


def timed(rep, th, n=8):
    t0 = time.perf_counter()
    for _ in range(n):
        rep.engine.loss_grad_f64(th)
    return (time.perf_counter() - t0) / n * 1e3


mode = sys.argv[1] if len(sys.argv) > 1 else "free"
first, th1 = make(workloads.cfg2_poisson2d(points=65536))
t_first = timed(first, th1)
if mode == "lanes":                                  # the lane-per-point leg of tools/time_f64.py on the first handle
    os.environ["PINN_F64_NO_MFMA"] = "1"
    for _ in range(4):
        first.engine.loss_grad_f64(th1)
    del os.environ["PINN_F64_NO_MFMA"]
if mode == "free":
    first.engine.close()
    del first
    gc.collect()
second, th2 = make(workloads.cfg3_burgers(points=65536, bcs_points=8192))
print(f"{mode}: bench workload {t_first:.3f} ms, then cfg3 reduced {timed(second, th2):.3f} ms", flush=True)
