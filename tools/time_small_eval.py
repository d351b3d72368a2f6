"""Host-entry loss + gradient evaluation (pinn_loss_grad: theta in over PCIe, losses + gradient out) of SMALL problems, microseconds per call:
the reference's per-iteration cost under a host-side optimiser (BFGS / L-BFGS stages, src/discretize.jl:567-598 + 778).
  one launch          residual kernel + grid barrier + fixed-order sums in ONE launch (csrc/pinn_train.hpp, evaluation-only form)   [default]
  stand-alone         residual kernel, then the reduction kernel                                                                    (PINN_NO_FUSED_EVAL=1)
  stand-alone+events  the same with the library's default HIP events around the phases (pinn_set_timing level 2: what every host-entry call
                      recorded before the Python mirror / Julia glue switched them off)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads
import test_emu_parity as tp


def problems():
    wl = workloads.cfg1_poisson1d()
    yield "cfg1 3x32 1,026 pts", wl.pde_system, wl.discretization()
    sysm, chain = tp.poisson2d(npde, "tanh", width=16, hidden=2)
    yield "poisson2d 2x16 165 pts", sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.1), init_params=tp.theta_for(chain, 5), precision="f32")
    sysm, chain = tp.poisson2d(npde, "tanh", width=32, hidden=3)
    yield "poisson2d 3x32 1,400 pts", sysm, npde.PhysicsInformedNN(chain, npde.QuasiRandomTraining(1000, bcs_points=100, sampling_alg=npde.SobolSample(seed=3), resampling=False, minibatch=1), init_params=tp.theta_for(chain, 5), precision="f32")


for name, sysm, disc in problems():
    rep = npde.symbolic_discretize(sysm, disc)
    eng = rep.engine
    th = np.ascontiguousarray(rep.flat_init_params, dtype=np.float32)
    losses = np.zeros(eng.K); grad = np.zeros(eng.P, dtype=np.float32)
    fn = eng.L.lib.pinn_loss_grad
    args = (eng.h, th.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(eng.P), None, losses.ctypes.data_as(C.POINTER(C.c_double)), grad.ctypes.data_as(C.POINTER(C.c_float)))
    ref = None
    for label, env, level in (("one launch", None, 0), ("stand-alone", "1", 0), ("stand-alone+events", "1", 2)):
        if env: os.environ["PINN_NO_FUSED_EVAL"] = env
        else: os.environ.pop("PINN_NO_FUSED_EVAL", None)
        eng.set_timing(level, -1)
        for _ in range(200): assert fn(*args) == 0
        t0 = time.perf_counter()
        n = 3000
        for _ in range(n): fn(*args)
        dt = (time.perf_counter() - t0) / n
        same = ref is None or (np.array_equal(ref[0], losses) and np.array_equal(ref[1], grad))
        ref = ref or (losses.copy(), grad.copy())
        print(f"{name:26s} {label:20s} {dt * 1e6:7.1f} us per pinn_loss_grad   path: {eng.get_option('eval_path'):20s} identical results: {same}", flush=True)
    eng.set_timing(0, -1)
