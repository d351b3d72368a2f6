"""Resident-theta Adam loop (pinn_adam_steps): us per iteration for BASELINE config 1 (1-D Poisson 3x32, 1,026 points: launch-bound), the
reference's 2-D Poisson test at its own size (2 x 16 net, GridTraining(0.1): 121 + 4 x 11 points) and config 2 (2-D Poisson 4x64,
65,536 + 4 x 65,536 points: kernel-bound).  Small problems run twice: the stand-alone loop (three launches per iteration, PINN_PERSISTENT=0)
and the persistent training kernel (csrc/pinn_train.hpp: every iteration inside one launch); the two must end at the SAME parameters.
PINN_GRAPH=1 replays one recorded step of the loop as a hipGraph instead of launching kernel by kernel."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads

# --lib <path>: another build of the library (an A/B variant, or a -DPINN_STAMP profiling build: then the per-phase share of an iteration of
# the persistent kernel — evaluation / barrier / update / barrier, thread 0's s_memtime ticks — is printed as well); --only <substring>
LIB = None
ONLY = None
if "--lib" in sys.argv:
    LIB = npde.Library(sys.argv[sys.argv.index("--lib") + 1])
    npde._lib.set_library(LIB)
if "--only" in sys.argv:
    ONLY = sys.argv[sys.argv.index("--only") + 1]


def small2d():
    import test_emu_parity as tp
    sysm, chain = tp.poisson2d(npde, "tanh", width=16, hidden=2)
    return sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.1), init_params=tp.theta_for(chain, 5), precision="f32")


def small2d_stochastic():
    import test_emu_parity as tp
    sysm, chain = tp.poisson2d(npde, "tanh", width=16, hidden=2)
    return sysm, npde.PhysicsInformedNN(chain, npde.StochasticTraining(128, bcs_points=32, rng=np.random.default_rng(3)), init_params=tp.theta_for(chain, 5), precision="f32")


def small2d_lhs():
    import test_emu_parity as tp
    sysm, chain = tp.poisson2d(npde, "tanh", width=32, hidden=3)
    return sysm, npde.PhysicsInformedNN(chain, npde.QuasiRandomTraining(1000, bcs_points=100, sampling_alg=npde.LatinHypercubeSample(seed=5)), init_params=tp.theta_for(chain, 5), precision="f32")


def mid2d_stochastic():
    wl = workloads.cfg2_poisson2d(points=4096, bcs_points=1024)
    disc = npde.PhysicsInformedNN(wl.chains[0], npde.StochasticTraining(4096, bcs_points=1024, rng=np.random.default_rng(3)), init_params=wl.theta, precision="f32")
    return wl.pde_system, disc


cases = [("cfg1 3x32 1,026 pts", lambda: (workloads.cfg1_poisson1d().pde_system, workloads.cfg1_poisson1d().discretization()), 5000),
         ("poisson2d 2x16 165 pts", small2d, 5000),
         ("2x16 Stochastic 256 pts", small2d_stochastic, 5000),          # points redrawn before every step (5 sampler + 1 source launches per loop step)
         ("3x32 LatinHyp. 1,400 pts", small2d_lhs, 3000),
         ("4x64 Stochastic 8,192 pts", mid2d_stochastic, 2000),           # too large for the kernel: the loop, ONE redraw launch per step (PINN_NO_FUSED_RESAMPLE=1: ten)
         ("cfg2 4x64 327,680 pts", lambda: (workloads.cfg2_poisson2d(points=65536).pde_system, workloads.cfg2_poisson2d(points=65536).discretization()), 1000)]
for name, make, iters in cases:
    if ONLY and ONLY not in name:
        continue
    ends = {}
    for mode in ("loop", "persistent"):
        os.environ["PINN_PERSISTENT"] = "0" if mode == "loop" else "1"
        sysm, disc = make()
        prob = npde.discretize(sysm, disc)
        if getattr(prob.pinnrep, "_device_samplers", None):                   # the same sampler seeds in both modes
            prob.pinnrep._device_samplers = {k: (lb, ub, n, 4321 + 17 * k, kind) for k, (lb, ub, n, _, kind) in prob.pinnrep._device_samplers.items()}
        res = npde.solve(prob, npde.Adam(1e-3), maxiters=50)                # warm-up: kernels loaded, clocks up
        t0 = time.perf_counter()
        res = npde.solve(npde.remake(prob, u0=res.u), npde.Adam(1e-3), maxiters=iters)
        dt = time.perf_counter() - t0
        path = prob.pinnrep.engine.get_option("adam_path")
        ends[mode] = res.u
        print(f"{name:24s} {mode:10s} -> ran '{path}': {iters} iterations in {dt:.3f} s = {dt / iters * 1e6:7.1f} us/iteration, final loss {res.losses[-1]:.6e}"
              f"  ({'graph replay' if os.environ.get('PINN_GRAPH') else 'plain launches'})", flush=True)
        if path == "persistent" and LIB is not None and hasattr(LIB.lib, "pinn_debug_train_stamps"):
            import ctypes as C
            st = (C.c_ulonglong * 4)()
            fn = LIB.lib.pinn_debug_train_stamps
            fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
            if fn(prob.pinnrep.engine.h, st) == 0:
                t = np.array(list(st), dtype=np.float64)
                us = dt / iters * 1e6 * t / t.sum()
                print(f"{name:24s} phase ticks per iteration (thread 0): " + "  ".join(f"{n} {x / iters:8.0f} ({u:5.1f} us)" for n, x, u in
                      zip(("evaluation", "barrier A", "update", "barrier B"), t, us)), flush=True)
        if path == "loop" and mode == "persistent":
            break
    if len(ends) == 2:
        print(f"{name:24s} parameters after {iters + 50} iterations identical: {bool(np.array_equal(ends['loop'], ends['persistent']))}", flush=True)
