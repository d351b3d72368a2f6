"""Small problems (the reference's own regime) under the DEFAULT precision policy: microseconds per host-entry evaluation (pinn_loss_grad_f64: theta in,
losses + gradient out) and per resident Adam iteration, float64 evaluation mode against the fp32 kernels on the same problem.
usage: python tools/r06/time_small_f64.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pinn_import
npde = pinn_import.load()
from neuralpde_jl_amd import workloads
import test_emu_parity as tp


def lorenz():
    (t,) = npde.parameters("t")
    sg, rho, beta = npde.parameters("sigma_ rho beta")
    xv, yv, zv = npde.variables("x y z")
    Dt = npde.Differential(t)
    eqs = [npde.Eq(Dt(xv(t)), sg * (yv(t) - xv(t))), npde.Eq(Dt(yv(t)), xv(t) * (rho - zv(t)) - yv(t)), npde.Eq(Dt(zv(t)), xv(t) * yv(t) - beta * zv(t))]
    bcs = [npde.Eq(xv(0), 1.0), npde.Eq(yv(0), 0.0), npde.Eq(zv(0), 0.0)]
    sysm = npde.PDESystem(eqs, bcs, [npde.In(t, npde.Interval(0.0, 1.0))], [t], [xv(t), yv(t), zv(t)], ps=[sg, rho, beta], defaults={sg: 1.0, rho: 1.0, beta: 1.0})
    chains = [npde.Chain(npde.Dense(1, 12, "tanh"), npde.Dense(12, 12, "sigmoid"), npde.Dense(12, 1)) for _ in range(3)]
    rng = np.random.default_rng(100)
    th = np.concatenate([npde.initialparameters(rng, c) for c in chains])
    return sysm, chains, npde.GridTraining(0.05), th, True


def problems():
    wl = workloads.cfg1_poisson1d()
    yield "cfg1 3x32, 1,026 points", wl.pde_system, wl.chains, npde.GridTraining(1.0 / 1024), None, False
    sysm, chain = tp.poisson2d(npde, "tanh", width=16, hidden=2)
    yield "poisson2d 2x16, 165 points", sysm, [chain], npde.GridTraining(0.1), tp.theta_for(chain, 5), False
    sysm, chain = tp.poisson2d(npde, "tanh", width=32, hidden=3)
    yield "poisson2d 3x32, 1,400 points", sysm, [chain], npde.QuasiRandomTraining(1000, bcs_points=100, sampling_alg=npde.SobolSample(seed=3), resampling=False, minibatch=1), tp.theta_for(chain, 6), False
    s, c, st, th, pe = lorenz()
    yield "lorenz 3 nets 2x12 + 3 parameters, 21 points per term", s, c, st, th, pe


for name, sysm, chains, strat, th0, pe in problems():
    row = []
    for prec in ("f32", "f64"):
        kw = {"init_params": th0} if th0 is not None else {}
        disc = npde.PhysicsInformedNN(chains if len(chains) > 1 else chains[0], strat, precision=prec, param_estim=pe, **kw)
        rep = npde.symbolic_discretize(sysm, disc)
        eng = rep.engine
        th = np.asarray(rep.flat_init_params, dtype=np.float64)
        f = (lambda: eng.loss_grad_f64(th)) if prec == "f64" else (lambda: eng.loss_grad(th.astype(np.float32)))
        for _ in range(50): f()
        t0 = time.perf_counter()
        for _ in range(300): f()
        ev = (time.perf_counter() - t0) / 300 * 1e6
        run = (lambda n: eng.adam_f64(th, n, 1e-3)) if prec == "f64" else (lambda n: eng.adam(th.astype(np.float32), n, 1e-3))
        run(50)
        t0 = time.perf_counter()
        run(500)
        it = (time.perf_counter() - t0) / 500 * 1e6
        row.append((ev, it, eng.K, eng.get_option("adam_path") if prec == "f32" else eng.get_option("f64_path")))
    (e32, i32, K, p32), (e64, i64, _, p64) = row
    print(f"{name:55s} K = {K}:  evaluation fp32 {e32:7.1f} us  float64 {e64:7.1f} us ({e64 / e32:4.1f}x)   |   Adam iteration fp32 {i32:7.1f} us [{p32}]  float64 {i64:7.1f} us [{p64}] ({i64 / i32:4.1f}x)", flush=True)
