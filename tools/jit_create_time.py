"""Create-time cost of runtime specialisation (csrc/jit.cpp): `symbolic_discretize` of problems whose kernels are NOT in the ahead-of-time
table, with a cold kernel cache (PINN_JIT_DIR = a fresh directory: hipcc runs) and again with the warm cache (dlopen only).
    python tools/jit_create_time.py
Cases: the reference's DGM Burgers network (50 modes, 5 gated layers: family 3 is always specialised), a 200-wide 3-layer tanh chain,
a 4 x 64 chain with a mixed third derivative (generated jet rules)."""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
cache = tempfile.mkdtemp(prefix="pinn_jit_")
os.environ["PINN_JIT_DIR"] = cache
import numpy as np, sympy as sp
import pinn_import
npde = pinn_import.load()


def burgers(nu=0.05):
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    Dt, Dx, Dxx = npde.Differential(t), npde.Differential(x), npde.Differential(x) ** 2
    eq = npde.Eq(Dt(u(t, x)) + u(t, x) * Dx(u(t, x)) - nu * Dxx(u(t, x)), 0)
    bcs = [npde.Eq(u(0, x), -sp.sin(sp.pi * x)), npde.Eq(u(t, -1), 0.0), npde.Eq(u(t, 1), 0.0)]
    dom = [npde.In(t, npde.Interval(0.0, 1.0)), npde.In(x, npde.Interval(-1.0, 1.0))]
    return npde.PDESystem([eq], bcs, dom, [t, x], [u(t, x)])


def third():
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    Dt, Dx = npde.Differential(t), npde.Differential(x)
    eq = npde.Eq(Dt(u(t, x)) + Dt(Dx(Dx(u(t, x)))), 0)
    bcs = [npde.Eq(u(0, x), sp.sin(x))]
    dom = [npde.In(t, npde.Interval(0.0, 1.0)), npde.In(x, npde.Interval(-1.0, 1.0))]
    return npde.PDESystem([eq], bcs, dom, [t, x], [u(t, x)])


strat = lambda: npde.QuasiRandomTraining(256, bcs_points=64, sampling_alg=npde.SobolSample(seed=1), resampling=False, minibatch=1)
mlp = lambda d, w, h: npde.Chain(*([npde.Dense(d, w, "tanh")] + [npde.Dense(w, w, "tanh") for _ in range(h - 1)] + [npde.Dense(w, 1)]))
cases = [("DGM 50 modes x 5 layers, Burgers (2 kernels)", burgers, lambda: npde.DGM(2, 1, 50, 5, "tanh", "tanh", "identity")),
         ("Chain 3 x 200 tanh, Burgers (2 kernels)", burgers, lambda: mlp(2, 200, 3)),
         ("Chain 4 x 64 tanh, u_t + u_txx (generated jet set)", third, lambda: mlp(2, 64, 4))]
for name, mk, net in cases:
    out = []
    for rnd in ("cold", "warm"):
        t0 = time.perf_counter()
        rep = npde.symbolic_discretize(mk(), npde.PhysicsInformedNN(net(), strat(), precision="f32"))
        out.append(time.perf_counter() - t0)
        njit = rep.engine.describe().count("kernel=")
        del rep
    print(f"{name:58s} create: cold {out[0]:7.2f} s   warm {out[1]:6.3f} s   ({njit} launch groups)")
