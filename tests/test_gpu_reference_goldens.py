"""End-to-end checks against golden data the REFERENCE's own tests hold (tests/golden/reference/): the same statement, the same
training recipe, the same acceptance threshold as the reference test, on the engine."""
import os

import numpy as np
import pytest
import sympy as sp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def test_dgm_burgers_against_the_reference_mol_table(npde, hip_lib):
    """test/DGM/dgm__burger_s_equation.jl:27-68: Burgers u_t + u u_x - 0.05 u_xx = 0, DeepGalerkin(2, 1, 50, 5, tanh, tanh, identity,
    QuasiRandomTraining(256, minibatch = 32)), Adam(0.01) x 500 then Adam(0.001) x 200, `u_predict ≈ BURGER_REF_U rtol = 0.2`
    (isapprox on matrices: ||a - b|| <= rtol * max(||a||, ||b||)) against the MethodOfLines table the reference test carries."""
    g = np.load(os.path.join(GOLD, "dgm_burgers_mol_table.npz"))
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    Dt, Dx, Dxx = npde.Differential(t), npde.Differential(x), npde.Differential(x) ** 2
    eq = npde.Eq(Dt(u(t, x)) + u(t, x) * Dx(u(t, x)) - 0.05 * Dxx(u(t, x)), 0)
    bcs = [npde.Eq(u(0.0, x), -sp.sin(sp.pi * x)), npde.Eq(u(t, -1.0), 0.0), npde.Eq(u(t, 1.0), 0.0)]
    dom = [npde.In(t, npde.Interval(0.0, 1.0)), npde.In(x, npde.Interval(-1.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [t, x], [u(t, x)])
    worst = None
    for seed in (0, 1):                                   # the reference draws from the global RNG: any seed must pass
        rng = np.random.default_rng(seed)
        strategy = npde.QuasiRandomTraining(256, minibatch=32, sampling_alg=npde.LatinHypercubeSample(seed=seed), rng=rng)
        disc = npde.DeepGalerkin(2, 1, 50, 5, "tanh", "tanh", "identity", strategy,
                                 init_params=npde.initialparameters(rng, npde.DGM(2, 1, 50, 5, "tanh", "tanh")), precision="f32")
        prob = npde.discretize(sysm, disc)
        assert prob.pinnrep.engine.L.backend == "hip"
        res = npde.solve(prob, npde.Adam(0.01), maxiters=500)
        res = npde.solve(npde.remake(prob, u0=res.u), npde.Adam(0.001), maxiters=200)
        phi = disc.phi
        tt, xx = np.meshgrid(g["ts"], g["xs"], indexing="ij")
        pred = phi(np.stack([tt.reshape(-1), xx.reshape(-1)]), res.u).reshape(tt.shape)
        err = np.linalg.norm(pred - g["u"]) / max(np.linalg.norm(pred), np.linalg.norm(g["u"]))
        print(f"seed {seed}: final loss {res.losses[-1]:.3e}, relative error vs the MOL table {err:.3f}")
        worst = err if worst is None else max(worst, err)
    assert worst < 0.2
