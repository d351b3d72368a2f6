"""The reference's own END-TO-END acceptance tests, run through the engine on the hardware: same PDE systems, same networks, same point
designs, the reference's own known answers (analytic solutions) and its own tolerances (`isapprox` semantics: 2-norm of the whole
prediction vector unless the test names another norm).  Where the reference finishes with BFGS / LBFGS the mirror runs more Adam
iterations of the resident-theta loop instead (`pinn_adam_steps`; the engine ships no quasi-Newton optimiser).  These are the
known-answer tests the reference holds for the PhysicsInformedNN path; every case cites its file and line.
`PINN_ACCEPT_ON_EMU=1` runs the same statements on the CPU emulation (development aid)."""
import math
import os

import numpy as np
import pytest
import sympy as sp

pytestmark = pytest.mark.gpu


@pytest.fixture()
def lib(npde, request):
    if os.environ.get("PINN_ACCEPT_ON_EMU"):
        emu = request.getfixturevalue("emu_lib")
        npde._lib.set_library(emu)
        yield emu
        npde._lib.set_library(None)
    else:
        yield request.getfixturevalue("hip_lib")


def train(npde, prob, schedule):
    """schedule: [(learning rate, iterations), ...]; every stage is solve(remake(prob, u0 = res.u), Adam(lr))."""
    u, losses = prob.u0, []
    for lr, iters in schedule:
        res = npde.solve(npde.remake(prob, u0=u), npde.Adam(lr), maxiters=iters)
        u = res.u
        losses += list(res.losses)
    assert np.all(np.isfinite(losses))
    return u, losses


def grid2(a, b, step):
    xs = np.arange(a[0], a[1] + 0.5 * step, step)
    ys = np.arange(b[0], b[1] + 0.5 * step, step)
    X, Y = np.meshgrid(xs, ys, indexing="ij")              # `for x in xs for y in ys`
    return np.stack([X.ravel(), Y.ravel()])


def chain_of(npde, n_in, width, hidden, act):
    layers = [npde.Dense(n_in, width, act)] + [npde.Dense(width, width, act) for _ in range(hidden - 1)] + [npde.Dense(width, 1)]
    return npde.Chain(*layers)


@pytest.mark.parametrize("strategy", ["grid", "stochastic", "quasirandom"])
def test_pde_ii_2d_poisson(npde, lib, strategy):
    """test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:58-94 (Chain(Dense(2,12,sigma), Dense(12,12,sigma), Dense(12,1)); Adam(0.01) x 1000 then
    BFGS x 1000; `u_predict ≈ u_real atol = 2.0` on the 101 x 101 grid) for the strategies of its set-up module (:30-50)."""
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dxx, Dyy = npde.Differential(x) ** 2, npde.Differential(y) ** 2
    eq = npde.Eq(Dxx(u(x, y)) + Dyy(u(x, y)), -sp.sin(sp.pi * x) * sp.sin(sp.pi * y))
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(u(1, y), 0.0), npde.Eq(u(x, 0), 0.0), npde.Eq(u(x, 1), 0.0)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    chain = chain_of(npde, 2, 12, 2, "sigmoid")
    strat = {"grid": lambda: npde.GridTraining(0.1),
             "stochastic": lambda: npde.StochasticTraining(100, bcs_points=50, rng=np.random.default_rng(1)),
             "quasirandom": lambda: npde.QuasiRandomTraining(100, bcs_points=50, sampling_alg=npde.LatinHypercubeSample(seed=2))}[strategy]()
    theta0 = npde.initialparameters(np.random.default_rng(100), chain)
    prob = npde.discretize(npde.PDESystem([eq], bcs, dom, [x, y], [u(x, y)]), npde.PhysicsInformedNN(chain, strat, init_params=theta0))
    theta, losses = train(npde, prob, [(0.01, 1000), (0.003, 2000)])
    pts = grid2((0.0, 1.0), (0.0, 1.0), 0.01)
    pred = prob.pinnrep.phi(pts, theta)[0]
    real = np.sin(np.pi * pts[0]) * np.sin(np.pi * pts[1]) / (2 * np.pi ** 2)
    err = np.linalg.norm(pred - real)
    print(f"pde_ii {strategy}: ||u_predict - u_real||_2 = {err:.3f} over {pts.shape[1]} points (reference tolerance 2.0), loss {losses[0]:.3e} -> {losses[-1]:.3e}")
    assert err <= 2.0
    assert losses[-1] < 0.1 * losses[0]


def test_pde_iv_system_of_pdes(npde, lib):
    """test/NNPDE1/nnpde__pde_iv_system_of_pdes.jl:46-106: two networks Dense(2,15,tanh) -> Dense(15,1), QuadratureTraining, Adam(0.01) x 2000;
    max-norm error of each component against the analytic solution <= 0.3."""
    x, y = npde.parameters("x y")
    u1, u2 = npde.variables("u1 u2")
    Dx, Dy = npde.Differential(x), npde.Differential(y)
    eqs = [npde.Eq(Dx(u1(x, y)) + 4 * Dy(u2(x, y)), 0), npde.Eq(Dx(u2(x, y)) + 9 * Dy(u1(x, y)), 0)]
    bcs = [npde.Eq(u1(x, 0), 2 * x), npde.Eq(u2(x, 0), 3 * x)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    chains = [npde.Chain(npde.Dense(2, 15, "tanh"), npde.Dense(15, 1)) for _ in range(2)]
    rng = np.random.default_rng(7)
    theta0 = np.concatenate([npde.initialparameters(rng, c) for c in chains])
    disc = npde.PhysicsInformedNN(chains, npde.QuadratureTraining(), init_params=theta0)
    prob = npde.discretize(npde.PDESystem(eqs, bcs, dom, [x, y], [u1(x, y), u2(x, y)]), disc)
    theta, losses = train(npde, prob, [(0.01, 2000)])
    pts = grid2((0.0, 1.0), (0.0, 1.0), 0.01)
    rep = prob.pinnrep
    real = [(6 * pts[0] - pts[1]) / 3, (6 * pts[0] - pts[1]) / 2]
    for i, name in enumerate(("u1", "u2")):
        pred = rep.phi[i](pts, npde.depvar_params(rep, theta, name))[0]          # phi[i]([x, y], res.u.depvar[depvars[i]])
        err = np.max(np.abs(pred - real[i]))
        print(f"pde_iv {name}: max |u_predict - u_real| = {err:.3f} (reference tolerance 0.3)")
        assert err <= 0.3


def test_pde_v_2d_wave_equation(npde, lib):
    """test/NNPDE1/nnpde__pde_v_2d_wave_equation.jl:60-124: u_tt = u_xx, Dense(2,16,sigma) x 2, QuadratureTraining, Adam(0.01) x 2000 then
    BFGS x 2000; `u_predict ≈ u_real atol = 0.5` on the 11 x 11 grid."""
    x, t = npde.parameters("x t")
    (u,) = npde.variables("u")
    Dxx, Dtt, Dt = npde.Differential(x) ** 2, npde.Differential(t) ** 2, npde.Differential(t)
    eq = npde.Eq(Dtt(u(x, t)), 1 ** 2 * Dxx(u(x, t)))
    bcs = [npde.Eq(u(0, t), 0.0), npde.Eq(u(1, t), 0.0), npde.Eq(u(x, 0), x * (1.0 - x)), npde.Eq(Dt(u(x, 0)), 0.0)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(t, npde.Interval(0.0, 1.0))]
    chain = chain_of(npde, 2, 16, 2, "sigmoid")
    theta0 = npde.initialparameters(np.random.default_rng(3), chain)
    prob = npde.discretize(npde.PDESystem([eq], bcs, dom, [x, t], [u(x, t)]),
                           npde.PhysicsInformedNN(chain, npde.QuadratureTraining(), init_params=theta0))
    theta, losses = train(npde, prob, [(0.01, 2000), (0.003, 4000)])
    pts = grid2((0.0, 1.0), (0.0, 1.0), 0.1)
    k = np.arange(1, 2001, 2)[:, None]
    real = np.sum(8 / (k ** 3 * np.pi ** 3) * np.sin(k * np.pi * pts[0][None, :]) * np.cos(k * np.pi * pts[1][None, :]), axis=0)
    err = np.linalg.norm(prob.pinnrep.phi(pts, theta)[0] - real)
    print(f"pde_v: ||u_predict - u_real||_2 = {err:.3f} over 121 points (reference tolerance 0.5)")
    assert err <= 0.5


def test_pde_vi_mixed_derivative(npde, lib):
    """test/NNPDE1/nnpde__pde_vi_pde_with_mixed_derivative.jl:58-97: u_xx + u_xy - 2 u_yy = -1 with value / derivative boundary conditions,
    Dense(2,32,sigma) x 2, QuasiRandomTraining(2048; SobolSample, resampling = false), BFGS x 500; `u_predict ≈ u_real rtol = 0.1`."""
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dx, Dy = npde.Differential(x), npde.Differential(y)
    eq = npde.Eq((Dx ** 2)(u(x, y)) + Dx(Dy(u(x, y))) - 2 * (Dy ** 2)(u(x, y)), -1.0)
    bcs = [npde.Eq(u(x, 0), x), npde.Eq(Dy(u(x, 0)), x), npde.Eq(u(x, 0), Dy(u(x, 0)))]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    chain = chain_of(npde, 2, 32, 2, "sigmoid")
    theta0 = npde.initialparameters(np.random.default_rng(100), chain)
    strat = npde.QuasiRandomTraining(2048, sampling_alg=npde.SobolSample(seed=1), resampling=False, minibatch=1)
    prob = npde.discretize(npde.PDESystem([eq], bcs, dom, [x, y], [u(x, y)]), npde.PhysicsInformedNN(chain, strat, init_params=theta0))
    theta, losses = train(npde, prob, [(0.01, 3000), (0.003, 3000)])
    pts = grid2((0.0, 1.0), (0.0, 1.0), 0.01)
    real = pts[0] + pts[0] * pts[1] + pts[1] ** 2 / 2
    pred = prob.pinnrep.phi(pts, theta)[0]
    rel = np.linalg.norm(pred - real) / max(np.linalg.norm(pred), np.linalg.norm(real))
    print(f"pde_vi: relative 2-norm error {rel:.4f} (reference tolerance 0.1)")
    assert rel <= 0.1


def test_direct_function_approximation_1d(npde, lib):
    """test/NNPDE2/direct_function__approximation_of_function_1d.jl:15-38: `u(x) ~ 2 + abs(x - 0.5)` with the trivial boundary condition
    `u(0) ~ u(0)`, Dense(1,10,tanh) x 2, GridTraining(0.01), Adam(0.05) x 1000 then BFGS x 500; rtol = 0.02 on 2001 points."""
    (x,) = npde.parameters("x")
    (u,) = npde.variables("u")
    eq = [npde.Eq(u(x), 2 + sp.Abs(x - 0.5))]
    dom = [npde.In(x, npde.Interval(0.0, 2.0))]
    chain = chain_of(npde, 1, 10, 2, "tanh")
    theta0 = npde.initialparameters(np.random.default_rng(110), chain)
    prob = npde.discretize(npde.PDESystem(eq, [npde.Eq(u(0), u(0))], dom, [x], [u(x)]),
                           npde.PhysicsInformedNN(chain, npde.GridTraining(0.01), init_params=theta0))
    theta, losses = train(npde, prob, [(0.05, 1000), (0.01, 2000), (0.003, 2000)])
    xs = np.arange(0.0, 2.0 + 0.0005, 0.001)[None, :]
    real = 2 + np.abs(xs[0] - 0.5)
    pred = prob.pinnrep.phi(xs, theta)[0]
    rel = np.linalg.norm(pred - real) / max(np.linalg.norm(pred), np.linalg.norm(real))
    print(f"direct function 1d: relative 2-norm error {rel:.4f} (reference tolerance 0.02)")
    assert rel <= 0.02


def test_docs_third_order_ode(npde, lib):
    """docs/src/examples/3rd.md:22-52: u''' = cos(pi x), u(0) = 0, u(1) = cos(pi), u'(1) = 1; Chain(Dense(1,8,sigma), Dense(8,1)),
    QuasiRandomTraining(20), Adam(0.01) x 2000.  The page only plots prediction against the analytic solution (no numeric tolerance in
    the reference); asserted here: max-norm error <= 0.05 and a loss three orders of magnitude below its start."""
    (x,) = npde.parameters("x")
    (u,) = npde.variables("u")
    Dx = npde.Differential(x)
    eq = npde.Eq((Dx ** 3)(u(x)), sp.cos(sp.pi * x))
    bcs = [npde.Eq(u(0.0), 0.0), npde.Eq(u(1.0), math.cos(math.pi)), npde.Eq(Dx(u(1.0)), 1.0)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0))]
    chain = npde.Chain(npde.Dense(1, 8, "sigmoid"), npde.Dense(8, 1))
    theta0 = npde.initialparameters(np.random.default_rng(5), chain)
    strat = npde.QuasiRandomTraining(20, sampling_alg=npde.LatinHypercubeSample(seed=4))
    prob = npde.discretize(npde.PDESystem([eq], bcs, dom, [x], [u(x)]), npde.PhysicsInformedNN(chain, strat, init_params=theta0))
    theta, losses = train(npde, prob, [(0.01, 2000), (0.003, 4000)])
    xs = np.arange(0.0, 1.0 + 0.0025, 0.005)[None, :]
    real = (np.pi * xs[0] * (-xs[0] + (np.pi ** 2) * (2 * xs[0] - 3) + 1) - np.sin(np.pi * xs[0])) / (np.pi ** 3)
    err = np.max(np.abs(prob.pinnrep.phi(xs, theta)[0] - real))
    print(f"docs 3rd-order ODE: max |u_predict - u_real| = {err:.4f}, loss {losses[0]:.3e} -> {losses[-1]:.3e}")
    assert err <= 0.05 and losses[-1] < 1e-3 * losses[0]
