"""GPU parity tests proper: HIP path (through the C ABI) vs the float64 oracle on identical collocation sets.
Tolerance (north star): 1e-5 relative — per-term loss |L-L*|/|L*|, gradient norm-wise in L2 and Linf
(SURVEY.md §8c)."""
import numpy as np
import pytest

import helpers
import pinn_oracle as po

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check(npde, wl, weights=None):
    from neuralpde_jl_amd import workloads  # noqa
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    assert rep.engine.L.backend == "hip"
    sets = rep.pde_train_sets + rep.bcs_train_sets
    losses, grad = rep.engine.loss_grad(wl.theta, weights)
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    ref = po.loss_and_grad(prob, wl.theta, sets, weights=weights, mode="stencil")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL, (losses, ref.term_losses)
    assert g2 < TOL and gi < TOL, (g2, gi)
    # determinism: bit-identical on a second call
    l2, gr2 = rep.engine.loss_grad(wl.theta, weights)
    assert np.array_equal(l2, losses) and np.array_equal(gr2, grad)
    return rep, losses, grad, ref


def test_cfg1_poisson1d(npde, hip_lib):
    from neuralpde_jl_amd import workloads
    _check(npde, workloads.cfg1_poisson1d(1024))


def test_cfg2_poisson2d_small(npde, hip_lib):
    from neuralpde_jl_amd import workloads
    _check(npde, workloads.cfg2_poisson2d(points=4096, bcs_points=1000))


def test_cfg2_ragged_and_weights(npde, hip_lib):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=1237, bcs_points=77)
    _check(npde, wl, weights=[1.0, 10.0, 0.5, 2.0, 3.0])


def test_cfg3_burgers_small(npde, hip_lib):
    from neuralpde_jl_amd import workloads
    _check(npde, workloads.cfg3_burgers(points=4096, bcs_points=512))


def test_residual_and_phi(npde, hip_lib):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=333, bcs_points=50)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    pts = rep.pde_train_sets[0]
    r = rep.engine.residual(0, wl.theta, pts.shape[1])
    r_ref = po.residual_values(prob, wl.theta, 0, pts)[0]
    assert np.max(np.abs(r - r_ref)) < 2e-5 * max(1.0, np.max(np.abs(r_ref)))
    u = rep.phi(pts, wl.theta)[0]
    u_ref = po.phi_values(prob.chains[0], wl.theta, pts)[0]
    assert np.max(np.abs(u - u_ref)) < 1e-5


@pytest.mark.parametrize("name", ["cfg1_poisson1d_1024", "cfg2_poisson2d_512", "cfg3_burgers_512"])
def test_golden_fixtures(npde, hip_lib, name):
    """Committed golden vectors (oracle/make_golden.py): inputs AND expected outputs come from the fixture file."""
    import os
    from neuralpde_jl_amd import workloads
    makers = {"cfg1_poisson1d_1024": lambda: workloads.cfg1_poisson1d(1024),
              "cfg2_poisson2d_512": lambda: workloads.cfg2_poisson2d(points=512, bcs_points=128),
              "cfg3_burgers_512": lambda: workloads.cfg3_burgers(points=512, bcs_points=128)}
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    wl = makers[name]()
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    for k in range(int(g["nsets"])):
        rep.engine.set_points(k, g[f"set{k}"])
    losses, grad = rep.engine.loss_grad(g["theta"], g["weights"])
    assert np.max(np.abs(losses - g["losses_stencil"]) / g["losses_stencil"]) < TOL
    assert np.linalg.norm(grad - g["grad_stencil"]) / np.linalg.norm(g["grad_stencil"]) < TOL
    assert np.max(np.abs(grad - g["grad_stencil"])) / np.max(np.abs(g["grad_stencil"])) < TOL
    l64, g64 = rep.engine.loss_grad_f64(g["theta"], g["weights"])          # Float64 boundary (the reference's default eltype)
    assert np.array_equal(l64, losses) and np.array_equal(g64.astype(np.float32), grad)


def test_full_size_cfg2_properties(npde, hip_lib):
    """BASELINE.json config 2 at full size (65,536 interior + 4 x 65,536 boundary points): size-independent properties.
    (a) additivity over point shards with n_norm = global count (what the multi-GPU path relies on),
    (b) linearity of the gradient in the term weights / per-term gradients,
    (c) residual spot check of 512 random points against the oracle, (d) bit-determinism."""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=65536)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    sets = rep.pde_train_sets + rep.bcs_train_sets
    w = np.array([1.0, 2.0, 0.5, 4.0, 3.0], dtype=np.float32)
    L, G = eng.loss_grad(wl.theta, w)
    L2, G2 = eng.loss_grad(wl.theta, w)
    assert np.array_equal(L, L2) and np.array_equal(G, G2)                                      # (d)
    Lt, TG = eng.term_grads(wl.theta)                                                            # (b)
    np.testing.assert_allclose(Lt, L, rtol=1e-12)
    Gw = (w[:, None].astype(np.float64) * TG.astype(np.float64)).sum(axis=0)
    assert np.linalg.norm(Gw - G) / np.linalg.norm(G) < 1e-6
    acc_l, acc_g = np.zeros(5), np.zeros(eng.P)                                                  # (a)
    for part in range(2):
        for k, s in enumerate(sets):
            n = s.shape[1]
            eng.set_points(k, s[:, part * n // 2:(part + 1) * n // 2], n_norm=n)
        l_, g_ = eng.loss_grad(wl.theta, w)
        acc_l += l_
        acc_g += g_
    for k, s in enumerate(sets):
        eng.set_points(k, s)
    np.testing.assert_allclose(acc_l, L, rtol=1e-6)
    assert np.linalg.norm(acc_g - G) / np.linalg.norm(G) < 1e-6
    rng = np.random.default_rng(0)                                                                # (c)
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    for k in (0, 2):
        idx = rng.choice(sets[k].shape[1], 512, replace=False)
        r = eng.residual(k, wl.theta, sets[k].shape[1])[idx]
        r_ref = po.residual_values(prob, wl.theta, k, sets[k][:, idx])[0]
        assert np.max(np.abs(r - r_ref)) < 2e-5 * max(1.0, np.max(np.abs(r_ref)))
    assert abs(L[0] - np.mean(eng.residual(0, wl.theta, 65536).astype(np.float64) ** 2)) < 1e-6 * L[0]


def test_param_estim_4x64(npde, hip_lib):
    import sympy as sp
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    (k,) = npde.parameters("k")
    Dt, Dxx = npde.Differential(t), npde.Differential(x) ** 2
    eq = npde.Eq(Dt(u(t, x)), k * Dxx(u(t, x)))
    bcs = [npde.Eq(u(0, x), sp.sin(sp.pi * x)), npde.Eq(u(t, 0), 0.0), npde.Eq(u(t, 1), 0.0)]
    dom = [npde.In(t, npde.Interval(0.0, 1.0)), npde.In(x, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [t, x], [u(t, x)], ps=[k], defaults={k: 0.7})
    from neuralpde_jl_amd import workloads
    chain = workloads.mlp(2, 64, 4)
    theta = workloads.synthetic_theta([chain], 77)
    strat = npde.QuasiRandomTraining(3000, bcs_points=500, sampling_alg=npde.SobolSample(seed=9), resampling=False, minibatch=1)
    rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=theta, param_estim=True, precision="f32"))
    th = rep.flat_init_params
    sets = rep.pde_train_sets + rep.bcs_train_sets
    losses, grad = rep.engine.loss_grad(th)
    ref = po.loss_and_grad(helpers.oracle_problem(npde, sysm, [chain], param_estim=True), th, sets, mode="stencil")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL
    assert abs(grad[-1] - ref.grad[-1]) < TOL * abs(ref.grad[-1])          # dL/dk itself


@pytest.mark.parametrize("width,hidden", [(64, 4), (128, 5)])
def test_cfg4_cavity_coupled_three_nets(npde, hip_lib, width, hidden):
    """BASELINE config 4 (lid-driven cavity, three coupled networks u, v, p, bc weights 10): 3 x (5x128) as stated in
    BASELINE.json, and 3 x (4x64); reduced point counts so the float64 oracle finishes in seconds."""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg4_cavity(points=3000, bcs_points=400, width=width, hidden=hidden)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    assert "coupled" in rep.engine.describe()
    sets = rep.pde_train_sets + rep.bcs_train_sets
    w = rep._weights
    assert list(w) == [1.0] * 3 + [10.0] * 8
    losses, grad = rep.engine.loss_grad(wl.theta, w)
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    ref = po.loss_and_grad(prob, wl.theta, sets, weights=w, mode="stencil")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)
    l2, gr2 = rep.engine.loss_grad(wl.theta, w)
    assert np.array_equal(l2, losses) and np.array_equal(gr2, grad)


def test_cfg5_heat_inverse_6x128(npde, hip_lib):
    """BASELINE config 5 (3-D heat inverse problem, 6x128 MLP, 4 inputs, 8 jet channels, kappa estimated, stochastic points)
    at a reduced point count; the stochastic sets drawn by the strategy are the ones handed to the oracle."""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg5_heat_inverse(points=4000, bcs_points=500)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    th = rep.flat_init_params
    assert th.size == wl.chains[0].nparams + 1
    sets = rep._state["pde_sets"] + rep._state["bc_sets"]
    for k, sset in enumerate(sets):
        rep.engine.set_points(k, sset)
    losses, grad = rep.engine.loss_grad(th)
    ref = po.loss_and_grad(helpers.oracle_problem(npde, wl.pde_system, wl.chains, param_estim=True), th, sets, mode="stencil")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)
    assert abs(grad[-1] - ref.grad[-1]) < TOL * abs(ref.grad[-1])
    # the strategy redraws on every full_loss_function call (training_strategies.jl:277-281)
    f1 = rep.loss_functions.full_loss_function(th)
    f2 = rep.loss_functions.full_loss_function(th)
    assert np.isfinite(f1) and f1 != f2


def test_training_converges_resident_adam(npde, hip_lib):
    """End-to-end check in the style of the reference's NNPDE1 integration tests (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:
    59-101: train, then compare phi with the analytic solution sin(pi x) sin(pi y) / (2 pi^2)) on the resident-theta Adam
    loop: loss and solution error must drop."""
    import sympy as sp
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dxx, Dyy = npde.Differential(x) ** 2, npde.Differential(y) ** 2
    eq = npde.Eq(Dxx(u(x, y)) + Dyy(u(x, y)), -sp.sin(sp.pi * x) * sp.sin(sp.pi * y))
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(u(1, y), 0.0), npde.Eq(u(x, 0), 0.0), npde.Eq(u(x, 1), 0.0)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [x, y], [u(x, y)])
    chain = npde.Chain(npde.Dense(2, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1))
    th0 = npde.initialparameters(np.random.default_rng(0), chain)
    strat = npde.QuasiRandomTraining(2048, bcs_points=256, sampling_alg=npde.SobolSample(seed=1), resampling=False, minibatch=1)
    prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th0, precision="f32"))
    res = npde.solve(prob, npde.Adam(0.01), maxiters=3000)
    xs = np.linspace(0, 1, 21)
    grid = np.array([[a, b] for a in xs for b in xs]).T
    analytic = np.sin(np.pi * grid[0]) * np.sin(np.pi * grid[1]) / (2 * np.pi ** 2)
    err0 = np.max(np.abs(prob.pinnrep.phi(grid, prob.u0)[0] - analytic))
    err1 = np.max(np.abs(prob.pinnrep.phi(grid, res.u)[0] - analytic))
    print("loss", res.losses[0], "->", res.losses[-1], "max error", err0, "->", err1)
    assert res.losses[-1] < res.losses[0] / 50 and err1 < 0.02 and err1 < err0 / 3


def test_device_samplers_gpu(npde, hip_lib):
    """k_sample_sobol on the GPU == elements 1..n of scipy's un-randomised Sobol' sequence, bit for bit (axes 1-2 through the
    Poisson problem, n = 65,536); Latin-hypercube and uniform redraws stay inside their bounds and stratify."""
    import sympy as sp
    import warnings
    from scipy.stats import qmc
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dxx, Dyy = npde.Differential(x) ** 2, npde.Differential(y) ** 2
    eq = npde.Eq(Dxx(u(x, y)) + Dyy(u(x, y)), -sp.sin(sp.pi * x) * sp.sin(sp.pi * y))
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(u(1, y), 0.0), npde.Eq(u(x, 0), 0.0), npde.Eq(u(x, 1), 0.0)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [x, y], [u(x, y)])
    chain = npde.Chain(npde.Dense(2, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1))
    th0 = npde.initialparameters(np.random.default_rng(0), chain)
    strat = npde.QuasiRandomTraining(65536, bcs_points=1024, sampling_alg=npde.SobolSample(scramble=False))
    prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th0, precision="f32"))
    eng = prob.pinnrep.engine
    assert eng.L.backend == "hip"
    n = 65536
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = qmc.Sobol(2, scramble=False).random(n + 1)[1:].T.astype(np.float32)
    eng.set_sampler(0, [0.0, 0.0], [1.0, 1.0], n, seed=0, kind=3)
    assert np.array_equal(eng.get_points(0, 2, n), ref)
    eng.set_sampler(0, [0.0, 0.0], [1.0, 1.0], n, seed=9, kind=2)          # Latin hypercube
    p = eng.get_points(0, 2, n).astype(np.float64)
    for i in range(2):
        assert sorted(np.floor(p[i] * n).astype(int).clip(0, n - 1)) == list(range(n))
    eng.set_sampler(0, [0.25, -1.0], [0.5, 3.0], n, seed=9, kind=1)         # uniform
    p = eng.get_points(0, 2, n)
    assert p[0].min() >= 0.25 and p[0].max() <= 0.5 and p[1].min() >= -1.0 and p[1].max() <= 3.0
    assert abs(p[1].mean() - 1.0) < 0.05
    res = npde.solve(prob, npde.Adam(0.01), maxiters=20)                     # Sobol design drawn on the device every iteration
    assert np.all(np.isfinite(res.losses)) and res.losses[-1] < res.losses[0]


@pytest.mark.parametrize("width,hidden,d", [(10, 3, 1), (16, 3, 2), (25, 2, 2), (12, 3, 3), (25, 3, 3), (30, 1, 3),
                                            (40, 2, 2), (50, 3, 2), (64, 4, 1), (48, 3, 3), (40, 4, 3)])
def test_small_net_shape_grid_gpu(npde, hip_lib, width, hidden, d):
    """the reference's test-suite net shapes (4..64 wide, 1-4 hidden layers, 1-3 inputs) on the HIP kernels, incl. the full-Hessian jet
    sets of 3-input nets (10 channels): loss and gradient vs the float64 oracle (exact-derivative mode) at 1e-5."""
    sysm, chain = helpers.shape_problem(npde, width, hidden, d)
    strat = npde.QuasiRandomTraining(700, bcs_points=300, sampling_alg=npde.SobolSample(seed=width + hidden), resampling=False, minibatch=1)
    th = po.glorot_theta(po.Chain(tuple(chain.sizes), chain.act), np.random.default_rng(100 + width + 10 * hidden + d))
    rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th, precision="f32"))
    assert rep.engine.L.backend == "hip"
    sets = rep.pde_train_sets + rep.bcs_train_sets
    losses, grad = rep.engine.loss_grad(rep.flat_init_params)
    prob = helpers.oracle_problem(npde, sysm, [chain])
    ref = po.loss_and_grad(prob, rep.flat_init_params, sets, mode="exact")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)
    for _ in range(3):                                   # bit-reproducible
        l2, gr2 = rep.engine.loss_grad(rep.flat_init_params)
        assert np.array_equal(l2, losses) and np.array_equal(gr2, grad)


@pytest.mark.parametrize("which", ["cfg2", "cfg3", "cfg4_64", "cfg4_128", "cfg5"])
def test_bit_reproducibility(npde, hip_lib, which):
    """fixed theta + fixed points => bit-identical losses and gradient on every call (SURVEY 8b "Determinism": BFGS/LBFGS callers need a
    fixed objective): six repeated evaluations per configuration, every kernel family / launch mode (fused, chained slab sets, coupled
    forward-with-records + reverse launches, 4- and 8-wave workgroups)."""
    from neuralpde_jl_amd import workloads
    wl = {"cfg2": lambda: workloads.cfg2_poisson2d(points=8192),
          "cfg3": lambda: workloads.cfg3_burgers(points=6000, bcs_points=3000),
          "cfg4_64": lambda: workloads.cfg4_cavity(points=3000, bcs_points=400, width=64, hidden=4),
          "cfg4_128": lambda: workloads.cfg4_cavity(points=3000, bcs_points=400, width=128, hidden=5),
          "cfg5": lambda: workloads.cfg5_heat_inverse(points=4000, bcs_points=500)}[which]()
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    th = rep.flat_init_params
    l0, g0 = rep.engine.loss_grad(th)
    assert np.all(np.isfinite(g0))
    for _ in range(6):
        l, g = rep.engine.loss_grad(th)
        assert np.array_equal(l, l0) and np.array_equal(g, g0), int(np.sum(g != g0))


@pytest.mark.parametrize("d,width,acts", [(1, 8, ("tanh", "sigmoid")), (2, 12, ("tanh", "sigmoid")), (2, 16, ("sigmoid", "tanh", "sigmoid"))])
def test_per_layer_activations_gpu(npde, hip_lib, d, width, acts):
    """tanh / sigmoid mixed per hidden layer (the reference's Lorenz chains: Dense(1, n, tanh), Dense(n, n, σ), Dense(n, 1)) on the HIP
    kernels' branch-free ACT_MIXED variant, against the oracle; bit-reproducible."""
    sysm, _ = helpers.shape_problem(npde, width, len(acts), d)
    layers = [npde.Dense(d, width, acts[0])] + [npde.Dense(width, width, a) for a in acts[1:]] + [npde.Dense(width, 1)]
    chain = npde.Chain(*layers)
    strat = npde.QuasiRandomTraining(900, bcs_points=300, sampling_alg=npde.SobolSample(seed=width), resampling=False, minibatch=1)
    th = po.glorot_theta(po.Chain(tuple(chain.sizes), chain.act), np.random.default_rng(width))
    rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th, precision="f32"))
    assert rep.engine.L.backend == "hip"
    sets = rep.pde_train_sets + rep.bcs_train_sets
    losses, grad = rep.engine.loss_grad(rep.flat_init_params)
    ref = po.loss_and_grad(helpers.oracle_problem(npde, sysm, [chain]), rep.flat_init_params, sets, mode="exact")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)
    for _ in range(4):
        l2, gr2 = rep.engine.loss_grad(rep.flat_init_params)
        assert np.array_equal(l2, losses) and np.array_equal(gr2, grad)


@pytest.mark.parametrize("width,hidden,act", [(100, 3, "tanh"), (128, 4, "sigmoid")])
def test_wide_nets_3_and_4_hidden_layers_gpu(npde, hip_lib, width, hidden, act):
    """65..128-wide nets with 3 / 4 hidden layers on the HIP kernels (8-wave workgroups): 2-D Poisson vs the oracle, bit-reproducible."""
    import sympy as sp
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dxx, Dyy = npde.Differential(x) ** 2, npde.Differential(y) ** 2
    eq = npde.Eq(Dxx(u(x, y)) + Dyy(u(x, y)), -sp.sin(sp.pi * x) * sp.sin(sp.pi * y))
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(u(1, y), 0.0), npde.Eq(u(x, 0), 0.0), npde.Eq(u(x, 1), 0.0)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [x, y], [u(x, y)])
    layers = [npde.Dense(2, width, act)] + [npde.Dense(width, width, act) for _ in range(hidden - 1)] + [npde.Dense(width, 1)]
    chain = npde.Chain(*layers)
    strat = npde.QuasiRandomTraining(2000, bcs_points=500, sampling_alg=npde.SobolSample(seed=hidden), resampling=False, minibatch=1)
    th = po.glorot_theta(po.Chain(tuple(chain.sizes), chain.act), np.random.default_rng(300 + hidden))
    rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th, precision="f32"))
    sets = rep.pde_train_sets + rep.bcs_train_sets
    losses, grad = rep.engine.loss_grad(rep.flat_init_params)
    ref = po.loss_and_grad(helpers.oracle_problem(npde, sysm, [chain]), rep.flat_init_params, sets, mode="stencil")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)
    for _ in range(4):
        l2, gr2 = rep.engine.loss_grad(rep.flat_init_params)
        assert np.array_equal(l2, losses) and np.array_equal(gr2, grad)


def test_higher_order_derivatives_gpu(npde, hip_lib):
    """pure third / fourth derivative jets on the hardware: the reference's 3rd-order ODE set-up, a 4th-order 1-D problem and the
    Kuramoto-Sivashinsky jet set (family 1 sigmoid 2x12 and family 2 tanh 4x64), against the oracle's exact derivatives
    (the reference's order-3/4 stencils themselves carry ~1e-5 of finite-difference error)."""
    import test_emu_parity as tp
    import sympy as sp

    def run(sysm, chain, strat, seed, weights=None):
        th = tp.theta_for(chain, seed)
        rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th, precision="f32"))
        assert rep.engine.L.backend == "hip"
        sets = rep.pde_train_sets + rep.bcs_train_sets
        losses, grad = rep.engine.loss_grad(th, weights)
        prob = helpers.oracle_problem(npde, sysm, [chain])
        ref = po.loss_and_grad(prob, th, sets, weights=weights, mode="exact")
        le, g2, gi = helpers.rel_errors(losses, grad, ref)
        assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)

    run(tp._third_order_ode(npde), npde.Chain(npde.Dense(1, 8, "sigmoid"), npde.Dense(8, 1)), npde.GridTraining(0.01), 31)
    (x,) = npde.parameters("x")
    (u,) = npde.variables("u")
    D4, D3, D2 = npde.Differential(x) ** 4, npde.Differential(x) ** 3, npde.Differential(x) ** 2
    sys4 = npde.PDESystem([npde.Eq(D4(u(x)) + 0.5 * u(x) * D3(u(x)) - D2(u(x)) ** 2, sp.sin(2 * x))],
                          [npde.Eq(u(0.0), 0.0), npde.Eq(D2(u(1.0)), 0.3)], [npde.In(x, npde.Interval(0.0, 1.0))], [x], [u(x)])
    run(sys4, npde.Chain(npde.Dense(1, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1)), npde.GridTraining(0.002), 32,
        weights=[1.0, 2.0, 0.5])
    ks = tp._ks(npde)
    strat = npde.QuasiRandomTraining(3000, bcs_points=500, sampling_alg=npde.SobolSample(seed=12), resampling=False, minibatch=1)
    run(ks, npde.Chain(npde.Dense(2, 12, "sigmoid"), npde.Dense(12, 12, "sigmoid"), npde.Dense(12, 1)), strat, 41)
    run(ks, npde.Chain(npde.Dense(2, 64, "tanh"), *[npde.Dense(64, 64, "tanh") for _ in range(3)], npde.Dense(64, 1)), strat, 42,
        weights=[1.0, 1.0, 2.0, 2.0, 0.5, 0.5])


def test_heterogeneous_system_gpu(npde, hip_lib):
    """u(x,y,z), v(y,x), h(z), p(x,z) in one system (test/NNPDE1/nnpde__pde_i_heterogeneous_system.jl): per-network input maps."""
    import test_emu_parity as tp
    import sympy as sp
    x, y, z = npde.parameters("x y z")
    u, v, h, p = npde.variables("u v h p")
    Dz = npde.Differential(z)
    eqs = [npde.Eq(u(x, y, z), x + y + z), npde.Eq(v(y, x), x ** 2 + y ** 2), npde.Eq(h(z), sp.cos(z)),
           npde.Eq(p(x, z), sp.exp(x) * sp.exp(z)),
           npde.Eq(u(x, y, z) + v(y, x) * Dz(h(z)) - p(x, z), x + y + z - (x ** 2 + y ** 2) * sp.sin(z) - sp.exp(x) * sp.exp(z))]
    sysm = npde.PDESystem(eqs, [npde.Eq(u(0.0, 0.0, 0.0), 0.0)], [npde.In(s, npde.Interval(0.0, 1.0)) for s in (x, y, z)], [x, y, z],
                          [u(x, y, z), v(y, x), h(z), p(x, z)])
    chains = [npde.Chain(npde.Dense(n, 12, "tanh"), npde.Dense(12, 12, "tanh"), npde.Dense(12, 1)) for n in (3, 2, 1, 2)]
    theta = np.concatenate([tp.theta_for(c, 50 + i) for i, c in enumerate(chains)])
    rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chains, npde.GridTraining(0.1), init_params=theta, precision="f32"))
    assert rep.engine.L.backend == "hip"
    sets = rep.pde_train_sets + rep.bcs_train_sets
    w = [1.0, 2.0, 0.5, 1.5, 3.0, 1.0]
    losses, grad = rep.engine.loss_grad(theta, w)
    ref = po.loss_and_grad(helpers.oracle_problem(npde, sysm, chains), theta, sets, weights=w, mode="stencil")
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)


def test_sin_activation_gpu(npde, hip_lib):
    """sin hidden activations (records keep z): 2-D Poisson on a 4x64 sin net and the small KS set."""
    import test_emu_parity as tp
    sysm, _ = tp.poisson2d(npde, "tanh")
    strat = npde.QuasiRandomTraining(3000, bcs_points=700, sampling_alg=npde.SobolSample(seed=6), resampling=False, minibatch=1)
    for chain, seed, s, mode in ((npde.Chain(npde.Dense(2, 64, "sin"), *[npde.Dense(64, 64, "sin") for _ in range(3)], npde.Dense(64, 1)), 82, sysm, "stencil"),
                                 (npde.Chain(npde.Dense(2, 16, "sin"), npde.Dense(16, 16, "sin"), npde.Dense(16, 1)), 83, tp._ks(npde), "exact")):
        th = tp.theta_for(chain, seed)
        rep = npde.symbolic_discretize(s, npde.PhysicsInformedNN(chain, strat, init_params=th, precision="f32"))
        sets = rep.pde_train_sets + rep.bcs_train_sets
        losses, grad = rep.engine.loss_grad(th)
        ref = po.loss_and_grad(helpers.oracle_problem(npde, s, [chain]), th, sets, mode=mode)
        le, g2, gi = helpers.rel_errors(losses, grad, ref)
        assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)


def test_bench_two_ranks_share_the_gpu(hip_lib):
    """bench.py's N > 1 path (point sharding with n_norm = global N, all-reduce of [gradient | sums]) with two ranks folded onto
    the one visible GPU and the gloo backend: per-term losses must equal the single-rank run's (RCCL itself needs a multi-GPU node)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--points", "8192"],
                         cwd=root, capture_output=True, text=True, timeout=300)
    env = dict(os.environ, PINN_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "8192"],
                         cwd=root, capture_output=True, text=True, timeout=300, env=env)
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    l2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert l2["n_gpus"] == 2 and l2["config"]["parallelism"] == "point-shard x2"
    np.testing.assert_allclose(l2["loss_terms"], l1["loss_terms"], rtol=1e-6)
