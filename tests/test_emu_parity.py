"""CPU parity tests: the REAL kernel + host sources, compiled by g++ as a lock-step 64-lane emulation
(tests/emu/libpinn_emu.so, test infrastructure), driven through the same C ABI and Python host code as the product,
checked against the float64 oracle.  Tolerance: the north star's 1e-5 relative (per-term loss; gradient L2 and Linf)."""
import numpy as np
import pytest
import sympy as sp

import helpers
import pinn_oracle as po

TOL = 1e-5
EXPECTED_BACKEND = "emu"      # tests/test_gpu_mirror.py re-runs this module's tests on the hardware with "hip" here


def check(npde, sysm, chains, strat, theta, weights=None, param_estim=False, tol=TOL, mode="stencil"):
    disc = npde.PhysicsInformedNN(chains if len(chains) > 1 else chains[0], strat, init_params=theta,
                                  param_estim=param_estim, precision="f32")
    rep = npde.symbolic_discretize(sysm, disc)
    assert rep.engine.L.backend == EXPECTED_BACKEND
    sets = rep.pde_train_sets + rep.bcs_train_sets
    th = rep.flat_init_params
    losses, grad = rep.engine.loss_grad(th, weights)
    prob = helpers.oracle_problem(npde, sysm, chains, param_estim=param_estim)
    ref = po.loss_and_grad(prob, th, sets, weights=weights, mode=mode)
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < tol and g2 < tol and gi < tol, (le, g2, gi)
    l2, g2_ = rep.engine.loss_grad(th, weights)
    assert np.array_equal(l2, losses) and np.array_equal(g2_, grad)       # deterministic
    return rep, prob, sets, th


def poisson2d(npde, act="tanh", width=16, hidden=2):
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dxx, Dyy = npde.Differential(x) ** 2, npde.Differential(y) ** 2
    eq = npde.Eq(Dxx(u(x, y)) + Dyy(u(x, y)), -sp.sin(sp.pi * x) * sp.sin(sp.pi * y))
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(u(1, y), 0.0), npde.Eq(u(x, 0), 0.0), npde.Eq(u(x, 1), 0.0)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    layers = [npde.Dense(2, width, act)] + [npde.Dense(width, width, act) for _ in range(hidden - 1)] + [npde.Dense(width, 1)]
    return npde.PDESystem([eq], bcs, dom, [x, y], [u(x, y)]), npde.Chain(*layers)


def theta_for(chain, seed):
    return po.glorot_theta(po.Chain(tuple(chain.sizes), chain.act), np.random.default_rng(seed))


def periodic_heat(npde, inner=16, hidden=2, act="sigmoid"):
    """test/CUDA/nnpde_cuda__1d_pde_dirichlet_bc_cuda.jl:26-48: Dt(u) ~ Dxx(u) on [0, 1] x [0, 2 pi] behind
    Chain(PeriodicEmbedding([2], [2 pi]), Dense(3, inner, sigma), ..., Dense(inner, 1))."""
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    Dt, Dxx = npde.Differential(t), npde.Differential(x) ** 2
    eq = npde.Eq(Dt(u(t, x)), Dxx(u(t, x)))
    bcs = [npde.Eq(u(0, x), sp.cos(x)), npde.Eq(u(t, 0), sp.exp(-t)), npde.Eq(u(t, 2 * sp.pi), sp.exp(-t))]
    dom = [npde.In(t, npde.Interval(0.0, 1.0)), npde.In(x, npde.Interval(0.0, 2 * np.pi))]
    layers = [npde.PeriodicEmbedding([2], [2 * np.pi]), npde.Dense(3, inner, act)] + [npde.Dense(inner, inner, act) for _ in range(hidden - 1)] + [npde.Dense(inner, 1)]
    return npde.PDESystem([eq], bcs, dom, [t, x], [u(t, x)]), npde.Chain(*layers)


def test_periodic_embedding_heat_equation(npde, use_emu):
    """PeriodicEmbedding in front of the Dense stack: the engine rewrites the term over the features (t, sin x, cos x) by the chain rule
    (csrc/descriptor.cpp: apply_embeddings); losses, gradient, residual and phi against the oracle, which embeds inside the chain and
    differentiates with respect to (t, x) directly."""
    sysm, chain = periodic_heat(npde)
    assert chain.sizes[0] == 3 and chain.n_inputs == 2 and chain.embed == ((1, 2 * np.pi),)
    strat = npde.QuasiRandomTraining(60, bcs_points=23, sampling_alg=npde.SobolSample(seed=5), resampling=False, minibatch=1)
    rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, 11), weights=[1.0, 2.0, 0.5, 1.5], mode="exact", tol=2e-5)
    assert "embed 0 1 1 " in rep.ir.to_descriptor() and all(s.shape[0] == 2 for s in sets)
    r = rep.loss_functions.datafree_pde_loss_functions[0](sets[0], th)
    np.testing.assert_allclose(r, po.residual_values(prob, th, 0, sets[0], mode="exact"), atol=3e-5)
    np.testing.assert_allclose(rep.phi(sets[0], th), po.phi_values(prob.chains[0], th, sets[0]), atol=2e-6)
    # u is 2 pi - periodic in x by construction
    pts = np.array([[0.3, 0.7], [0.4, 5.0]])
    np.testing.assert_allclose(rep.phi(pts, th), rep.phi(pts + np.array([[0.0], [2 * np.pi]]), th), atol=2e-6)
    # the installed point sets come back in the caller's coordinates
    np.testing.assert_allclose(rep.engine.get_points(0, 2, sets[0].shape[1]), sets[0], atol=1e-6)
    # the same problem through the s-expression front end (the form the Julia glue emits)
    eng2 = npde._lib.Engine(rep.ir.to_descriptor2())
    for k, sset in enumerate(sets):
        eng2.set_points(k, sset)
    l1, g1 = rep.engine.loss_grad(th)
    l2, g2 = eng2.loss_grad(th)
    np.testing.assert_allclose(l2, l1, rtol=1e-6)
    np.testing.assert_allclose(g2, g1, rtol=0, atol=2e-6 * np.abs(g1).max())
    # device samplers draw in the caller's coordinates; the embedding rows follow every draw
    eng2.set_sampler(0, [0.0, 0.0], [1.0, 2 * np.pi], 50, seed=7, kind=1)
    drawn = eng2.get_points(0, 2, 50)
    assert drawn.shape == (2, 50) and drawn[1].max() <= 2 * np.pi and drawn[1].max() > 1.0
    ls, gs = eng2.loss_grad(th)
    ref = po.loss_and_grad(prob, th, [drawn.astype(np.float64)] + [s for s in sets[1:]], mode="exact")
    le, e2, ei = helpers.rel_errors(ls, gs, ref)
    assert le.max() < 2e-5 and e2 < 2e-5 and ei < 2e-5, (le, e2, ei)


def test_periodic_embedding_reference_shape_and_limits(npde, use_emu):
    """the reference test's own chain (6 x 30 sigmoid behind the embedding) on a handful of points; what the rewrite refuses"""
    sysm, chain = periodic_heat(npde, inner=30, hidden=6)
    strat = npde.QuasiRandomTraining(24, bcs_points=9, sampling_alg=npde.SobolSample(seed=2), resampling=False, minibatch=1)
    check(npde, sysm, [chain], strat, theta_for(chain, 3), mode="exact", tol=2e-5)
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    eq3 = npde.Eq(npde.Differential(t)(u(t, x)), (npde.Differential(x) ** 3)(u(t, x)))
    sys3 = npde.PDESystem([eq3], sysm.bcs, sysm.domain, [t, x], [u(t, x)])
    with pytest.raises(RuntimeError, match="order > 2 in a periodically embedded coordinate"):
        npde.symbolic_discretize(sys3, npde.PhysicsInformedNN(chain, strat, init_params=theta_for(chain, 3), precision="f32"))
    with pytest.raises(ValueError, match="first layer"):
        npde.Chain(npde.Dense(2, 8, "tanh"), npde.PeriodicEmbedding([1], [1.0]), npde.Dense(8, 1))
    with pytest.raises(ValueError, match="DimensionMismatch"):
        npde.Chain(npde.PeriodicEmbedding([3], [1.0]), npde.Dense(3, 8, "tanh"), npde.Dense(8, 1))


def test_small_poisson_tanh_and_sigmoid(npde, use_emu):
    for act, seed in (("tanh", 1), ("sigmoid", 2)):
        sysm, chain = poisson2d(npde, act)
        strat = npde.QuasiRandomTraining(70, bcs_points=37, sampling_alg=npde.SobolSample(seed=3), resampling=False, minibatch=1)
        check(npde, sysm, [chain], strat, theta_for(chain, seed), weights=[1.0, 3.0, 0.5, 2.0, 1.5])


def test_two_stage_and_one_stage_reductions_agree(npde, use_emu, monkeypatch):
    """small launches (every group <= 32 workgroups) sum the slabs in one kernel, large ones in two stages: same numbers from both."""
    sysm, chain = poisson2d(npde, "tanh")
    strat = npde.QuasiRandomTraining(70, bcs_points=300, sampling_alg=npde.SobolSample(seed=3), resampling=False, minibatch=1)
    rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, 1), weights=[1.0, 3.0, 0.5, 2.0, 1.5])
    out = []
    for lim in ("0", "1000"):                      # (the limit is read at every evaluation)
        monkeypatch.setenv("PINN_REDUCE_DIRECT_MAX", lim)
        out.append(rep.engine.loss_grad(th, [1.0, 3.0, 0.5, 2.0, 1.5]))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-12)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=0, atol=1e-6 * np.abs(out[1][1]).max())
    monkeypatch.delenv("PINN_REDUCE_DIRECT_MAX", raising=False)


def test_padded_width_12(npde, use_emu):
    # the reference's own test net (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:97): 2 -> 12 -> 12 -> 1 sigmoid; padded to 16
    sysm, chain = poisson2d(npde, "sigmoid", width=12)
    check(npde, sysm, [chain], npde.GridTraining(0.2), theta_for(chain, 5))


def test_cfg1_grid_3x32(npde, use_emu):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg1_poisson1d(64)
    rep, prob, sets, th = check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta)
    # residual (datafree loss function) and trial function against the oracle
    r = rep.loss_functions.datafree_pde_loss_functions[0](sets[0], th)
    np.testing.assert_allclose(r, po.residual_values(prob, th, 0, sets[0]), atol=3e-5)
    u = rep.phi(sets[0], th)
    np.testing.assert_allclose(u, po.phi_values(prob.chains[0], th, sets[0]), atol=1e-6)
    assert abs(rep.phi(np.array([0.3]), th)[0] - po.phi_values(prob.chains[0], th, np.array([[0.3]]))[0, 0]) < 1e-6
    # the two one-point boundary terms ride on the interior term's launch (descriptor `hint` lines: sizes of the sets about to be installed)
    groups = [l for l in rep.engine.describe().splitlines() if l.startswith("group")]
    assert len(groups) == 1 and "terms=0,1,2," in groups[0], groups
    losses, tg = rep.engine.term_grads(th)
    ref = po.loss_and_grad(prob, th, sets, mode="stencil", per_term_grads=True)
    assert np.max(np.abs(tg - ref.term_grads)) / np.max(np.abs(ref.term_grads)) < TOL
    # the same problem without hints (a caller that does not send them): one launch group per channel set, same numbers
    import os
    eng2 = npde._lib.Engine(rep.ir.to_descriptor2() if os.environ.get("PINN_DESCRIPTOR") == "2" else rep.ir.to_descriptor()) if hasattr(rep, "ir") else None
    if eng2 is not None:
        assert len([l for l in eng2.describe().splitlines() if l.startswith("group")]) == 2
        for k, sset in enumerate(sets):
            eng2.set_points(k, sset)
        l2, g2 = eng2.loss_grad(th)
        l1, g1 = rep.engine.loss_grad(th)
        np.testing.assert_allclose(l2, l1, rtol=1e-6)
        np.testing.assert_allclose(g2, g1, rtol=0, atol=2e-6 * np.abs(g1).max())


def test_cfg2_4x64_small_ragged(npde, use_emu):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=45, bcs_points=70)      # ragged tiles, dummy tiles, 4x64 cooperative kernels
    check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta)


def test_chained_launch_groups_share_one_slab_set(npde, use_emu):
    """interior + boundary launch groups of one 4x64 network: the boundary group's workgroups add onto the interior group's slabs
    (one reduction input); per-term gradients (head group not launched) fall back to the group's own slabs; both vs the oracle."""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=200, bcs_points=70)     # 13 interior tiles / 8 boundary tiles (4 emulated workgroup slots; more than 64
                                                                 # points per boundary term: small hinted terms would ride on the interior launch)
    w = np.array([1.0, 2.0, 0.5, 3.0, 1.5])
    rep, _, _, th = check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta, weights=list(w))
    assert "chained onto group 0" in rep.engine.describe()
    losses, grad = rep.engine.loss_grad(th, list(w))
    tl, tg = rep.engine.term_grads(th)                    # one evaluation per term: groups run alone
    np.testing.assert_allclose((w[:, None] * tg).sum(axis=0), grad, rtol=0, atol=2e-6 * np.abs(grad).max())
    np.testing.assert_allclose(tl, losses, rtol=1e-6)
    # more boundary workgroups than interior ones: not chained, same answers
    wl2 = workloads.cfg2_poisson2d(points=20, bcs_points=70)
    rep2, *_ = check(npde, wl2.pde_system, wl2.chains, wl2.strategy, wl2.theta)
    assert "own (more workgroups than group 0)" in rep2.engine.describe()


@pytest.mark.parametrize("width,hidden,d", [(10, 3, 1), (7, 1, 1), (16, 3, 2), (25, 2, 2), (32, 1, 2), (20, 3, 2), (8, 1, 3), (12, 3, 3),
                                            (25, 3, 3), (18, 2, 3), (30, 1, 3), (25, 1, 1), (18, 2, 1),
                                            (40, 2, 2), (50, 3, 2), (40, 2, 1), (64, 4, 1), (48, 3, 3), (33, 2, 3), (40, 4, 3)])
def test_small_net_shape_grid(npde, use_emu, width, hidden, d):
    """the widths / depths / input counts of the reference's own test nets (4..32 wide, 1-3 hidden layers, 1-3 inputs): value-only and
    full-Hessian kernels exist for every such shape (inst_h16_*.hip, inst_h32_*.hip), and for widths 33..64 with 2-4 hidden layers on
    the neuron-split kernels (inst2_h64_*.hip); residuals with mixed second derivatives."""
    sysm, chain = helpers.shape_problem(npde, width, hidden, d)
    strat = npde.QuasiRandomTraining(40, bcs_points=20, sampling_alg=npde.SobolSample(seed=width + hidden), resampling=False, minibatch=1)
    # exact-derivative oracle mode: with mixed third-order-total stencils (D_t D_x) the FD oracle's own truncation error reaches 5e-5
    # on some of these random nets, the engine's derivatives are exact
    check(npde, sysm, [chain], strat, theta_for(chain, 100 + width + 10 * hidden + d), mode="exact")


@pytest.mark.parametrize("width,hidden,act", [(100, 3, "tanh"), (128, 4, "sigmoid")])
def test_wide_nets_3_and_4_hidden_layers(npde, use_emu, width, hidden, act):
    """65..128-wide nets with 3 / 4 hidden layers (inst2_h128_mid_d2.hip): 2-D Poisson, 8-wave workgroups, slab-resident dW."""
    sysm, chain = poisson2d(npde, act, width=width, hidden=hidden)
    strat = npde.QuasiRandomTraining(40, bcs_points=70, sampling_alg=npde.SobolSample(seed=hidden), resampling=False, minibatch=1)
    check(npde, sysm, [chain], strat, theta_for(chain, 200 + hidden))


def test_cfg3_burgers_4x64_small(npde, use_emu):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg3_burgers(points=40, bcs_points=30)
    check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta, weights=[1.0, 2.0, 2.0, 2.0])


def test_per_term_gradients(npde, use_emu):
    sysm, chain = poisson2d(npde)
    strat = npde.QuasiRandomTraining(33, bcs_points=20, sampling_alg=npde.SobolSample(seed=4), resampling=False, minibatch=1)
    rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, 7))
    losses, tg = rep.engine.term_grads(th)
    ref = po.loss_and_grad(prob, th, sets, mode="stencil", per_term_grads=True)
    assert np.max(np.abs(tg - ref.term_grads)) / np.max(np.abs(ref.term_grads)) < TOL
    np.testing.assert_allclose(losses, ref.term_losses, rtol=TOL)


def test_param_estim_gradient(npde, use_emu):
    # inverse problem: Dt(u) ~ k * Dxx(u) with k estimated (theta.p, src/discretize.jl:83-95) — gradient incl. dL/dk
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    (k,) = npde.parameters("k")
    Dt, Dxx = npde.Differential(t), npde.Differential(x) ** 2
    eq = npde.Eq(Dt(u(t, x)), k * Dxx(u(t, x)))
    bcs = [npde.Eq(u(0, x), sp.sin(sp.pi * x)), npde.Eq(u(t, 0), 0.0), npde.Eq(u(t, 1), 0.0)]
    dom = [npde.In(t, npde.Interval(0.0, 1.0)), npde.In(x, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [t, x], [u(t, x)], ps=[k], defaults={k: 0.7})
    chain = npde.Chain(npde.Dense(2, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1))
    strat = npde.QuasiRandomTraining(50, bcs_points=20, sampling_alg=npde.SobolSample(seed=9), resampling=False, minibatch=1)
    rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, 11), param_estim=True)
    assert th.size == chain.nparams + 1 and th[-1] == 0.7
    # fixed (non-estimated) parameter: value comes from default_p, no gradient entry
    check(npde, sysm, [chain], strat, theta_for(chain, 11), param_estim=False)


def test_high_level_api_and_resampling(npde, use_emu):
    sysm, chain = poisson2d(npde)
    th0 = theta_for(chain, 21)
    disc = npde.PhysicsInformedNN(chain, npde.StochasticTraining(40, bcs_points=10, rng=np.random.default_rng(1)), init_params=th0,
                                  adaptive_loss=npde.NonAdaptiveLoss(pde_loss_weights=2.0, bc_loss_weights=[1, 2, 3, 4]), precision="f32")
    prob = npde.discretize(sysm, disc)
    assert prob.u0.dtype == np.float64 and prob.u0.size == chain.nparams
    f1, f2 = prob.f(prob.u0), prob.f(prob.u0)
    assert np.isfinite(f1) and f1 != f2                 # StochasticTraining redraws on every call (training_strategies.jl:277-281)
    val, g = prob.f.value_and_grad(prob.u0)
    assert np.isfinite(val) and g.shape == prob.u0.shape and g.dtype == np.float64
    assert disc.iteration[0] == 4                       # self-incremented by every evaluation (discretize.jl:574-576)
    # a few Adam steps on a fixed grid lower the loss (the reference's own convergence-style check, loosely)
    disc = npde.PhysicsInformedNN(chain, npde.GridTraining(0.25), init_params=th0, precision="f32")
    prob = npde.discretize(sysm, disc)
    th, m_, v_ = prob.u0.copy(), 0.0, 0.0
    l0 = prob.f(th)
    for it in range(1, 31):
        val, g = prob.f.value_and_grad(th)
        m_ = 0.9 * m_ + 0.1 * g; v_ = 0.999 * v_ + 0.001 * g * g
        th -= 0.01 * (m_ / (1 - 0.9 ** it)) / (np.sqrt(v_ / (1 - 0.999 ** it)) + 1e-8)
    assert prob.f(th) < 0.7 * l0
    rep = prob.pinnrep
    assert len(rep.loss_functions.pde_loss_functions) == 1 and len(rep.loss_functions.bc_loss_functions) == 4
    total = rep.loss_functions.pde_loss_functions[0](th) + sum(f(th) for f in rep.loss_functions.bc_loss_functions)
    assert abs(total - prob.f(th)) < 1e-6 * max(1.0, abs(total))


def test_golden_fixture_cfg1_through_engine(npde, use_emu):
    import os
    from neuralpde_jl_amd import workloads
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_poisson1d_1024.npz"))
    wl = workloads.cfg1_poisson1d(1024)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    sets = rep.pde_train_sets + rep.bcs_train_sets
    for k in range(int(g["nsets"])):
        np.testing.assert_allclose(sets[k], g[f"set{k}"], rtol=0, atol=1e-15)
    losses, grad = rep.engine.loss_grad(g["theta"], g["weights"])
    assert np.max(np.abs(losses - g["losses_stencil"]) / g["losses_stencil"]) < TOL
    assert np.linalg.norm(grad - g["grad_stencil"]) / np.linalg.norm(g["grad_stencil"]) < TOL


def test_coupled_system_of_pdes(npde, use_emu):
    """Equations coupling several networks (src/discretize.jl:58-80): the reference's own system test
    (test/NNPDE1/nnpde__pde_iv_system_of_pdes.jl:55-86: two chains Dense(2,15,tanh) -> Dense(15,1)), plus a nonlinear
    coupling with second derivatives on deeper nets.  Forward launch per network -> k_expr -> reverse launch per network."""
    x, y = npde.parameters("x y")
    u1, u2 = npde.variables("u1 u2")
    Dx, Dy = npde.Differential(x), npde.Differential(y)
    eqs = [npde.Eq(Dx(u1(x, y)) + 4 * Dy(u2(x, y)), 0), npde.Eq(Dx(u2(x, y)) + 9 * Dy(u1(x, y)), 0)]
    bcs = [npde.Eq(u1(x, 0), 2 * x), npde.Eq(u2(x, 0), 3 * x)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem(eqs, bcs, dom, [x, y], [u1(x, y), u2(x, y)])
    chains = [npde.Chain(npde.Dense(2, 15, "tanh"), npde.Dense(15, 1)) for _ in range(2)]
    theta = np.concatenate([theta_for(c, 31 + i) for i, c in enumerate(chains)])
    strat = npde.QuasiRandomTraining(70, bcs_points=20, sampling_alg=npde.SobolSample(seed=5), resampling=False, minibatch=1)
    rep, prob, sets, th = check(npde, sysm, chains, strat, theta, weights=[1.0, 2.0, 3.0, 0.5])
    assert "coupled" in rep.engine.describe()
    r = rep.loss_functions.datafree_pde_loss_functions[1](sets[1], th)
    np.testing.assert_allclose(r, po.residual_values(prob, th, 1, sets[1]), atol=3e-5)
    losses, tg = rep.engine.term_grads(th)
    ref = po.loss_and_grad(prob, th, sets, mode="stencil", per_term_grads=True)
    assert np.max(np.abs(tg - ref.term_grads)) / np.max(np.abs(ref.term_grads)) < TOL
    # nonlinear coupling, second derivatives, deeper nets, a PDE parameter estimated through the coupled path
    (nu,) = npde.parameters("nu")
    Dxx, Dyy = Dx ** 2, Dy ** 2
    eqs = [npde.Eq(u1(x, y) * Dx(u1(x, y)) + u2(x, y) * Dy(u1(x, y)), nu * (Dxx(u1(x, y)) + Dyy(u1(x, y)))),
           npde.Eq(Dx(u1(x, y)) + Dy(u2(x, y)), sp.sin(sp.pi * x) * u2(x, y))]
    bcs = [npde.Eq(u1(x, 1), 1.0), npde.Eq(u2(0, y), 0.0), npde.Eq(u1(0, y), 0.0)]
    sysm = npde.PDESystem(eqs, bcs, dom, [x, y], [u1(x, y), u2(x, y)], ps=[nu], defaults={nu: 0.05})
    chains = [npde.Chain(npde.Dense(2, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1)) for _ in range(2)]
    theta = np.concatenate([theta_for(c, 41 + i) for i, c in enumerate(chains)])
    check(npde, sysm, chains, strat, theta, param_estim=True)


def test_one_first_derivative_kernels_5x128(npde, use_emu):
    """{u, u_x} / {u, u_y} kernels of the 5 x 128 nets (8-wave workgroups, two point groups per tile): alone (fused launch) and as what a
    coupled system reads from its networks — every (equation, network) pair of a coupled system runs the kernel of ITS OWN channel
    set, so the cavity problem's continuity equation reads {u, u_x} and {v, v_y}, its momentum equations {p, p_x} / {p, p_y} and a bare
    value of the other velocity.  (On the hardware this shape once stored a hidden layer's record from registers the next instruction
    rewrote: gradient off by 1e-3 with all losses right.  vec.hpp: store_pad.)"""
    from neuralpde_jl_amd import workloads
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    chain = workloads.mlp(2, 128, 5)
    strat = npde.QuasiRandomTraining(100, bcs_points=20, sampling_alg=npde.SobolSample(seed=9), resampling=False, minibatch=1)
    for D, kern in ((npde.Differential(x), "D2_F1_"), (npde.Differential(y), "D2_F2_")):
        sysm = npde.PDESystem([npde.Eq(D(u(x, y)), 0.5 * x - y)], [npde.Eq(u(0, y), 0.0)], dom, [x, y], [u(x, y)])
        rep, *_ = check(npde, sysm, [chain], strat, theta_for(chain, 77))
        assert kern in rep.engine.describe() and "(C=2)" in rep.engine.describe()
    wl = workloads.cfg4_cavity(points=100, bcs_points=20)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    desc = rep.engine.describe()
    assert desc.count("coupled") == 8 and desc.count("(C=2)") == 4 and desc.count("(C=4)") == 2, desc
    th = rep.flat_init_params
    sets = rep.pde_train_sets + rep.bcs_train_sets
    w = [1.0] * 3 + [10.0] * 8
    losses, tg = rep.engine.term_grads(th)
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    ref = po.loss_and_grad(prob, th, sets, mode="stencil", per_term_grads=True)
    assert np.max(np.abs(losses - ref.term_losses) / np.abs(ref.term_losses)) < TOL
    for k in range(len(sets)):
        assert np.max(np.abs(tg[k] - ref.term_grads[k])) / np.max(np.abs(ref.term_grads[k])) < TOL, k
    l2, grad = rep.engine.loss_grad(th, w)
    gref = (np.asarray(w)[:, None] * ref.term_grads).sum(0)
    assert np.linalg.norm(grad - gref) / np.linalg.norm(gref) < TOL


def test_coupled_tail_launch_matches_expr_path(npde, use_emu, monkeypatch):
    """Every equation of the cavity system has a TAIL network (the widest channel set of the equation) whose kernel runs forward pass +
    the equation's tape (the other networks' jets as source rows) + reverse sweep in one launch and hands the other networks their
    seeds: no k_expr launch, and the tail network's records stay out of HBM.  PINN_NO_TAIL_FUSE=1 restores forward / k_expr / reverse
    launches for every network.  Same losses (per-point residuals are the same program), gradients equal to rounding, per-term
    gradients, the loss-only evaluation, pinn_residual, and both GEMM modes."""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg4_cavity(points=100, bcs_points=20)
    w = [1.0] * 3 + [10.0] * 8
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    desc = rep.engine.describe()
    assert desc.count("coupled tail") == 3 and desc.count("coupled fwd/gradin") == 5, desc
    th = rep.flat_init_params
    sets = rep.pde_train_sets + rep.bcs_train_sets
    l_t, g_t = rep.engine.loss_grad(th, w)
    monkeypatch.setenv("PINN_NO_TAIL_FUSE", "1")
    rep2 = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    monkeypatch.delenv("PINN_NO_TAIL_FUSE")
    assert "coupled tail" not in rep2.engine.describe() and rep2.engine.describe().count("coupled fwd/gradin") == 8
    for k, sset in enumerate(sets):
        rep2.engine.set_points(k, sset)
    l_e, g_e = rep2.engine.loss_grad(th, w)
    np.testing.assert_allclose(l_t, l_e, rtol=2e-6)
    assert np.linalg.norm(g_t - g_e) / np.linalg.norm(g_e) < 2e-6
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    ref = po.loss_and_grad(prob, th, sets, weights=w, mode="stencil")
    le, g2, gi = helpers.rel_errors(l_t, g_t, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)
    l_lo, _ = rep.engine.loss_grad(th, w, want_grad=False)
    np.testing.assert_allclose(l_lo, l_t, rtol=1e-12)
    tl, tg = rep.engine.term_grads(th)
    np.testing.assert_allclose(tl, l_t, rtol=1e-12)
    np.testing.assert_allclose((np.asarray(w)[:, None] * tg).sum(axis=0), g_t, rtol=0, atol=2e-6 * np.abs(g_t).max())
    for k in range(3):
        r_t = rep.engine.residual(k, th, sets[k].shape[1])
        r_e = rep2.engine.residual(k, th, sets[k].shape[1])
        np.testing.assert_allclose(r_t, r_e, rtol=0, atol=1e-6 * np.abs(r_e).max())
    rep.engine.set_option("gemm", "fp32")
    assert rep.engine.describe().count("coupled tail") == 3
    l_f, g_f = rep.engine.loss_grad(th, w)
    le, g2, gi = helpers.rel_errors(l_f, g_f, ref)
    assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)


def test_wide_nets_family2(npde, use_emu):
    """Neuron-split kernel family: 4x64 (register-resident dW), 2x128 with 5 jet channels, 5x128 (slab-resident dW) and the
    4-D config-5 shape (8 jet channels, chunked dW staging, estimated PDE parameter) at 2x128."""
    from neuralpde_jl_amd import workloads
    for wl in (workloads.cfg2_poisson2d(points=40, bcs_points=70, width=128, hidden=2),
               workloads.cfg2_poisson2d(points=20, bcs_points=70, width=100, hidden=5),
               workloads.cfg5_heat_inverse(points=40, bcs_points=70, width=128, hidden=2)):
        rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
        assert "F2_HP128" in rep.engine.describe()
        th = rep.flat_init_params
        sets = rep._state["pde_sets"] + rep._state["bc_sets"]
        for k, sset in enumerate(sets):
            rep.engine.set_points(k, sset)
        losses, grad = rep.engine.loss_grad(th)
        prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains, param_estim=wl.param_estim)
        ref = po.loss_and_grad(prob, th, sets, mode="stencil")
        le, g2, gi = helpers.rel_errors(losses, grad, ref)
        assert le.max() < TOL and g2 < TOL and gi < TOL, (wl.name, le, g2, gi)


def test_resident_adam_matches_host_adam_and_sampler(npde, use_emu):
    """Resident-theta Adam (pinn_adam_steps) == a host Adam loop over the engine's own gradients ([3P] Optimisers.Adam
    update rule); the on-device uniform sampler stays inside the StochasticTraining bounds and changes every step."""
    sysm, chain = poisson2d(npde)
    th0 = theta_for(chain, 51)
    disc = npde.PhysicsInformedNN(chain, npde.GridTraining(0.25), init_params=th0,
                                  adaptive_loss=npde.NonAdaptiveLoss(pde_loss_weights=1.0, bc_loss_weights=[2.0, 1.0, 3.0, 1.0]), precision="f32")
    prob = npde.discretize(sysm, disc)
    res = npde.solve(prob, npde.Adam(0.01), maxiters=25)
    th, m_, v_, hist = prob.u0.astype(np.float32).copy(), 0.0, 0.0, []
    for it in range(1, 26):
        val, g = prob.f.value_and_grad(th)
        hist.append(val)
        g = g.astype(np.float32)
        m_ = np.float32(0.9) * m_ + np.float32(0.1) * g
        v_ = np.float32(0.999) * v_ + np.float32(0.001) * g * g
        th = (th - np.float32(0.01) * (m_ / np.float32(1 - 0.9 ** it)) / (np.sqrt(v_ / np.float32(1 - 0.999 ** it)) + np.float32(1e-8))).astype(np.float32)
    np.testing.assert_allclose(res.losses, hist, rtol=2e-5)
    assert np.max(np.abs(res.u - th)) < 5e-5
    assert res.losses[-1] < res.losses[0]
    # resume idiom of the reference: prob = remake(prob, u0 = res.u); solve again (nnpde__pde_ii_2d_poisson.jl:84-85)
    res_b = npde.solve(npde.remake(prob, u0=res.u), npde.Adam(0.01), maxiters=5)
    assert len(res_b.losses) == 5 and res_b.losses[-1] < res.losses[0]
    # callback protocol: stop after the first chunk
    calls = []
    res2 = npde.solve(prob, npde.Adam(0.01), maxiters=200, callback=lambda st, l: calls.append(st["iter"]) or True)
    assert calls == [50] and len(res2.losses) == 50
    # StochasticTraining -> on-device sampler
    disc = npde.PhysicsInformedNN(chain, npde.StochasticTraining(64, bcs_points=32, rng=np.random.default_rng(3)), init_params=th0, precision="f32")
    prob = npde.discretize(sysm, disc)
    res = npde.solve(prob, npde.Adam(0.01), maxiters=6)
    assert np.all(np.isfinite(res.losses)) and len(set(np.round(res.losses, 12))) == 6
    rep = prob.pinnrep
    lb, ub, n, seed, kind = rep._device_samplers[1]
    assert lb[0] == ub[0] == 0.0 and n == 32 and kind == 1        # bc u(0, y): x pinned to 0, y in [1/64, 1 - 1/64]
    r = rep.engine.residual(1, res.u, 32)              # runs on the device-sampled set
    assert r.shape == (32,) and np.all(np.isfinite(r))
    pts = rep.engine.get_points(1, 2, 32)
    assert np.all(pts[0] == 0.0) and pts[1].min() >= 1 / 64 - 1e-6 and pts[1].max() <= 1 - 1 / 64 + 1e-6
    # QuasiRandomTraining(resampling = true) with its default LatinHypercubeSample -> on-device Latin-hypercube redraw
    disc = npde.PhysicsInformedNN(chain, npde.QuasiRandomTraining(100, bcs_points=37, sampling_alg=npde.LatinHypercubeSample(seed=5)),
                                  init_params=th0, precision="f32")
    prob = npde.discretize(sysm, disc)
    res = npde.solve(prob, npde.Adam(0.01), maxiters=4)
    assert np.all(np.isfinite(res.losses)) and len(set(np.round(res.losses, 12))) == 4
    rep = prob.pinnrep
    assert rep._device_samplers[0][4] == 2
    draws = []
    for n_, k, d_ in ((100, 0, 2), (37, 3, 2)):
        p = rep.engine.get_points(k, d_, n_).astype(np.float64)
        lb, ub = rep._device_samplers[k][0], rep._device_samplers[k][1]
        for i in range(d_):
            if ub[i] == lb[i]:
                assert np.all(p[i] == np.float32(lb[i]))
                continue
            strata = np.floor((p[i] - lb[i]) / (ub[i] - lb[i]) * n_ - 1e-9).astype(int).clip(0, n_ - 1)
            assert sorted(strata) == list(range(n_))            # every stratum exactly once along every free axis
        draws.append(p)
    assert not np.array_equal(np.argsort(draws[0][0]), np.argsort(draws[0][1]))      # axes are permuted independently


def test_adam_loop_graph_replay_matches_plain_launches(npde, use_emu, monkeypatch):
    """PINN_GRAPH=1 (step index, bias corrections and draw counters in device memory, one step recorded and replayed as a hipGraph; on
    the emulation: the same device-counter kernels launched one by one) reproduces the default loop bit for bit, with fixed point sets
    and with on-device resampling, across a resume."""
    sysm, chain = poisson2d(npde)
    th0 = theta_for(chain, 53)
    for strat in (lambda: npde.GridTraining(0.25), lambda: npde.StochasticTraining(64, bcs_points=32, rng=np.random.default_rng(3))):
        out = []
        for graph in (False, True):
            if graph:
                monkeypatch.setenv("PINN_GRAPH", "1")
            else:
                monkeypatch.delenv("PINN_GRAPH", raising=False)
            prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, strat(), init_params=th0, precision="f32"))
            res = npde.solve(prob, npde.Adam(0.01), maxiters=20)
            res_b = npde.solve(npde.remake(prob, u0=res.u), npde.Adam(0.01), maxiters=12)
            out.append((np.asarray(res.losses), res.u, np.asarray(res_b.losses), res_b.u))
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b)
    monkeypatch.delenv("PINN_GRAPH", raising=False)


def test_library_lbfgs(npde, use_emu):
    """`pinn_lbfgs` (two-loop L-BFGS + Armijo backtracking inside the library, one fused evaluation per objective call): monotone decrease,
    a far lower objective than the same number of Adam iterations reaches, agreement of the reported history with re-evaluated losses,
    and refusal when a term's points are redrawn on the device."""
    sysm, chain = poisson2d(npde, "tanh")
    th0 = theta_for(chain, 61)
    disc = npde.PhysicsInformedNN(chain, npde.GridTraining(0.1), init_params=th0, precision="f32")
    prob = npde.discretize(sysm, disc)
    rep = prob.pinnrep
    w = rep._weights_now()
    f0 = float(prob.f.value_and_grad(th0)[0])
    theta, hist = rep.engine.lbfgs(th0, 150, w)
    assert len(hist) >= 20 and np.all(np.diff(hist) <= 1e-12) and hist[-1] < 1e-2 * f0
    f1 = float(prob.f.value_and_grad(theta)[0])
    assert abs(f1 - hist[-1]) <= 1e-5 * abs(f1) + 1e-12
    res_adam = npde.solve(prob, npde.Adam(0.01), maxiters=150)
    assert hist[-1] < res_adam.losses[-1]
    res = npde.solve(prob, npde.LBFGS(), maxiters=150)                  # the mirror's LBFGS goes through the same entry point
    np.testing.assert_allclose(res.losses[-1], hist[-1], rtol=1e-12)
    with pytest.raises(npde.EngineError, match="history in 1..64"):
        rep.engine.lbfgs(th0, 10, w, history=0)
    disc_s = npde.PhysicsInformedNN(chain, npde.StochasticTraining(64, bcs_points=32, rng=np.random.default_rng(3)), init_params=th0, precision="f32")
    prob_s = npde.discretize(sysm, disc_s)
    npde.solve(prob_s, npde.Adam(0.01), maxiters=2)                     # installs the device samplers
    with pytest.raises(npde.EngineError, match="fixed objective"):
        prob_s.pinnrep.engine.lbfgs(th0, 5, prob_s.pinnrep._weights_now())


def test_device_sobol_sampler_matches_reference_sequence(npde, use_emu):
    """kind-3 device sampler == elements 1..n of the un-randomised Sobol' sequence (Joe-Kuo direction numbers, Gray-code order,
    first element skipped as Sobol.jl does): bit-exact against scipy.stats.qmc.Sobol(scramble=False), which shares the table;
    with a seed every draw is a fresh digital shift of the same net."""
    from scipy.stats import qmc
    sysm, chain = poisson2d(npde)
    th0 = theta_for(chain, 52)
    strat = npde.QuasiRandomTraining(256, bcs_points=64, sampling_alg=npde.SobolSample(scramble=False))
    prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th0, precision="f32"))
    res = npde.solve(prob, npde.Adam(0.01), maxiters=3)
    rep = prob.pinnrep
    assert np.all(np.isfinite(res.losses)) and rep._device_samplers[0][4] == 3 and rep._device_samplers[0][3] == 0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = qmc.Sobol(2, scramble=False).random(257)[1:].T
    lb, ub = rep._device_samplers[0][0], rep._device_samplers[0][1]
    want = (np.float32(lb)[:, None] + (np.float32(ub) - np.float32(lb))[:, None] * ref.astype(np.float32)).astype(np.float32)
    got = rep.engine.get_points(0, 2, 256)
    assert np.array_equal(got, want)                      # same design on every draw (3 Adam steps = 3 draws)
    # boundary term u(0, y): the pinned axis stays at its constant, the free one follows its own axis of the sequence
    pb = rep.engine.get_points(1, 2, 64)
    assert np.all(pb[0] == 0.0) and len(np.unique(pb[1])) == 64
    # engine level, d = 4 .. 8 axes (a term of a d-input net is needed only for the buffer shape: use the raw entry point on term 0)
    eng = rep.engine
    eng.set_sampler(0, [0.0, 0.0], [1.0, 1.0], 1024, seed=0, kind=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = qmc.Sobol(2, scramble=False).random(1025)[1:].T.astype(np.float32)
    assert np.array_equal(eng.get_points(0, 2, 1024), ref)
    # seeded: digitally shifted nets — elements 1..1023 still occupy distinct elementary intervals of length 1/1024 along every
    # axis (element 1024 stands in for the skipped element 0, so at most one interval is hit twice)
    eng.set_sampler(0, [0.0, 0.0], [1.0, 1.0], 1024, seed=77, kind=3)
    a = eng.get_points(0, 2, 1024)
    assert not np.array_equal(a, ref)
    for i in range(2):
        assert len(set(np.floor(a[i].astype(np.float64) * 1024).astype(int))) >= 1023


def test_sobol_direction_numbers_up_to_8_axes(npde, use_emu):
    """all eight supported axes of the device Sobol' sampler against scipy: a 3-input problem through the sampler entry point
    (axes 1-3), and the bit generator itself for axes 1-8 through the emulation build's test hook."""
    from scipy.stats import qmc
    import ctypes, warnings
    t, x, y = npde.parameters("t x y")
    (u,) = npde.variables("u")
    U = u(t, x, y)
    eq = npde.Eq(npde.Differential(t)(U) + npde.Differential(x)(U) + npde.Differential(y)(U), 0)
    bcs = [npde.Eq(u(0, x, y), 0)]
    dom = [npde.In(v, npde.Interval(0.0, 1.0)) for v in (t, x, y)]
    sysm = npde.PDESystem([eq], bcs, dom, [t, x, y], [U])
    chain = npde.Chain(npde.Dense(3, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1))
    th0 = theta_for(chain, 53)
    strat = npde.QuasiRandomTraining(512, bcs_points=32, sampling_alg=npde.SobolSample(scramble=False))
    prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th0, precision="f32"))
    eng = prob.pinnrep.engine
    eng.set_sampler(0, [0.0] * 3, [1.0] * 3, 512, seed=0, kind=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = qmc.Sobol(3, scramble=False).random(513)[1:].T.astype(np.float32)
        ref8 = qmc.Sobol(8, scramble=False, bits=32).random(4097)[1:]
    assert np.array_equal(eng.get_points(0, 3, 512), ref)
    hook = getattr(eng.L.lib, "pinn_emu_sobol_bits", None)
    if hook is not None:                                  # emulation build only
        hook.restype, hook.argtypes = ctypes.c_uint, [ctypes.c_uint, ctypes.c_int]
        for axis in range(8):
            got = np.array([hook(i + 1, axis) for i in range(4096)], dtype=np.float64) / 2.0 ** 32
            assert np.array_equal(got, ref8[:, axis]), axis


def test_hoisted_sources_and_mixed_ops(npde, use_emu):
    """coordinate-only subexpressions (variable coefficients, source terms, boundary data) are evaluated by k_src once per
    point set; what stays in the fused tape mixes dispatch-free arithmetic with transcendental ops of u."""
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dx, Dy = npde.Differential(x), npde.Differential(y)
    Dxx, Dyy = Dx ** 2, Dy ** 2
    U = u(x, y)
    a = sp.exp(-x) * sp.cos(y)                          # used twice: as a coefficient and inside the source
    eq = npde.Eq(a * Dxx(U) + (1 + x * y) * Dyy(U) + sp.sin(U) * Dx(U) - U ** 2 / (2 + sp.cos(x)),
                 a * sp.sin(sp.pi * x) + sp.sqrt(1 + y) - 3.0)
    bcs = [npde.Eq(u(0, y), sp.cos(sp.pi * y) * sp.exp(y)), npde.Eq(u(x, 1), x ** 3 - x), npde.Eq(Dx(u(1, y)), 0.5)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [x, y], [U])
    for width, seed in ((16, 21), (64, 22)):            # family 1 (LDS tape) and family 2 (register tape, one tape wave per point group)
        hidden = 2 if width == 16 else 4                # compiled kernels: 2 x 16 (family 1), 4 x 64 (family 2)
        chain = npde.Chain(npde.Dense(2, width, "tanh"), *[npde.Dense(width, width, "tanh") for _ in range(hidden - 1)], npde.Dense(width, 1))
        strat = npde.QuasiRandomTraining(50, bcs_points=37, sampling_alg=npde.SobolSample(seed=8), resampling=False, minibatch=1)
        rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, seed), weights=[1.0, 2.0, 0.5, 3.0])
        assert "sources" in rep.engine.describe()
        r = rep.loss_functions.datafree_pde_loss_functions[0](sets[0], th)
        np.testing.assert_allclose(r, po.residual_values(prob, th, 0, sets[0]), rtol=2e-5, atol=2e-5)
        # a new point set re-evaluates the sources
        new = [s[:, ::-1].copy() * 0.9 + 0.05 for s in sets]
        new[1][0, :] = 0.0; new[2][1, :] = 1.0; new[3][0, :] = 1.0
        for k, s in enumerate(new):
            rep.engine.set_points(k, s)
        losses, grad = rep.engine.loss_grad(th, [1.0, 2.0, 0.5, 3.0])
        ref = po.loss_and_grad(prob, th, new, weights=[1.0, 2.0, 0.5, 3.0], mode="stencil")
        le, g2, gi = helpers.rel_errors(losses, grad, ref)
        assert le.max() < TOL and g2 < TOL and gi < TOL, (le, g2, gi)


def _third_order_ode(npde):
    # test/NNPDE1/nnpde__pde_iii_3rd_order_ode.jl:66-80
    (x,) = npde.parameters("x")
    (u,) = npde.variables("u")
    Dx, Dxxx = npde.Differential(x), npde.Differential(x) ** 3
    eq = npde.Eq(Dxxx(u(x)), sp.cos(sp.pi * x))
    bcs = [npde.Eq(u(0.0), 0.0), npde.Eq(u(1.0), sp.cos(sp.pi)), npde.Eq(Dx(u(1.0)), 1.0)]
    return npde.PDESystem([eq], bcs, [npde.In(x, npde.Interval(0.0, 1.0))], [x], [u(x)])


def _ks(npde):
    # docs/src/examples/ks.md:33-62 (alpha = 1, beta = 4, gamma = 1), Dirichlet + Neumann data from the analytic solution
    x, t = npde.parameters("x t")
    (u,) = npde.variables("u")
    Dt, Dx = npde.Differential(t), npde.Differential(x)
    Dx2, Dx3, Dx4 = Dx ** 2, Dx ** 3, Dx ** 4
    U = u(x, t)
    th = lambda xx, tt: xx / 2 - 1.2 * tt                  # (scaled so that tanh stays away from saturation on the test box)
    ua = lambda xx, tt: 11 + 15 * sp.tanh(th(xx, tt)) - 15 * sp.tanh(th(xx, tt)) ** 2 - 15 * sp.tanh(th(xx, tt)) ** 3
    eq = npde.Eq(Dt(U) + U * Dx(U) + 1 * Dx2(U) + 4 * Dx3(U) + 1 * Dx4(U), 0)
    bcs = [npde.Eq(u(x, 0), ua(x, 0)), npde.Eq(u(-1.0, t), ua(-1.0, t)), npde.Eq(u(1.0, t), ua(1.0, t)),
           npde.Eq(Dx(u(-1.0, t)), sp.diff(ua(x, t), x).subs(x, -1.0)), npde.Eq(Dx(u(1.0, t)), sp.diff(ua(x, t), x).subs(x, 1.0))]
    dom = [npde.In(x, npde.Interval(-1.0, 1.0)), npde.In(t, npde.Interval(0.0, 1.0))]
    return npde.PDESystem([eq], bcs, dom, [x, t], [U])


def test_third_and_fourth_order_derivatives(npde, use_emu):
    """pure d3/dx3, d4/dx4 jets (Faa di Bruno through the activation, hand-derived adjoints) against the float64 oracle's exact
    derivatives; the reference's own order-3/4 stencils (eps^(1/5), eps^(1/6)) agree with them to ~1e-5 only, checked loosely."""
    # the reference's 3rd-order ODE set-up: Dense(1, 8, sigma) -> Dense(8, 1)
    sysm = _third_order_ode(npde)
    chain = npde.Chain(npde.Dense(1, 8, "sigmoid"), npde.Dense(8, 1))
    rep, prob, sets, th = check(npde, sysm, [chain], npde.GridTraining(0.05), theta_for(chain, 31), mode="exact")
    ref_fd = po.loss_and_grad(prob, th, sets, mode="stencil")
    losses, grad = rep.engine.loss_grad(th)
    le, g2, gi = helpers.rel_errors(losses, grad, ref_fd)
    assert le.max() < 2e-4 and g2 < 2e-4, (le, g2)
    # fourth order, 1-D, two hidden layers, both activations
    (x,) = npde.parameters("x")
    (u,) = npde.variables("u")
    D4, D3, D2 = npde.Differential(x) ** 4, npde.Differential(x) ** 3, npde.Differential(x) ** 2
    eq = npde.Eq(D4(u(x)) + 0.5 * u(x) * D3(u(x)) - D2(u(x)) ** 2, sp.sin(2 * x))
    sys4 = npde.PDESystem([eq], [npde.Eq(u(0.0), 0.0), npde.Eq(D2(u(1.0)), 0.3)], [npde.In(x, npde.Interval(0.0, 1.0))], [x], [u(x)])
    for act, seed in (("tanh", 32), ("sigmoid", 33)):
        chain = npde.Chain(npde.Dense(1, 16, act), npde.Dense(16, 16, act), npde.Dense(16, 1))
        check(npde, sys4, [chain], npde.GridTraining(0.04), theta_for(chain, seed), weights=[1.0, 2.0, 0.5], mode="exact")


def test_kuramoto_sivashinsky_jets(npde, use_emu):
    sysm = _ks(npde)
    strat = npde.QuasiRandomTraining(40, bcs_points=20, sampling_alg=npde.SobolSample(seed=12), resampling=False, minibatch=1)
    # the reference's KS net (Dense(2,12,sigma) x2, padded to 16; family 1) and a 4x64 tanh net (family 2)
    small = npde.Chain(npde.Dense(2, 12, "sigmoid"), npde.Dense(12, 12, "sigmoid"), npde.Dense(12, 1))
    rep, prob, sets, th = check(npde, sysm, [small], strat, theta_for(small, 41), mode="exact")
    assert "_H4_" in rep.engine.describe()                      # the kernel carrying d3/dx3, d4/dx4 along axis 0
    r = rep.loss_functions.datafree_pde_loss_functions[0](sets[0], th)
    np.testing.assert_allclose(r, po.residual_values(prob, th, 0, sets[0], mode="exact"), rtol=3e-5, atol=3e-5)
    big = npde.Chain(npde.Dense(2, 64, "tanh"), *[npde.Dense(64, 64, "tanh") for _ in range(3)], npde.Dense(64, 1))
    check(npde, sysm, [big], strat, theta_for(big, 42), weights=[1.0, 1.0, 2.0, 2.0, 0.5, 0.5], mode="exact")


def test_heterogeneous_system(npde, use_emu):
    """dependent variables with different argument lists in one system (test/NNPDE1/nnpde__pde_i_heterogeneous_system.jl:58-80):
    u(x,y,z), v(y,x), h(z), p(x,z) — every network reads its own rows of the term's coordinate matrix (permuted for v)."""
    x, y, z = npde.parameters("x y z")
    u, v, h, p = npde.variables("u v h p")
    Dz = npde.Differential(z)
    eqs = [npde.Eq(u(x, y, z), x + y + z),
           npde.Eq(v(y, x), x ** 2 + y ** 2),
           npde.Eq(h(z), sp.cos(z)),
           npde.Eq(p(x, z), sp.exp(x) * sp.exp(z)),
           npde.Eq(u(x, y, z) + v(y, x) * Dz(h(z)) - p(x, z), x + y + z - (x ** 2 + y ** 2) * sp.sin(z) - sp.exp(x) * sp.exp(z))]
    bcs = [npde.Eq(u(0.0, 0.0, 0.0), 0.0)]
    dom = [npde.In(s, npde.Interval(0.0, 1.0)) for s in (x, y, z)]
    sysm = npde.PDESystem(eqs, bcs, dom, [x, y, z], [u(x, y, z), v(y, x), h(z), p(x, z)])
    chains = [npde.Chain(npde.Dense(n, 12, "tanh"), npde.Dense(12, 12, "tanh"), npde.Dense(12, 1)) for n in (3, 2, 1, 2)]
    theta = np.concatenate([theta_for(c, 50 + i) for i, c in enumerate(chains)])
    rep, prob, sets, th = check(npde, sysm, chains, npde.GridTraining(0.25), theta, weights=[1.0, 2.0, 0.5, 1.5, 3.0, 1.0])
    d = rep.engine.describe()
    assert "coupled" in d
    # the permuted-input network alone: residual of v(y, x) ~ x^2 + y^2 on its own (y, x) point matrix
    r = rep.loss_functions.datafree_pde_loss_functions[1](sets[1], th)
    np.testing.assert_allclose(r, po.residual_values(prob, th, 1, sets[1]), rtol=2e-5, atol=2e-5)


def test_per_layer_activations(npde, use_emu):
    """tanh and sigmoid mixed inside one chain (the reference's Lorenz chains: Dense(1, n, tanh), Dense(n, n, σ), Dense(n, 1)): kernel
    variant ACT_MIXED of the small-net specs — the layer kind enters the activation rules as a scalar (tanh(z) = 2 s(2z) - 1), no branch."""
    for d, width, acts, seed in ((2, 12, ("tanh", "sigmoid"), 71), (2, 16, ("sigmoid", "tanh", "sigmoid"), 72), (1, 8, ("tanh", "sigmoid"), 73),
                                 (1, 10, ("sigmoid", "sigmoid", "tanh"), 74)):
        sysm, _ = helpers.shape_problem(npde, width, len(acts), d)
        layers = [npde.Dense(d, width, acts[0])] + [npde.Dense(width, width, a) for a in acts[1:]] + [npde.Dense(width, 1)]
        chain = npde.Chain(*layers)
        assert chain.act == ",".join(acts)
        strat = npde.QuasiRandomTraining(50, bcs_points=30, sampling_alg=npde.SobolSample(seed=seed), resampling=False, minibatch=1)
        rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, seed), mode="exact")
        pts = np.array([[0.3, 0.7], [0.6, 0.2]])[:d]
        np.testing.assert_allclose(rep.phi(pts, th)[0], po.phi_values(prob.chains[0], th, pts)[0], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError, match="mixes only tanh and sigmoid"):
        npde.Chain(npde.Dense(2, 8, "sin"), npde.Dense(8, 8, "tanh"), npde.Dense(8, 1))
    # shapes without the mixed variant fail loudly at discretize time
    sysm, _ = helpers.shape_problem(npde, 64, 4, 2)
    big = npde.Chain(npde.Dense(2, 64, "tanh"), npde.Dense(64, 64, "sigmoid"), npde.Dense(64, 64, "tanh"), npde.Dense(64, 64, "tanh"), npde.Dense(64, 1))
    with pytest.raises(Exception, match="per-layer tanh/sigmoid"):
        npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(big, npde.GridTraining(0.25), init_params=theta_for(big, 75), precision="f32"))


def test_bpinn_physics_loglikelihood(npde, use_emu):
    """l(theta) = sum_k logpdf(MvNormal(r_k, sigma_k^2 I), 0) (src/training_strategies.jl:113-127, ext/bpinn/PDE_BPINN.jl:425)
    and its gradient from the engine's per-term sums, against the oracle's residuals."""
    sysm, chain = poisson2d(npde, "tanh")
    th = theta_for(chain, 61)
    rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.125), init_params=th, precision="f32"))
    sets = rep.pde_train_sets + rep.bcs_train_sets
    stds = [0.7, 0.05, 0.08, 0.11, 0.2]
    sizes = [s.shape[1] for s in sets]
    ll, g = npde.physics_loglikelihood(rep.engine, th, stds, sizes)
    prob = helpers.oracle_problem(npde, sysm, [chain])
    ll_ref = 0.0
    for k, (s, sd) in enumerate(zip(sets, stds)):
        r = po.residual_values(prob, th, k, s).reshape(-1)
        ll_ref += -0.5 * r.size * np.log(2 * np.pi) - r.size * np.log(sd) - np.sum(r * r) / (2 * sd * sd)
    assert abs(ll - ll_ref) < 1e-5 * abs(ll_ref)
    w = [n / (2 * sd * sd) for n, sd in zip(sizes, stds)]
    ref = po.loss_and_grad(prob, th, sets, weights=w, mode="stencil")
    assert np.linalg.norm(g + ref.grad) < 1e-5 * np.linalg.norm(ref.grad)
    with pytest.raises(ValueError):
        npde.physics_loglikelihood(rep.engine, th, stds[:-1], sizes[:-1])


def test_forward_laplacian_fusion(npde, use_emu):
    """pure second derivatives that occur only summed (with one common coefficient) travel as ONE jet channel (program.cpp:
    fuse_laplacian, JetSet::LAP); sums with unequal coefficients, or second derivatives used elsewhere too, keep their own channels."""
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dx, Dy = npde.Differential(x), npde.Differential(y)
    Dxx, Dyy = Dx ** 2, Dy ** 2
    U = u(x, y)
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(Dx(u(1, y)), 0.2)]
    chain = npde.Chain(npde.Dense(2, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1))
    strat = npde.QuasiRandomTraining(45, bcs_points=20, sampling_alg=npde.SobolSample(seed=4), resampling=False, minibatch=1)
    cases = [(npde.Eq(-0.3 * (Dxx(U) + Dyy(U)) + U * Dx(U) + sp.exp(x) * Dy(U), sp.cos(x * y)), "_L3_"),     # c * lap + other leaves
             (npde.Eq(Dxx(U) + Dyy(U), 0), "_L3_"),                                                          # residual == lap
             (npde.Eq(Dxx(U) + 2 * Dyy(U), sp.sin(x)), "_L0_"),                                               # unequal coefficients: not fused
             (npde.Eq(Dxx(U) + Dyy(U) + Dxx(U) * U, 0), "_L0_")]                                              # u_xx used twice: not fused
    for k, (eq, tag) in enumerate(cases):
        sysm = npde.PDESystem([eq], bcs, dom, [x, y], [U])
        rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, 70 + k), weights=[1.0, 2.0, 0.5])
        kern = [l for l in rep.engine.describe().split("\n") if l.startswith("group 0")][0]
        assert tag in kern, kern
        r = rep.loss_functions.datafree_pde_loss_functions[0](sets[0], th)
        np.testing.assert_allclose(r, po.residual_values(prob, th, 0, sets[0]), rtol=3e-5, atol=3e-5)


def test_sin_activation(npde, use_emu):
    """sin hidden activations (test/NNPDE2/direct_function__approximation_of_function_1d_2.jl:23-26): the layer records keep the
    pre-activation z (cos z is not a function of sin z); second derivatives and, on the KS set, phi''''' = cos."""
    sysm, _ = poisson2d(npde, "tanh")
    chain = npde.Chain(npde.Dense(2, 16, "sin"), npde.Dense(16, 16, "sin"), npde.Dense(16, 1))
    strat = npde.QuasiRandomTraining(50, bcs_points=21, sampling_alg=npde.SobolSample(seed=6), resampling=False, minibatch=1)
    check(npde, sysm, [chain], strat, theta_for(chain, 81), weights=[1.0, 2.0, 0.5, 1.5, 1.0])
    big = npde.Chain(npde.Dense(2, 64, "sin"), *[npde.Dense(64, 64, "sin") for _ in range(3)], npde.Dense(64, 1))
    check(npde, sysm, [big], strat, theta_for(big, 82))
    check(npde, _ks(npde), [chain], strat, theta_for(chain, 83), mode="exact")


def test_long_residual_takes_the_two_launch_path(npde, use_emu):
    """a single-network residual with more rows than the fused kernel's 32-row tape is not an error: it is evaluated by the
    forward launch -> k_expr (96 rows) -> reverse launch path that systems of equations use."""
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    Dx, Dy = npde.Differential(x), npde.Differential(y)
    Dxx, Dyy = Dx ** 2, Dy ** 2
    U = u(x, y)
    expr = Dxx(U) * sp.exp(U) + Dyy(U) * sp.cos(U) + U ** 3 * Dx(U) + sp.tanh(U) * Dy(U) + sp.sin(U) * sp.cos(U) / (1 + U ** 2) \
        + sp.exp(-U ** 2) * Dx(U) ** 2 + sp.log(1 + U ** 2) * Dy(U) ** 2 + sp.sqrt(1 + U ** 2) + sp.sinh(U) * sp.cosh(U) * 1e-2 \
        + Dx(Dy(U)) * U + sp.Abs(U) * 0.1 + (U + 0.5) ** 4 * 0.01
    eq = npde.Eq(expr, sp.sin(x) * sp.cos(y))
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(u(x, 1), x)]
    dom = [npde.In(x, npde.Interval(0.0, 1.0)), npde.In(y, npde.Interval(0.0, 1.0))]
    sysm = npde.PDESystem([eq], bcs, dom, [x, y], [U])
    chain = npde.Chain(npde.Dense(2, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1))
    strat = npde.QuasiRandomTraining(45, bcs_points=20, sampling_alg=npde.SobolSample(seed=4), resampling=False, minibatch=1)
    rep, prob, sets, th = check(npde, sysm, [chain], strat, theta_for(chain, 91), weights=[1.0, 2.0, 0.5])
    assert len(rep.ir.terms[0].ops) > 32 and "coupled" in rep.engine.describe()


def test_forward_derivatives_mirror(npde, use_emu):
    """Mirror of test/Forward/forward__derivatives.jl:7-44 at the C ABI: a 2 -> 16 -> 16 -> 1 sigmoid chain at the point [1, 2];
    `pinn_derivative` (the engine's numeric_derivative, exact Taylor jets carried by the forward kernel) against (a) the exact
    derivatives (the reference compares with Zygote.gradient / Zygote.hessian) and (b) the reference's central-difference stencils with
    its get_eps steps, at the reference's tolerances: first order atol 1e-8 in Float64 — here bounded by fp32 arithmetic, so 2e-6 —
    and second order (xx, xy, yy) atol 4e-5."""
    import torch
    sysm, _ = helpers.shape_problem(npde, 16, 2, 2)                # any residual with u_x, u_y, u_xx, u_xy, u_yy: binds the network to the Hessian kernel
    chain = npde.Chain(npde.Dense(2, 16, "sigmoid"), npde.Dense(16, 16, "sigmoid"), npde.Dense(16, 1))
    ochain = po.Chain((2, 16, 16, 1), "sigmoid")
    u = lambda cord, th, phi: phi(cord, th).sum(dim=0, keepdim=True)      # u_ of the reference test
    x = np.array([[1.0], [2.0]])
    xt = torch.tensor(x, dtype=po.DT)
    for seed in range(3):
        theta = po.glorot_theta(ochain, np.random.default_rng(seed), bias_amp=0.0)
        rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.25), init_params=theta, precision="f32"))
        eng = rep.engine
        tht = torch.tensor(theta, dtype=po.DT)
        assert abs(eng.derivative(0, theta, x, [])[0] - float(ochain(xt, tht))) < 2e-6             # phi([1, 2], theta)
        for ax in (0, 1):
            got = float(eng.derivative(0, theta, x, [ax])[0])
            fd = float(po.numeric_derivative(ochain, u, xt, [po.get_eps(2, ax + 1, np.float64, 1)], 1, tht))
            ex = float(po.exact_derivative(ochain, u, xt, [ax], tht))
            assert abs(got - ex) < 2e-6 and abs(got - fd) < 2e-6, (ax, got, ex, fd)
        ex_, ey_ = po.get_eps(2, 1, np.float64, 2), po.get_eps(2, 2, np.float64, 2)
        for epss, axes in [([ex_, ex_], [0, 0]), ([ex_, ey_], [0, 1]), ([ey_, ey_], [1, 1])]:
            got = float(eng.derivative(0, theta, x, axes)[0])
            fd = float(po.numeric_derivative(ochain, u, xt, epss, 2, tht))
            ex = float(po.exact_derivative(ochain, u, xt, axes, tht))
            assert abs(got - ex) < 4e-5 and abs(got - fd) < 4e-5, (axes, got, ex, fd)
        assert abs(float(eng.derivative(0, theta, x, [1, 0])[0]) - float(eng.derivative(0, theta, x, [0, 1])[0])) == 0.0    # axes are sorted
    # a batch of points, and the loud failures
    pts = np.random.default_rng(5).uniform(0, 1, size=(2, 100))
    got = eng.derivative(0, theta, pts, [0, 0])
    ex = po.exact_derivative(ochain, u, torch.tensor(pts, dtype=po.DT), [0, 0], tht).detach().numpy().reshape(-1)
    assert np.max(np.abs(got - ex)) < 1e-5
    got = eng.derivative(0, theta, pts, [0, 0, 1])                   # mixed third derivative: a generated jet set (csrc/jit.cpp)
    ex = po.exact_derivative(ochain, u, torch.tensor(pts, dtype=po.DT), [0, 0, 1], tht).detach().numpy().reshape(-1)
    assert np.max(np.abs(got - ex)) < 2e-5 * max(1.0, np.max(np.abs(ex)))
    with pytest.raises(Exception, match="order must be 0..6"):
        eng.derivative(0, theta, pts, [0] * 7)
    with pytest.raises(Exception, match="axis out of range"):
        eng.derivative(0, theta, pts, [2])


def test_bpinn_loglikelihood_with_std_gradients_and_data_term(npde, use_emu):
    """pinn_loglik_grad: l(theta, sigma) = sum_k logpdf(MvNormal(r_k, sigma_k^2 I), 0) over the pde, bc AND an L2 data term (a DataLoss
    term = L2LossData of ext/bpinn/PDE_BPINN.jl:148-183), with d l / d theta (against the oracle's weighted gradient) and d l / d sigma_k
    (against central differences of the oracle's l) in one call."""
    sysm, chain = poisson2d(npde, "tanh")
    th = theta_for(chain, 62)
    rng = np.random.default_rng(9)
    xd = rng.uniform(0.1, 0.9, size=(2, 24))
    yd = np.sin(np.pi * xd[0]) * np.sin(np.pi * xd[1]) / (2 * np.pi ** 2) + 0.01 * rng.standard_normal(24)
    disc = npde.PhysicsInformedNN(chain, npde.GridTraining(0.125), init_params=th, data_loss=[npde.DataLoss(sysm.dvs[0], xd, yd)], precision="f32")
    rep = npde.symbolic_discretize(sysm, disc)
    eng = rep.engine
    sets = rep.pde_train_sets + rep.bcs_train_sets
    stds = np.array([0.7, 0.05, 0.08, 0.11, 0.2, 0.03])
    ll, g, gs = npde.loglikelihood(eng, th, stds)
    prob = helpers.oracle_problem(npde, sysm, [chain])
    ochain = po.Chain(tuple(chain.sizes), chain.act)

    def sse():
        out = [float(np.sum(po.residual_values(prob, th, k, s) ** 2)) for k, s in enumerate(sets)]
        out.append(float(np.sum((po.phi_values(ochain, th, xd)[0] - yd) ** 2)))
        return np.array(out)

    S = sse()
    N = np.array([s.shape[1] for s in sets] + [24], dtype=np.float64)
    ll_of = lambda sd: float(np.sum(-0.5 * N * np.log(2 * np.pi) - N * np.log(sd) - S / (2 * sd ** 2)))
    assert abs(ll - ll_of(stds)) < 1e-5 * abs(ll_of(stds))
    for k in range(6):                                        # d l / d sigma_k
        e = np.zeros(6); e[k] = 1e-6 * stds[k]
        fd = (ll_of(stds + e) - ll_of(stds - e)) / (2 * e[k])
        assert abs(gs[k] - fd) < 1e-5 * max(1.0, abs(fd)), (k, gs[k], fd)
    # d l / d theta = - grad sum_k w_k L_k, w_k = N_k / (2 sigma_k^2): physics terms from the oracle, data term by autograd of its SSE
    import torch
    w = N / (2 * stds ** 2)
    ref = po.loss_and_grad(prob, th, sets, weights=w[:5], mode="stencil")
    tht = torch.tensor(th, dtype=po.DT, requires_grad=True)
    sse_d = torch.sum((ochain(torch.tensor(xd, dtype=po.DT), tht)[0] - torch.tensor(yd, dtype=po.DT)) ** 2)
    (gd,) = torch.autograd.grad(sse_d / (2 * stds[5] ** 2), tht)
    gref = -(ref.grad + gd.numpy())
    assert np.linalg.norm(g - gref) < 1e-5 * np.linalg.norm(gref)
    with pytest.raises(Exception, match="positive"):
        eng.loglik_grad(th, [0.1, 0.1, 0.0, 0.1, 0.1, 0.1])


def _merge_cfg2(npde, points=200, bcs_points=70):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=points, bcs_points=bcs_points)
    return wl


def test_merged_launch_matches_chained_launches_and_oracle(npde, use_emu, monkeypatch):
    """interior (forward-Laplacian jet set) + boundary (value-only) launch groups of one 4x64 network run as ONE persistent launch
    (wave_main2m: both kernel-family members in one kernel, weight-gradient accumulators in registers across both tile lists);
    PINN_NO_MERGE=1 restores the two chained launches.  Both against the oracle, and against each other."""
    wl = _merge_cfg2(npde)
    w = [1.0, 2.0, 0.5, 3.0, 1.5]
    rep, _, _, th = check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta, weights=w)
    assert "launch=merged into group 0" in rep.engine.describe()
    l_m, g_m = rep.engine.loss_grad(th, w)
    assert [g["launched_by"] for g in rep.engine.group_timings()] == [0, 0]
    monkeypatch.setenv("PINN_NO_MERGE", "1")
    l_c, g_c = rep.engine.loss_grad(th, w)
    assert [g["launched_by"] for g in rep.engine.group_timings()] == [0, 1]
    monkeypatch.delenv("PINN_NO_MERGE")
    np.testing.assert_allclose(l_m, l_c, rtol=1e-12)                # same forward arithmetic, same per-wave partials
    np.testing.assert_allclose(g_m, g_c, rtol=0, atol=2e-6 * np.abs(g_c).max())
    # a merged evaluation must not leave anything behind that a later un-merged one picks up (per-term launches, loss-only launches)
    tl, tg = rep.engine.term_grads(th)
    np.testing.assert_allclose((np.array(w)[:, None] * tg).sum(axis=0), g_m, rtol=0, atol=2e-6 * np.abs(g_m).max())
    np.testing.assert_allclose(tl, l_m, rtol=1e-12)
    l_again, g_again = rep.engine.loss_grad(th, w)
    assert np.array_equal(l_again, l_m) and np.array_equal(g_again, g_m)


def test_gemm_mode_switch_on_a_live_handle(npde, use_emu):
    """pinn_set_option(h, "gemm", "fp32" | "split") (include/pinn_hip.h): both kernel sets are in the library; the switch rebuilds the kernel
    plan in place and keeps point sets and optimiser state.  fp32 mode = v_mfma_f32_16x16x4_f32 (an fmaf chain per product); split mode =
    three bf16 pieces per operand.  Both against the oracle (`check`), the fp32 mode at least as close; switching back reproduces the first
    result bit for bit; the merged launch exists in both modes."""
    wl = _merge_cfg2(npde, 200, 70)
    w = [1.0, 2.0, 0.5, 3.0, 1.5]
    rep, olosses, _, th = check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta, weights=w)
    eng = rep.engine
    assert eng.get_option("gemm") == "split" and "gemm=split-bf16(fwd,dA,dW)" in eng.describe()
    l_s, g_s = eng.loss_grad(th, w)
    eng.set_option("gemm", "fp32")
    assert eng.get_option("gemm") == "fp32" and "gemm=fp32" in eng.describe() and "launch=merged" in eng.describe()
    l_f, g_f = eng.loss_grad(th, w)
    np.testing.assert_allclose(l_f, l_s, rtol=2e-6)
    np.testing.assert_allclose(g_f, g_s, rtol=0, atol=2e-6 * np.abs(g_s).max())
    assert not np.array_equal(g_f, g_s)                            # (different arithmetic: the two modes do not round alike)
    tl, tg = eng.term_grads(th)                                    # per-term launches, loss-only launches, pinn_phi on the fp32 plan
    np.testing.assert_allclose((np.array(w)[:, None] * tg).sum(axis=0), g_f, rtol=0, atol=2e-6 * np.abs(g_f).max())
    l_lo, _ = eng.loss_grad(th, w, want_grad=False)
    np.testing.assert_allclose(l_lo, l_f, rtol=1e-12)
    t_f, h_f = eng.adam(th, 3, 1e-3, w)
    eng.set_option("gemm", "split")
    l_s2, g_s2 = eng.loss_grad(th, w)
    assert np.array_equal(l_s2, l_s) and np.array_equal(g_s2, g_s)
    t_s, h_s = eng.adam(th, 3, 1e-3, w)
    np.testing.assert_allclose(t_s, t_f, rtol=0, atol=1e-6)
    with pytest.raises(Exception):
        eng.set_option("gemm", "bf16")
    with pytest.raises(Exception):
        eng.set_option("nope", "1")


@pytest.mark.parametrize("scale", [2.0, 4.0])
def test_scaled_parameters_both_gemm_modes(npde, use_emu, scale):
    """saturating networks (theta x 2, x 4: large pre-activations, large second-derivative jets) in both GEMM modes of the 4x64 kernels,
    against both oracle modes — the small-size companion of tests/test_gpu_theta_variants.py (full size, on the hardware).
    Against the EXACT-derivative oracle (the mathematics the engine implements) both modes stay inside the 1e-5 bar, the fp32-MFMA
    kernels closer than the split products.  Against the STENCIL oracle (the reference's finite differences, src/pinn_types.jl:445-482)
    the distance is bounded by the stencil's own truncation error, which at theta x 4 exceeds 1e-5 by itself (stencil vs exact oracle:
    1.4e-5 on this design) — there the engine must be as close to the reference as exact derivatives can be."""
    wl = _merge_cfg2(npde, 96, 64)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    sets = rep.pde_train_sets + rep.bcs_train_sets
    th = np.asarray(rep.flat_init_params, dtype=np.float64) * scale
    w = [1.0, 2.0, 0.5, 3.0, 1.5]
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    ref = {om: po.loss_and_grad(prob, th, sets, weights=w, mode=om) for om in ("exact", "stencil")}
    fd = helpers.rel_errors(ref["stencil"].term_losses, ref["stencil"].grad, ref["exact"])      # the reference's own finite-difference error
    err = {}
    for mode in ("split", "fp32"):
        eng.set_option("gemm", mode)
        losses, grad = eng.loss_grad(th, w)
        for om in ("exact", "stencil"):
            err[(mode, om)] = helpers.rel_errors(losses, grad, ref[om])
        le, g2, gi = err[(mode, "exact")]
        assert le.max() < TOL and g2 < TOL and gi < TOL, (mode, "exact", scale, le, g2, gi)
        le, g2, gi = err[(mode, "stencil")]
        for e, f in ((le.max(), fd[0].max()), (g2, fd[1]), (gi, fd[2])):
            assert e < max(TOL, 1.3 * f + 5e-6), (mode, "stencil", scale, e, f)
    # (r05: with the small piece products on an accumulator of their own the split products round LESS often at full magnitude than an fmaf
    # chain — neither mode is systematically closer any more; what the bar asks is above: both inside 1e-5)
    e_f, e_s = err[("fp32", "exact")][1], err[("split", "exact")][1]
    assert max(e_f, e_s) < 2.5 * min(e_f, e_s) + 1e-7


def test_gemm_mode_from_the_environment_and_128_wide(npde, use_emu, monkeypatch):
    """$PINN_GEMM selects the mode new handles start in; the 128-wide kernels (8-wave workgroups, slab-resident dW) in both modes"""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg2_poisson2d(points=40, bcs_points=20, width=128, hidden=2)
    rep, _, sets, th = check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta)
    assert "gemm=split" in rep.engine.describe()
    l_s, g_s = rep.engine.loss_grad(th)
    monkeypatch.setenv("PINN_GEMM", "fp32")
    rep2 = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    monkeypatch.delenv("PINN_GEMM")
    assert rep2.engine.get_option("gemm") == "fp32" and "gemm=fp32" in rep2.engine.describe()
    for k, sset in enumerate(sets):
        rep2.engine.set_points(k, sset)
    l_f, g_f = rep2.engine.loss_grad(th)
    np.testing.assert_allclose(l_f, l_s, rtol=2e-6)
    np.testing.assert_allclose(g_f, g_s, rtol=0, atol=2e-6 * np.abs(g_s).max())
    rep.engine.set_option("gemm", "fp32")
    l_f2, g_f2 = rep.engine.loss_grad(th)
    assert np.array_equal(l_f2, l_f) and np.array_equal(g_f2, g_f)
    monkeypatch.setenv("PINN_GEMM", "tf32")
    with pytest.raises(Exception):
        npde.symbolic_discretize(wl.pde_system, wl.discretization())


def test_merged_launch_uneven_tile_counts(npde, use_emu):
    """tile counts that do not divide the grid: the tail group's tiles continue the round-robin where the head's stopped"""
    for pts, bpts in ((17, 300), (333, 65), (64, 96)):
        wl = _merge_cfg2(npde, pts, bpts)
        rep, _, _, th = check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta)
        assert "launch=merged" in rep.engine.describe()


def test_one_kernel_reduction_matches_two_stage(npde, use_emu, monkeypatch):
    """a single slab set (merged / chained groups of one network): aux::k_reduce_one; PINN_NO_REDUCE_ONE=1: reduce1 + reduce2"""
    wl = _merge_cfg2(npde, 300, 130)
    w = [1.0, 2.0, 0.5, 3.0, 1.5]
    rep, _, _, th = check(npde, wl.pde_system, wl.chains, wl.strategy, wl.theta, weights=w)
    monkeypatch.setenv("PINN_REDUCE_DIRECT_MAX", "0")             # (the emulated device has 4 workgroup slots: "small" otherwise)
    l1, g1 = rep.engine.loss_grad(th, w)
    monkeypatch.setenv("PINN_NO_REDUCE_ONE", "1")
    l2, g2 = rep.engine.loss_grad(th, w)
    monkeypatch.setenv("PINN_NO_MERGE", "1")                       # chained launches: still one slab set
    monkeypatch.delenv("PINN_NO_REDUCE_ONE")
    l3, g3 = rep.engine.loss_grad(th, w)
    for l, g in ((l2, g2), (l3, g3)):
        np.testing.assert_allclose(l, l1, rtol=1e-12)
        np.testing.assert_allclose(g, g1, rtol=0, atol=2e-6 * np.abs(g1).max())
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    ref = po.loss_and_grad(prob, th, rep.pde_train_sets + rep.bcs_train_sets, weights=w, mode="stencil")
    le, e2, ei = helpers.rel_errors(l1, g1, ref)
    assert le.max() < TOL and e2 < TOL and ei < TOL


def _loss_only_cases(npde):
    from neuralpde_jl_amd import workloads
    yield workloads.cfg1_poisson1d(64)                           # family 1 (3 x 32), boundary terms riding on the interior launch
    yield workloads.cfg2_poisson2d(points=200, bcs_points=70)    # family 2, merged launch in the full evaluation
    yield workloads.cfg4_cavity(points=70, bcs_points=40, width=16, hidden=2)      # coupled equations: forward launches + k_expr
    yield workloads.cfg5_heat_inverse(points=150, bcs_points=70, width=128, hidden=2)   # 8-wave workgroups, PDE parameter


def test_loss_only_evaluation_returns_the_fused_losses(npde, use_emu, monkeypatch):
    """pinn_loss_grad(grad = NULL): MODE_LOSS kernels (forward + tape + sums of squares, no reverse sweep) — the per-point residuals are
    the fused kernel's bit for bit (same forward arithmetic) and their squares are summed in double, so the term losses agree to the
    order of the double-precision sums (the launches partition the points differently): 1e-13 relative, identical once rounded to
    float; with and without term weights"""
    for wl in _loss_only_cases(npde):
        disc = wl.discretization()
        rep = npde.symbolic_discretize(wl.pde_system, disc)
        assert rep.engine.L.backend == EXPECTED_BACKEND
        th = rep.flat_init_params
        K = rep.engine.K
        w = list(np.linspace(0.5, 2.0, K))
        for weights in (None, w):
            l_full, g_full = rep.engine.loss_grad(th, weights)
            l_only, g_only = rep.engine.loss_grad(th, weights, want_grad=False)
            assert g_only is None
            np.testing.assert_allclose(l_only, l_full, rtol=1e-13, atol=0, err_msg=wl.name)
            assert np.array_equal(l_only.astype(np.float32), l_full.astype(np.float32)), (wl.name, l_full, l_only)
        l_again, g_again = rep.engine.loss_grad(th, w)            # a loss-only evaluation leaves nothing behind
        assert np.array_equal(l_again, l_full) and np.array_equal(g_again, g_full)
        monkeypatch.setenv("PINN_REDUCE_DIRECT_MAX", "0")         # the large-launch reductions (one kernel; PINN_NO_REDUCE_ONE: two stages)
        for no_one in (False, True):
            if no_one:
                monkeypatch.setenv("PINN_NO_REDUCE_ONE", "1")
            l_big, _ = rep.engine.loss_grad(th, w, want_grad=False)
            np.testing.assert_allclose(l_big, l_full, rtol=1e-13, atol=0, err_msg=wl.name)
        monkeypatch.delenv("PINN_NO_REDUCE_ONE")
        monkeypatch.delenv("PINN_REDUCE_DIRECT_MAX")
