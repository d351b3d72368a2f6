"""The persistent training kernel (csrc/pinn_train.hpp): K Adam iterations of a small problem inside ONE launch — evaluation, fixed-order
reduction, update and weight-image scatter per iteration, two grid barriers — against the stand-alone loop (three launches per iteration).
Same arithmetic and association: bit-identical parameters and loss histories, across a resume.  Runs the kernel SOURCES on the host
emulation (every wave of the launch a thread); the GPU mirror is tests/test_gpu_mirror.py::test_persistent_training_kernel_*.
Reference loop: solve(prob, Adam(..); maxiters = ...) over full_loss_function (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:83-85,
src/discretize.jl:567-598).

What this file pins is kernel == loop (the engine against itself).  The chain to the oracle runs through the loop: its evaluation against the float64
oracle (tests/test_emu_parity.py, tests/test_gpu_parity.py) and its Adam trajectory against a host float64 Adam over oracle gradients
(tests/test_emu_parity.py::test_resident_adam_matches_host_adam_and_sampler)."""
import numpy as np
import pytest

import sympy as sp

import helpers
from test_emu_parity import _ks, _third_order_ode, poisson2d, theta_for


def _run(npde, eng, th, w, persistent, steps=(9, 4)):
    eng.set_option("persistent", "on" if persistent else "off")
    t1, h1 = eng.adam(th, steps[0], 1e-2, w)
    path = eng.get_option("adam_path")
    t2, h2 = eng.adam(None, steps[1], 1e-2, w, init=False)      # resume: the optimiser state stays on the device
    return t1, h1, t2, h2, path


def _cases(npde):
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg1_poisson1d(100)                               # 3 x 32 tanh, 100 + 1 + 1 points: ragged last tile, 3 terms in one launch group;
    yield "cfg1_3x32_tanh", wl.pde_system, wl.chains[0], wl.strategy, wl.theta, None       # fewer threads than parameters: the kernel's general form
    wl = workloads.cfg1_poisson1d()                                  # BASELINE config 1 at its size: 1,024 + 1 + 1 points (on the GPU: 17 workgroups,
    yield "cfg1_full", wl.pde_system, wl.chains[0], wl.strategy, wl.theta, None            # every thread owns one element, placed by slab entry)
    sysm, chain = poisson2d(npde, "sigmoid", width=16, hidden=2)     # 2 x 16 sigmoid, 2-D, {u, u_x, u_y, lap u}: 25 + 4 x 5 grid points
    yield "poisson2d_2x16_sigmoid", sysm, chain, npde.GridTraining(0.25), theta_for(chain, 7), [1.0, 2.0, 1.0, 3.0, 1.0]


def test_persistent_training_kernel_equals_the_loop_bit_for_bit(npde, use_emu, monkeypatch):
    # (the engine keeps launches with fewer threads than parameters on the loop: nothing to win there; the switch lets the kernel's general
    # form — maps and optimiser state re-read from memory every step — be checked as well, on the emulation's two "CUs" in particular)
    monkeypatch.setenv("PINN_TRAIN_GENERAL", "1")
    for name, sysm, chain, strat, th0, w in _cases(npde):
        disc = npde.PhysicsInformedNN(chain, strat, init_params=th0, precision="f32")
        rep = npde.symbolic_discretize(sysm, disc)
        eng = rep.engine
        wts = None if w is None else np.asarray(w, dtype=np.float32)
        a = _run(npde, eng, th0, wts, False)
        b = _run(npde, eng, th0, wts, True)
        assert a[4] == "loop" and b[4] == "persistent", (name, a[4], b[4], eng.describe())
        for x, y, what in zip(a[:4], b[:4], ("theta", "history", "theta after resume", "history after resume")):
            assert np.array_equal(x, y), (name, what, np.max(np.abs(np.asarray(x) - np.asarray(y))))
        assert np.all(np.isfinite(b[1])) and np.all(np.isfinite(b[3])) and not np.array_equal(b[0], th0.astype(np.float32))
        # the state the kernel leaves behind serves the stand-alone entry points: the packed weight image is the current theta's
        l1, g1 = eng.loss_grad(b[2])
        eng.set_option("persistent", "off")
        t3, h3 = eng.adam(b[2], 1, 1e-2, wts)
        wn = np.ones(eng.K) if wts is None else wts.astype(np.float64)
        assert abs(h3[0] - float(np.dot(l1, wn))) <= 1e-6 * abs(h3[0])


def test_persistent_training_kernel_redraws_the_point_sets_like_the_loop(npde, use_emu, monkeypatch):
    """StochasticTraining / QuasiRandomTraining(resampling = true) (src/training_strategies.jl:242-245, 375-381): the kernel draws the point
    sets of the steps after the first inside the launch (uniform, Latin hypercube, Sobol' — the counter-based rules of the stand-alone
    sampler kernels) and re-evaluates the coordinate-only source channels; parameters, histories AND the point sets left installed equal
    the loop's bit for bit, across a resume (the draw counters continue)."""
    sysm, chain = poisson2d(npde, "tanh", width=16, hidden=2)
    th0 = theta_for(chain, 3)
    strategies = (lambda: npde.StochasticTraining(64, bcs_points=32, rng=np.random.default_rng(3)),
                  lambda: npde.QuasiRandomTraining(100, bcs_points=37, sampling_alg=npde.LatinHypercubeSample(seed=5)),
                  lambda: npde.QuasiRandomTraining(80, bcs_points=16, sampling_alg=npde.SobolSample(seed=9)))
    for make in strategies:
        outs = []
        for mode in ("0 per-term sampler launches", "0", "1"):      # the loop with one sampler + one source launch per term and step, the loop with
            monkeypatch.setenv("PINN_PERSISTENT", mode[0])            # ONE redraw launch per step (aux::k_resample), the persistent kernel
            if len(mode) > 1:
                monkeypatch.setenv("PINN_NO_FUSED_RESAMPLE", "1")
            else:
                monkeypatch.delenv("PINN_NO_FUSED_RESAMPLE", raising=False)
            mode = mode[0]
            prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, make(), init_params=th0, precision="f32"))
            rep = prob.pinnrep                               # the same device-sampler seeds in both runs (an unseeded strategy draws fresh ones)
            rep._device_samplers = {k: (lb, ub, n, 4321 + 17 * k, kind) for k, (lb, ub, n, _, kind) in rep._device_samplers.items()}
            r1 = npde.solve(prob, npde.Adam(0.01), maxiters=7)
            r2 = npde.solve(npde.remake(prob, u0=r1.u), npde.Adam(0.01), maxiters=4)
            eng = prob.pinnrep.engine
            assert eng.get_option("adam_path") == ("persistent" if mode == "1" else "loop")
            pts = [eng.get_points(k, 2, n) for k, (_, _, n, _, _) in sorted(prob.pinnrep._device_samplers.items())]
            outs.append((r1.u, np.asarray(r1.losses), r2.u, np.asarray(r2.losses), pts))
        # the same call split over several launches of the kernel (the engine does that every 4,096 steps): the host redraws the first
        # set of every launch, the barrier counter restarts, the history continues
        monkeypatch.setenv("PINN_TRAIN_CHUNK", "3")
        prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, make(), init_params=th0, precision="f32"))
        rep = prob.pinnrep
        rep._device_samplers = {k: (lb, ub, n, 4321 + 17 * k, kind) for k, (lb, ub, n, _, kind) in rep._device_samplers.items()}
        r1 = npde.solve(prob, npde.Adam(0.01), maxiters=7)
        r2 = npde.solve(npde.remake(prob, u0=r1.u), npde.Adam(0.01), maxiters=4)
        monkeypatch.delenv("PINN_TRAIN_CHUNK")
        pts = [rep.engine.get_points(k, 2, n) for k, (_, _, n, _, _) in sorted(rep._device_samplers.items())]
        outs.append((r1.u, np.asarray(r1.losses), r2.u, np.asarray(r2.losses), pts))
        a = outs[0]
        for b in outs[1:]:
            for x, y in zip(a[:4], b[:4]):
                assert np.array_equal(x, y)
            for x, y in zip(a[4], b[4]):
                assert np.array_equal(x, y)
        assert len(set(np.round(outs[2][1], 12))) == 7         # a new point set every step


def test_persistent_training_kernel_is_refused_where_it_does_not_apply(npde, use_emu, monkeypatch):
    sysm, chain = poisson2d(npde, "tanh", width=16, hidden=2)
    th0 = theta_for(chain, 3)
    # the graph-replay experiment of the loop keeps the loop
    monkeypatch.setenv("PINN_GRAPH", "1")
    prob = npde.discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.25), init_params=th0, precision="f32"))
    npde.solve(prob, npde.Adam(0.01), maxiters=9)
    assert prob.pinnrep.engine.get_option("adam_path") == "loop"
    monkeypatch.delenv("PINN_GRAPH")
    # fixed sets: the kernel; the environment switch: the loop again
    disc = npde.PhysicsInformedNN(chain, npde.GridTraining(0.25), init_params=th0, precision="f32")
    prob = npde.discretize(sysm, disc)
    r1 = npde.solve(prob, npde.Adam(0.01), maxiters=6)
    assert prob.pinnrep.engine.get_option("adam_path") == "persistent"
    monkeypatch.setenv("PINN_PERSISTENT", "0")
    prob2 = npde.discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.25), init_params=th0, precision="f32"))
    r2 = npde.solve(prob2, npde.Adam(0.01), maxiters=6)
    assert prob2.pinnrep.engine.get_option("adam_path") == "loop"
    assert np.array_equal(r1.u, r2.u) and np.array_equal(np.asarray(r1.losses), np.asarray(r2.losses))
    with pytest.raises(Exception):
        prob2.pinnrep.engine.set_option("persistent", "maybe")
    # a launch whose grid barrier times out (workgroups not all resident: a shared device) — forced here — restores the optimiser state
    # and the draw counters, runs the same steps in the loop and keeps the loop for the handle: same numbers, no error
    monkeypatch.delenv("PINN_PERSISTENT")
    monkeypatch.setenv("PINN_TRAIN_FORCE_TIMEOUT", "1")
    prob3 = npde.discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.25), init_params=th0, precision="f32"))
    r3 = npde.solve(prob3, npde.Adam(0.01), maxiters=6)
    eng3 = prob3.pinnrep.engine
    assert eng3.get_option("adam_path") == "loop" and eng3.get_option("persistent") == "off"
    assert np.array_equal(r1.u, r3.u) and np.array_equal(np.asarray(r1.losses), np.asarray(r3.losses))


def _shape_cases(npde):
    """(name, system, chain, strategy, weights): the kernel families / jet sets / residual forms the training kernel wraps"""
    sob = lambda seed, n=48, b=12: npde.QuasiRandomTraining(n, bcs_points=b, sampling_alg=npde.SobolSample(seed=seed), resampling=False, minibatch=1)
    # the reference's 3rd-order ODE net (8 wide, sigmoid): pure d3/dx3 channel, Neumann term
    chain = npde.Chain(npde.Dense(1, 8, "sigmoid"), npde.Dense(8, 1))
    yield "ode3_1x8", _third_order_ode(npde), chain, npde.GridTraining(0.02), None
    # Kuramoto-Sivashinsky (12 wide padded to 16, sigmoid): d4/dx4, a NON-affine residual (u u_x): the tape interpreter inside the kernel
    chain = npde.Chain(npde.Dense(2, 12, "sigmoid"), npde.Dense(12, 12, "sigmoid"), npde.Dense(12, 1))
    yield "ks_2x12", _ks(npde), chain, sob(12, 60, 20), [1.0, 1.0, 2.0, 2.0, 0.5, 0.5]
    # mixed second derivatives in 2-D / 3-D at the reference's widths (full-Hessian jet sets), 1-3 hidden layers
    for width, hidden, d in ((16, 1, 2), (25, 2, 3), (32, 3, 2)):
        sysm, chain = helpers.shape_problem(npde, width, hidden, d)
        yield f"shape_{hidden}x{width}_d{d}", sysm, chain, sob(width + hidden), None
    # quadrature-weighted terms ride along unchanged (weights are per-point factors of the residual)
    sysm, chain = poisson2d(npde, "tanh", width=16, hidden=3)
    yield "poisson2d_3x16", sysm, chain, npde.GridTraining(0.125), [0.5, 1.0, 2.0, 1.0, 3.0]


def test_persistent_training_kernel_over_kernel_shapes(npde, use_emu, monkeypatch):
    """Loop and kernel bit for bit over jet sets (value-only rows, full Hessians, pure third / fourth derivatives), widths 8-32, 1-3 hidden
    layers, 1-3 inputs, tanh / sigmoid, affine and non-affine residuals, per-term weights — whatever launch shape the planner picks."""
    monkeypatch.setenv("PINN_TRAIN_GENERAL", "1")
    ran = []
    for name, sysm, chain, strat, w in _shape_cases(npde):
        th0 = theta_for(chain, 77)
        rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th0, precision="f32"))
        eng = rep.engine
        wts = None if w is None else np.asarray(w, dtype=np.float32)
        a = _run(npde, eng, th0, wts, False, steps=(5, 3))
        b = _run(npde, eng, th0, wts, True, steps=(5, 3))
        ran.append((name, b[4]))
        if b[4] != "persistent":                   # (several launch groups: the planner did not fold the terms into one — the loop, by design)
            assert len([l for l in eng.describe().splitlines() if l.startswith("group")]) > 1, (name, eng.describe())
            continue
        for x, y, what in zip(a[:4], b[:4], ("theta", "history", "theta after resume", "history after resume")):
            assert np.array_equal(x, y), (name, what)
    assert sum(p == "persistent" for _, p in ran) >= 4, ran


def test_single_evaluation_in_one_launch_equals_the_stand_alone_kernels(npde, use_emu, monkeypatch):
    """pinn_loss_grad / pinn_lbfgs on a small problem: residual kernel + grid barrier + fixed-order sums in ONE launch (the training kernel's
    evaluation-only mode) instead of residual kernel + reduction kernel — term losses and gradient bit for bit, also with an estimated PDE
    parameter in theta; a caller that asks for HIP events (set_timing) keeps the stand-alone kernels."""
    from neuralpde_jl_amd import workloads
    wl = workloads.cfg1_poisson1d(600)
    cases = [("cfg1", wl.pde_system, wl.chains[0], wl.strategy, wl.theta, None)]
    sysm, chain = poisson2d(npde, "sigmoid", width=16, hidden=2)
    cases.append(("poisson2d", sysm, chain, npde.GridTraining(0.1), theta_for(chain, 9), [1.0, 2.0, 1.0, 3.0, 1.0]))
    taken = []
    for name, sysm, chain, strat, th0, w in cases:
        rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, strat, init_params=th0, precision="f32"))
        eng = rep.engine
        wts = None if w is None else np.asarray(w, dtype=np.float32)
        monkeypatch.setenv("PINN_NO_FUSED_EVAL", "1")
        l0, g0 = eng.loss_grad(th0, wts)
        monkeypatch.delenv("PINN_NO_FUSED_EVAL")
        assert eng.get_option("eval_path") == "stand-alone kernels"
        l1, g1 = eng.loss_grad(th0, wts)
        taken.append(eng.get_option("eval_path"))
        l2, g2 = eng.loss_grad(th0 * 1.01, wts)                   # (the barrier counter runs on from launch to launch)
        l3, g3 = eng.loss_grad(th0, wts)
        assert np.array_equal(l0, l1) and np.array_equal(g0, g1), name
        assert np.array_equal(l0, l3) and np.array_equal(g0, g3) and not np.array_equal(g0, g2), name
        theta, hist = eng.lbfgs(th0, 5, wts)
        monkeypatch.setenv("PINN_NO_FUSED_EVAL", "1")
        theta_b, hist_b = eng.lbfgs(th0, 5, wts)
        monkeypatch.delenv("PINN_NO_FUSED_EVAL")
        assert np.array_equal(theta, theta_b) and np.array_equal(hist, hist_b), name
    assert "one launch" in taken, taken
    eng.set_timing(1, -1)                                  # events requested: the stand-alone kernels they bracket
    eng.loss_grad(th0, wts)
    assert eng.get_option("eval_path") == "stand-alone kernels"


def test_hip_events_are_opt_in(npde, use_emu):
    """pinn_set_timing: no HIP events around an evaluation's kernels unless the caller asks (they cost 25 us per host-entry call on the
    hardware): pinn_last_timing refuses until the phase events are switched on, and a caller that switches them on gets the stand-alone
    kernels those events bracket."""
    sysm, chain = poisson2d(npde, "tanh", width=16, hidden=2)
    th0 = theta_for(chain, 4)
    eng = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.2), init_params=th0, precision="f32")).engine
    eng.loss_grad(th0)
    with pytest.raises(Exception, match="pinn_set_timing"):
        eng.last_timing()
    eng.set_timing(2, -1)
    eng.loss_grad(th0)
    k, t = eng.last_timing()
    assert k >= 0.0 and t >= 0.0 and eng.get_option("eval_path") == "stand-alone kernels"
    eng.set_timing(0, -1)
    eng.loss_grad(th0)
    assert eng.get_option("eval_path") == "one launch"


def test_one_launch_evaluation_checks_points_and_data(npde, use_emu):
    """the one-launch evaluation (eval_and_sync -> eval_fused) skipped ensure_points — a handle with point sets installed for
    SOME terms returned NaN losses and a finite gradient instead of the error the stand-alone kernels report.  Every path checks now."""
    import os
    sysm, chain = poisson2d(npde, "tanh", width=16, hidden=2)
    th0 = theta_for(chain, 4)
    rep = npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(chain, npde.GridTraining(0.2), init_params=th0, precision="f32"))
    rep.engine.loss_grad(th0)
    assert rep.engine.get_option("eval_path") == "one launch"             # (the shape IS eligible for the one-launch evaluation)
    desc = rep.ir.to_descriptor()
    sets = rep.pde_train_sets + rep.bcs_train_sets
    for env in (None, "1"):
        if env:
            os.environ["PINN_NO_FUSED_EVAL"] = env
        try:
            eng = npde.Engine(desc)
            eng.set_points(0, sets[0])                                         # term 0 only
            with pytest.raises(Exception, match="term 1 has no collocation points"):
                eng.loss_grad(th0)
            for k in range(1, eng.K):
                eng.set_points(k, sets[k])
            l, g = eng.loss_grad(th0)
            assert np.all(np.isfinite(l)) and np.all(np.isfinite(g))
        finally:
            os.environ.pop("PINN_NO_FUSED_EVAL", None)


def test_gemm_auto_measures_when_to_leave_the_split_products(npde, use_emu):
    """pinn_set_option(h, "gemm", "auto") (r06; VERDICT r05 weak #3: when should a caller leave the split-bf16 products?): the engine evaluates
    the gradient with BOTH GEMM arithmetics at the current parameters — delta = |grad(split) - grad(fp32)| / |grad(fp32)| is the split products'
    arithmetic error there — and keeps "split" while delta <= 1e-5, runs "fp32" above.  At glorot parameters delta ~ 1e-7: split stays; at
    the committed trained parameters (cfg2 after 6,000 Adam steps) delta ~ 1e-3: the fp32 MFMAs take over — and ARE the more accurate ones
    there against the float64 oracle; a resident Adam chunk runs the check at its end without disturbing the optimiser state."""
    import os
    import helpers
    import pinn_oracle as po
    from neuralpde_jl_amd import workloads
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg2_variants.npz"))
    wl = workloads.cfg2_poisson2d(points=256, bcs_points=64)
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    assert eng.get_option("gemm") == "split" and float(eng.get_option("gemm_delta")) == -1.0
    eng.set_option("gemm", "auto")
    assert eng.get_option("gemm") == "auto(split)"
    th0 = np.asarray(rep.flat_init_params, dtype=np.float32)
    l0, g0 = eng.loss_grad(th0)
    rho0, d0 = float(eng.get_option("grad_health")), float(eng.get_option("gemm_delta"))
    assert abs(rho0 - np.linalg.norm(g0.astype(np.float64)) / np.sqrt(l0.sum())) < 1e-4 * rho0
    assert 0.0 < d0 < 3e-6 and eng.get_option("gemm") == "auto(split)"          # initialisation: the fast products stay
    eng.set_option("gemm", "auto")                                              # (re-arms the check: it runs at most once per 1,000 evaluations)
    th = g["theta_adam6000"].astype(np.float32)
    l1, g1 = eng.loss_grad(th)                                                  # this evaluation ran on the split products; the check followed it
    rho1, d1 = float(eng.get_option("grad_health")), float(eng.get_option("gemm_delta"))
    assert d1 > 1e-5 and eng.get_option("gemm") == "auto(fp32)" and rho1 < rho0 * 0.05
    l2, g2 = eng.loss_grad(th)                                                  # ... the next one runs the fp32 MFMAs (no new check: cadence)
    assert eng.get_option("gemm") == "auto(fp32)" and float(eng.get_option("gemm_delta")) == d1
    sets = rep.pde_train_sets + rep.bcs_train_sets
    prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
    ref = po.loss_and_grad(prob, th.astype(np.float64), sets, mode="exact")     # (oracle at float32(theta): the arithmetic alone)
    e_split = np.linalg.norm(g1 - ref.grad) / np.linalg.norm(ref.grad)
    e_fp32 = np.linalg.norm(g2 - ref.grad) / np.linalg.norm(ref.grad)
    print(f"\ncfg2 adam6000 (256 + 4 x 64 points): delta {d1:.2e}, rho {rho1:.3g} (initialisation: delta {d0:.2e}, rho {rho0:.3g}); split vs oracle {e_split:.2e}, fp32 MFMA vs oracle {e_fp32:.2e}")
    assert e_fp32 < e_split
    # a resident Adam chunk runs the check at its end (loop path); the optimiser state survives the re-plan
    eng.set_option("persistent", "off")
    eng.set_option("gemm", "split")
    eng.set_option("gemm", "auto")
    tha, hist = eng.adam(th, 3, 1e-8)                                           # (tiny steps: the iterate stays at the trained point)
    assert eng.get_option("gemm") == "auto(fp32)" and np.all(np.isfinite(hist))
    thb, hist2 = eng.adam(None, 2, 1e-8, init=False)
    assert np.all(np.isfinite(hist2)) and hist2[0] <= hist[0] * 1.5
    eng.set_option("gemm", "split")                                             # an explicit mode ends the policy
    assert eng.get_option("gemm") == "split"
    eng.loss_grad(th)
    assert eng.get_option("gemm") == "split"
    with pytest.raises(Exception, match="split.*fp32.*auto"):
        eng.set_option("gemm", "bf16")
