"""Pins the oracle (oracle/pinn_oracle.py) against every numeric pin the reference's own tests/docs hold for this path
(SURVEY.md §8c).  Beyond these the reference has no golden loss/gradient vectors: parity is otherwise unpinned."""
import numpy as np
import torch

import pinn_oracle as po


def test_get_eps_float32_order1_matches_debugging_doc():
    # docs/src/developer/debugging.md:64-65 prints 0.0049215667 for the Float32 first-order epsilon
    e = po.get_eps(2, 1, np.float32, 1)
    assert e.dtype == np.float32 and e[1] == 0
    assert abs(float(e[0]) - 0.0049215667) < 1e-9


def test_get_eps_float64_values():
    # eps(Float64)^(1/(2+order)), src/symbolic_utilities.jl:98-103
    for order, val in [(1, 6.0554544523933395e-06), (2, 1.220703125e-04), (3, 7.4009597974140505e-04), (4, 2.4607833005707410e-03)]:
        assert abs(po.get_eps(3, 2, np.float64, order)[1] - val) < 1e-15 * max(1.0, val / 1e-16) * 1e-3 + 1e-18


def test_forward_derivatives_fd_vs_exact():
    # restates test/Forward/forward__derivatives.jl:7-44: 2->16->16->1 sigmoid chain at [1, 2];
    # first order atol 1e-8, second order (xx, xy, yy) atol 4e-5
    chain = po.Chain((2, 16, 16, 1), "sigmoid")
    u = lambda cord, th, phi: phi(cord, th).sum(dim=0, keepdim=True)
    x = torch.tensor([[1.0], [2.0]], dtype=po.DT)
    for seed in range(5):
        theta = torch.tensor(po.glorot_theta(chain, np.random.default_rng(seed), bias_amp=0.0), dtype=po.DT)
        for ax in (0, 1):
            fd = po.numeric_derivative(chain, u, x, [po.get_eps(2, ax + 1, np.float64, 1)], 1, theta)
            ex = po.exact_derivative(chain, u, x, [ax], theta)
            assert abs(float(fd) - float(ex)) < 1e-8
        ex_, ey_ = po.get_eps(2, 1, np.float64, 2), po.get_eps(2, 2, np.float64, 2)
        for epss, axes in [([ex_, ex_], [0, 0]), ([ex_, ey_], [0, 1]), ([ey_, ey_], [1, 1])]:
            fd = po.numeric_derivative(chain, u, x, epss, 2, theta)
            ex = po.exact_derivative(chain, u, x, axes, theta)
            assert abs(float(fd) - float(ex)) < 4e-5


def test_forward_ode_golden(npde):
    # restates test/Forward/forward__ode.jl:10-47: parameter-free chain x -> x.^2, Dx(u(x)) ~ 0, GridTraining(0.1) on
    # [0, 1]: the datafree pde loss function returns 2x on the training set, rtol 1e-8
    import sympy as sp
    (x,) = npde.parameters("x")
    (u,) = npde.variables("u")
    eq = npde.Eq(npde.Differential(x)(u(x)), 0.0)
    bcs = [npde.Eq(u(0.0), u(0.0))]
    sysm = npde.PDESystem([eq], bcs, [npde.In(x, npde.Interval(0.0, 1.0))], [x], [u(x)])
    vi = npde.get_vars(sysm.ivs, sysm.dvs)
    pde_sets, bc_sets = npde.generate_training_sets(sysm.domain, 0.1, sysm.eqs, sysm.bcs, np.float64, vi)
    train = pde_sets[0]
    assert train.shape[0] == 1 and train.shape[1] >= 10
    phi = lambda cord, th: cord ** 2
    cord = torch.tensor(train, dtype=po.DT)
    r = po.numeric_derivative(phi, po.get_u(), cord, [po.get_eps(1, 1, np.float64, 1)], 1, None) - 0.0
    np.testing.assert_allclose(r.numpy(), 2 * train, rtol=1e-8)


def test_numeric_derivative_orders_3_4_and_mixed_formulas():
    # src/pinn_types.jl:454-474 on an analytic function: u = sin(x) * exp(y)
    phi = lambda cord, th: torch.sin(cord[0:1]) * torch.exp(cord[1:2])
    x = torch.tensor([[0.3, 1.1], [0.2, -0.4]], dtype=po.DT)
    u = po.get_u()
    e3 = po.get_eps(2, 1, np.float64, 3)
    d3 = po.numeric_derivative(phi, u, x, [e3, e3, e3], 3, None)
    np.testing.assert_allclose(d3.numpy(), (-torch.cos(x[0:1]) * torch.exp(x[1:2])).numpy(), rtol=1e-5)
    e4 = po.get_eps(2, 1, np.float64, 4)
    d4 = po.numeric_derivative(phi, u, x, [e4] * 4, 4, None)
    np.testing.assert_allclose(d4.numpy(), (torch.sin(x[0:1]) * torch.exp(x[1:2])).numpy(), rtol=1e-4)
    ex_, ey_ = po.get_eps(2, 1, np.float64, 2), po.get_eps(2, 2, np.float64, 2)
    dxy = po.numeric_derivative(phi, u, x, [ex_, ey_], 2, None)
    np.testing.assert_allclose(dxy.numpy(), (torch.cos(x[0:1]) * torch.exp(x[1:2])).numpy(), rtol=1e-6)


def test_lux_dense_layout_and_phi():
    # [3P] Lux Dense / ComponentArrays flat layout: [W (out x in, column-major) | b | ...]
    chain = po.Chain((2, 3, 1), "tanh")
    theta = np.arange(1, chain.nparams + 1, dtype=np.float64) / 10
    W1 = theta[:6].reshape(2, 3).T
    b1 = theta[6:9]
    W2 = theta[9:12].reshape(3, 1).T
    b2 = theta[12:13]
    x = np.array([[0.5], [-1.0]])
    ref = W2 @ np.tanh(W1 @ x + b1[:, None]) + b2[:, None]
    np.testing.assert_allclose(po.phi_values(chain, theta, x), ref, rtol=1e-14)


def test_golden_fixtures_reproduce(npde):
    """The committed fixtures (oracle/make_golden.py) are what the oracle computes today."""
    import glob, os
    import helpers
    from neuralpde_jl_amd import workloads
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    makers = {"cfg1_poisson1d_1024": lambda: workloads.cfg1_poisson1d(1024),
              "cfg2_poisson2d_512": lambda: workloads.cfg2_poisson2d(points=512, bcs_points=128),
              "cfg3_burgers_512": lambda: workloads.cfg3_burgers(points=512, bcs_points=128)}
    files = [os.path.join(root, n + ".npz") for n in sorted(makers)]
    for f in files:
        g = np.load(f)
        wl = makers[os.path.basename(f)[:-4]]()
        sets = [g[f"set{k}"] for k in range(int(g["nsets"]))]
        prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
        ev = po.loss_and_grad(prob, g["theta"], sets, weights=g["weights"], mode="stencil")
        np.testing.assert_allclose(ev.term_losses, g["losses_stencil"], rtol=1e-12)
        np.testing.assert_allclose(ev.grad, g["grad_stencil"], rtol=1e-9, atol=1e-14)
        # the two oracle modes (reference FD semantics vs exact derivatives) agree far inside the 1e-5 parity bar
        assert np.linalg.norm(g["grad_stencil"] - g["grad_exact"]) / np.linalg.norm(g["grad_exact"]) < 1e-6
        assert np.max(np.abs(g["losses_stencil"] - g["losses_exact"]) / g["losses_exact"]) < 1e-6


def test_full_size_fixtures_reproduce(npde):
    """The full-size fixtures (`python oracle/make_golden.py full`): the regenerated point sets have the pinned SHA-256 digests, and
    the benchmarked configuration's fixture (cfg2_full: 65,536 + 4 x 65,536 points) is what the oracle computes today."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import helpers
    import make_golden as mg
    for name in ("cfg2_full", "cfg3_full"):
        g = np.load(os.path.join(root, "tests", "golden", name + ".npz"))
        wl = mg.FULL_CASES[name][0]()
        sets = mg.point_sets(wl)
        assert [mg.set_digest(s) for s in sets] == list(g["set_sha256"])
        if name != "cfg2_full":
            continue
        prob = helpers.oracle_problem(npde, wl.pde_system, wl.chains)
        losses, grad = mg.chunked_loss_and_grad(prob, g["theta"], sets, g["weights"], chunk=int(g["chunk"]))
        np.testing.assert_allclose(losses, g["losses_stencil"], rtol=1e-12)
        np.testing.assert_allclose(grad, g["grad_stencil"], rtol=1e-9, atol=1e-13)
