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
