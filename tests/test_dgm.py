"""The Deep Galerkin architecture (src/dgm.jl:40-48, 97-115, 143-152) through the engine's third kernel family (csrc/pinn_kernels3.hpp):
gated layers Z, G, R, H with element-wise products of Taylor jets (Leibniz rule), hand-derived reverse sweep, weight gradients by a second
contraction kernel.  Loss and gradient against the float64 oracle's torch restatement of the same architecture; CPU through the
emulation build, GPU through tests/test_gpu_mirror.py."""
import numpy as np
import pytest
import sympy as sp

import helpers
import pinn_oracle as po
import test_emu_parity as tp


def _burgers(npde, nu=0.05):
    # the PDE of test/DGM/dgm__burger_s_equation.jl: u_t + u u_x - nu u_xx = 0 on (t, x) in [0, 1] x [-1, 1]
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    Dt, Dx, Dxx = npde.Differential(t), npde.Differential(x), npde.Differential(x) ** 2
    eq = npde.Eq(Dt(u(t, x)) + u(t, x) * Dx(u(t, x)) - nu * Dxx(u(t, x)), 0)
    bcs = [npde.Eq(u(0, x), -sp.sin(sp.pi * x)), npde.Eq(u(t, -1), 0.0), npde.Eq(u(t, 1), 0.0)]
    dom = [npde.In(t, npde.Interval(0.0, 1.0)), npde.In(x, npde.Interval(-1.0, 1.0))]
    return npde.PDESystem([eq], bcs, dom, [t, x], [u(t, x)])


@pytest.mark.parametrize("modes,layers,a1,a2", [(30, 3, "tanh", "tanh"), (12, 2, "sigmoid", "tanh"), (20, 1, "tanh", "sin")])
def test_dgm_burgers_parity(npde, use_emu, modes, layers, a1, a2):
    sysm = _burgers(npde)
    net = npde.DGM(2, 1, modes, layers, a1, a2, "identity")
    th = tp.theta_for(net, 100 + modes)
    assert th.size == net.nparams
    strat = npde.QuasiRandomTraining(70, bcs_points=20, sampling_alg=npde.SobolSample(seed=3), resampling=False, minibatch=1)
    rep, prob, sets, th = tp.check(npde, sysm, [net], strat, th, weights=[1.0, 2.0, 0.5, 3.0], mode="exact")
    kernels = [l.split("kernel=")[1].split()[0] for l in rep.engine.describe().splitlines() if "kernel=" in l]
    assert all(k.startswith("F3_") for k in kernels) and any("(C=4)" in k for k in kernels) and any("(C=1)" in k for k in kernels)
    # the reference's own algorithm (central differences through the network) agrees too
    ref = po.loss_and_grad(prob, th, sets, weights=[1.0, 2.0, 0.5, 3.0], mode="stencil")
    losses, grad = rep.engine.loss_grad(th, [1.0, 2.0, 0.5, 3.0])
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < 1e-5 and g2 < 1e-5 and gi < 1e-5
    # loss-only evaluation (MODE_LOSS of the DGM family): the same term losses, no gradient
    l_only, g_only = rep.engine.loss_grad(th, [1.0, 2.0, 0.5, 3.0], want_grad=False)
    assert g_only is None
    np.testing.assert_allclose(l_only, losses, rtol=1e-13, atol=0)
    # trial function, residual and pointwise derivatives
    pts = sets[0][:, :33]
    assert np.max(np.abs(rep.phi(pts, th)[0] - po.phi_values(prob.chains[0], th, pts)[0])) < 1e-5
    r = rep.engine.residual(0, th, sets[0].shape[1])
    r_ref = po.residual_values(prob, th, 0, sets[0], mode="exact")[0]
    assert np.max(np.abs(r - r_ref)) < 2e-5 * max(1.0, np.max(np.abs(r_ref)))
    import torch
    uu = lambda cord, t_, phi: phi(cord, t_).sum(dim=0, keepdim=True)
    ex = po.exact_derivative(prob.chains[0], uu, torch.tensor(pts, dtype=po.DT), [1, 1], torch.tensor(th, dtype=po.DT)).detach().numpy().reshape(-1)
    assert np.max(np.abs(rep.engine.derivative(0, th, pts, [1, 1]) - ex)) < 2e-5 * max(1.0, np.max(np.abs(ex)))


def test_deep_galerkin_constructor_trains_and_misuse(npde, use_emu):
    """DeepGalerkin(...) = PhysicsInformedNN over DGM (src/dgm.jl:143-152); the resident Adam loop lowers the loss; mixed third derivative and
    parameter estimation run through the same family; unsupported set-ups fail loudly."""
    sysm = _burgers(npde)
    strat = npde.QuasiRandomTraining(64, bcs_points=16, sampling_alg=npde.SobolSample(seed=5), resampling=False, minibatch=1)
    disc = npde.DeepGalerkin(2, 1, 8, 2, "tanh", "tanh", "identity", strat, precision="f32")      # (DGM networks run on the fp32 kernels: the explicit opt-in; "auto" + Float64 parameters fails with the reason, below)
    prob = npde.discretize(sysm, disc)
    assert prob.u0.size == npde.DGM(2, 1, 8, 2).nparams
    res = npde.solve(prob, npde.Adam(0.01), maxiters=30)
    assert np.all(np.isfinite(res.losses)) and res.losses[-1] < res.losses[0]
    # inverse problem: nu estimated (theta.p gradient through the one-lane-per-point tape), plus a mixed third derivative in the residual
    t, x = npde.parameters("t x")
    (u,) = npde.variables("u")
    (nu,) = npde.parameters("nu")
    Dt, Dx = npde.Differential(t), npde.Differential(x)
    eq = npde.Eq(Dt(u(t, x)) + u(t, x) * Dx(u(t, x)) - nu * (Dx ** 2)(u(t, x)) + 0.1 * Dt(Dx(Dx(u(t, x)))), 0)
    sys2 = npde.PDESystem([eq], [npde.Eq(u(0, x), -sp.sin(sp.pi * x))], list(sysm.domain), [t, x], [u(t, x)], ps=[nu], defaults={nu: 0.05})
    net = npde.DGM(2, 1, 8, 1, "tanh", "tanh")
    tp.check(npde, sys2, [net], strat, tp.theta_for(net, 7), param_estim=True, mode="exact")
    with pytest.raises(ValueError, match="single output"):
        npde.DGM(2, 2, 8, 1)
    with pytest.raises(ValueError, match="identity"):
        npde.DGM(2, 1, 8, 1, "tanh", "tanh", "tanh")
    # a DGM network inside an equation that couples two dependent variables is refused at create time
    (v,) = npde.variables("v")
    sys3 = npde.PDESystem([npde.Eq(Dt(u(t, x)) + v(t, x), 0), npde.Eq(Dx(v(t, x)) - u(t, x), 0)],
                          [npde.Eq(u(0, x), 0.0), npde.Eq(v(0, x), 0.0)], list(sysm.domain), [t, x], [u(t, x), v(t, x)])
    with pytest.raises(npde.EngineError, match="single-network equations"):
        npde.symbolic_discretize(sys3, npde.PhysicsInformedNN([npde.DGM(2, 1, 8, 1), npde.DGM(2, 1, 8, 1)], strat, precision="f32"))
