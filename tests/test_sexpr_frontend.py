"""The symbolic front end of the descriptor ("pinnir 2", csrc/sexpr.cpp): equations handed to the library as the s-expressions of
`toexpr(expand_derivatives(eq.lhs / eq.rhs))` — what the Julia glue emits (julia/NeuralPDEHIP.jl) — and lowered INSIDE the library
(the engine's restatement of `_transform_expression`, src/symbolic_utilities.jl:132-331).
(a) the CPU parity tests re-run through this front end (PINN_DESCRIPTOR=2) against the float64 oracle;
(b) both front ends build the same kernel plan; (c) Julia-style spellings (nested Differentials, unary minus, n-ary calls, rationals,
`inv`, `sqrt`) and the loud failures."""
import numpy as np
import pytest

import test_emu_parity as tp
import test_reference_examples as tr


@pytest.mark.parametrize("mod,name", [
    (tp, "test_small_poisson_tanh_and_sigmoid"), (tp, "test_cfg1_grid_3x32"), (tp, "test_cfg3_burgers_4x64_small"),
    (tp, "test_param_estim_gradient"), (tp, "test_coupled_system_of_pdes"), (tp, "test_hoisted_sources_and_mixed_ops"),
    (tp, "test_third_and_fourth_order_derivatives"), (tp, "test_kuramoto_sivashinsky_jets"), (tp, "test_heterogeneous_system"),
    (tp, "test_forward_laplacian_fusion"), (tr, "test_wave_equation"), (tr, "test_mixed_derivative_pde"),
    (tr, "test_nonlinear_elliptic_first_order_system"), (tr, "test_lorenz_parameter_estimation_terms"),
    (tr, "test_data_misfit_terms_on_device")])
def test_parity_suite_through_the_symbolic_front_end(npde, use_emu, monkeypatch, mod, name):
    monkeypatch.setenv("PINN_DESCRIPTOR", "2")
    created = []
    orig = npde._lib.Engine.__init__

    def spy(self, descriptor, *a, **k):
        created.append(descriptor.split("\n", 1)[0])
        orig(self, descriptor, *a, **k)

    monkeypatch.setattr(npde._lib.Engine, "__init__", spy)
    getattr(mod, name)(npde, None)
    assert created and all(c == "pinnir 2" for c in created)


def test_both_front_ends_build_the_same_plan(npde, use_emu, monkeypatch):
    from neuralpde_jl_amd import workloads
    for wl in (workloads.cfg2_poisson2d(points=64, bcs_points=16), workloads.cfg3_burgers(points=64, bcs_points=16),
               workloads.cfg4_cavity(points=32, bcs_points=16, width=64, hidden=4), workloads.cfg5_heat_inverse(points=32, bcs_points=16)):
        monkeypatch.delenv("PINN_DESCRIPTOR", raising=False)
        r1 = npde.symbolic_discretize(wl.pde_system, wl.discretization())
        monkeypatch.setenv("PINN_DESCRIPTOR", "2")
        r2 = npde.symbolic_discretize(wl.pde_system, wl.discretization())
        kern = lambda rep: [l.split("kernel=")[1].split()[0] for l in rep.engine.describe().splitlines() if "kernel=" in l]
        assert kern(r1) == kern(r2), wl.name                                    # same kernels (incl. the forward-Laplacian fusion)
        th = r1.flat_init_params
        for k, s in enumerate(r1.pde_train_sets + r1.bcs_train_sets):          # identical points (cfg5 draws randomly)
            r2.engine.set_points(k, s)
        l1, g1 = r1.engine.loss_grad(th)
        l2, g2 = r2.engine.loss_grad(th)
        np.testing.assert_allclose(l2, l1, rtol=2e-6)
        assert np.linalg.norm(g2 - g1) / np.linalg.norm(g1) < 2e-6, wl.name


def _engine(npde, lhs, rhs, nets=(("u", ("x", "y")),), indvars=("x", "y"), params=(), width=16):
    lines = ["pinnir 2", "ntheta %d" % (len(nets) * (len(nets[0][1]) * width + width + width * width + width + width + 1) + len(params)),
             f"params {len(params)} 0 0", "defaults " + " ".join("0.5" for _ in params), "pnames " + " ".join(params), f"nets {len(nets)}"]
    off = 0
    for i, (name, inputs) in enumerate(nets):
        d = len(inputs)
        lines += [f"net {i} tanh {off} 4 {d} {width} {width} 1", f"netvar {i} {name} {d} " + " ".join(inputs)]
        off += d * width + width + width * width + width + width + 1
    lines += ["terms 1", f"sterm 0 {len(indvars)} " + " ".join(indvars), "lhs " + lhs, "rhs " + rhs]
    return npde.Engine("\n".join(lines) + "\n")


def test_julia_style_spellings_and_errors(npde, use_emu):
    rng = np.random.default_rng(0)
    pts = rng.uniform(0.1, 0.9, size=(2, 40))

    def resid(lhs, rhs, **kw):
        e = _engine(npde, lhs, rhs, **kw)
        th = np.random.default_rng(1).uniform(-0.5, 0.5, e.P)
        e.set_points(0, pts[: len(kw.get("indvars", ("x", "y")))])
        return e.residual(0, th, pts.shape[1])

    base = resid("(+ (D x 2 (u x y)) (D y 2 (u x y)))", "(* -1 (sin (* pi x)) (sin (* pi y)))")
    # nested first-order Differentials (how Symbolics nests Dx(Dx(u))), binary minus, unary minus, n-ary product with the sign inside
    alt = resid("(+ (D x 1 (D x 1 (u x y))) (D y 1 (D y 1 (u x y))))", "(- (* (sin (* pi x)) (sin (* y pi))))")
    np.testing.assert_allclose(alt, base, rtol=0, atol=2e-6)
    alt = resid("(- (+ (D x 2 (u x y)) (D y 2 (u x y))) (* -1 (sin (* pi x)) (sin (* pi y))))", "0")
    np.testing.assert_allclose(alt, base, rtol=0, atol=2e-6)
    # mixed derivative in either nesting order is ONE slot; rationals, inv, sqrt, integer and real powers
    a = resid("(+ (D x 1 (D y 1 (u x y))) (* 1//2 (^ (u x y) 2)) (inv (+ 2 x)) (sqrt (+ 1 y)))", "(^ (+ 1 x) 1.5)")
    b = resid("(+ (D y 1 (D x 1 (u x y))) (* 0.5 (u x y) (u x y)) (/ 1 (+ x 2)) (^ (+ y 1) 0.5))", "(* (+ 1 x) (sqrt (+ 1 x)))")
    np.testing.assert_allclose(a, b, rtol=0, atol=5e-6)
    # the call arguments of a dependent variable are dropped (symbolic_utilities.jl:145-160): u(0, y) reads the point set
    np.testing.assert_array_equal(resid("(u 0 y)", "0"), resid("(u x y)", "0"))
    for lhs, msg in [("(D z 1 (u x y))", "not one of its inputs"), ("(D x 1 (sin x))", "expand_derivatives"),
                     ("(D x 4 (D y 3 (u x y)))", "derivative order 7 > 6"), ("(gamma (u x y))", "closed op set"),
                     ("(+ (u x y) w)", "neither an independent variable"), ("(+ (u x y)", "missing '\\)'"), ("(sin x)", "does not contain a dependent variable")]:
        with pytest.raises(Exception, match=msg):
            _engine(npde, lhs, "0")
    # numeric literals are decimal literals only: a symbol that strtod would read as a number ("inf", "nan", hex) stays a symbol, so a
    # parameter of that name is a parameter and an unknown one is an error — never silently a constant; derivative orders are integers
    a = resid("(+ (u x y) nan)", "0", params=("nan",))
    b = resid("(+ (u x y) q)", "0", params=("q",))
    np.testing.assert_array_equal(a, b)
    assert np.all(np.isfinite(a))
    np.testing.assert_allclose(resid("(+ (u x y) 1.5f0)", "2e-1"), resid("(+ (u x y) 1.3)", "0"), rtol=0, atol=1e-6)      # Julia Float32 spelling
    for lhs, msg in [("(+ (u x y) inf)", "neither an independent variable"), ("(+ (u x y) 0x10)", "neither an independent variable"),
                     ("(D x 2.5 (u x y))", "derivative order must be an integer"), ("(D x 1e1 (u x y))", "derivative order must be an integer"),
                     ("(+ 1 " * 300 + "(u x y)" + ")" * 300, "nested deeper")]:
        with pytest.raises(Exception, match=msg):
            _engine(npde, lhs, "0")


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_golden_descriptors(npde, use_emu, cfg):
    """tests/golden/descriptors/<cfg>.pinnir2 — the descriptors of the five BASELINE configurations in exactly the format the Julia glue's
    `descriptor(pinnrep)` emits (julia/NeuralPDEHIP.jl) — are what the Python mirror prints today, create an engine, and give the same
    residuals as the tape form (<cfg>.pinnir1) of the same problem."""
    import os
    from neuralpde_jl_amd import workloads
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "descriptors")
    mk = {"cfg1": lambda: workloads.cfg1_poisson1d(1024), "cfg2": lambda: workloads.cfg2_poisson2d(points=256, bcs_points=64),
          "cfg3": lambda: workloads.cfg3_burgers(points=256, bcs_points=64), "cfg4": lambda: workloads.cfg4_cavity(points=256, bcs_points=64),
          "cfg5": lambda: workloads.cfg5_heat_inverse(points=256, bcs_points=64)}
    wl = mk[cfg]()
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    d2 = open(os.path.join(root, cfg + ".pinnir2")).read()
    d1 = open(os.path.join(root, cfg + ".pinnir1")).read()
    assert rep.ir.to_descriptor2() == d2 and rep.ir.to_descriptor() == d1
    e1, e2 = npde.Engine(d1), npde.Engine(d2)
    th = rep.flat_init_params
    sets = rep.pde_train_sets + rep.bcs_train_sets
    for k, s in enumerate(sets):
        e1.set_points(k, s)
        e2.set_points(k, s)
    for k in (0, len(sets) - 1):
        r1, r2 = e1.residual(k, th, sets[k].shape[1]), e2.residual(k, th, sets[k].shape[1])
        assert np.max(np.abs(r1 - r2)) <= 2e-6 * max(1.0, np.max(np.abs(r1)))
