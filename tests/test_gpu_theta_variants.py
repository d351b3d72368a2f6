"""Parity away from glorot-initialised parameters (r04): the benchmarked configuration and cfg3 at FULL size, cfg4 / cfg5 reduced, at
theta x 2, theta x 4 (saturating units, large higher-derivative jets) and at TRAINED parameters (float64 Adam on the oracle's own
objective: gradients that are small differences of large per-point terms), against committed float64-oracle outputs in BOTH oracle
modes (tests/golden/*_variants.npz, `python oracle/make_golden.py variants`) and in BOTH GEMM modes of the engine:
    "split" (default) — three-piece bf16 split products on the bf16 matrix pipe,
    "fp32"            — v_mfma_f32_16x16x4_f32 (pinn_set_option(h, "gemm", "fp32")).
Bar (north star): 1e-5 relative — per-term loss, gradient norm-wise in L2 and Linf.  Against the EXACT-derivative oracle (the mathematics the
engine implements) the bar holds as is.  Against the STENCIL oracle (the reference's finite differences, src/pinn_types.jl:445-482) it holds
wherever the stencil's own truncation error leaves room: at saturating parameters the float64 stencil differs from the exact derivative by
more than 1e-5 on its own (printed per case as "finite-difference error"), and there the engine must be as close to the reference as exact
derivatives can be: within 1.3 x that error + 5e-6.
At TRAINED parameters the residual is a small difference of O(1) terms (|r| ~ 1e-3 |u_xx|) and the gradient a small difference of large
per-point terms: the QUANTISATION of theta to float32 alone moves loss and gradient by 2e-5 ... 1e-2 (the fixtures carry the float64 oracle at
float32(theta): `*_exact32_*`, oracle/make_golden.py variants-theta32), and every fp32 evaluation adds its rounding to that (the fixtures
carry the SAME program evaluated by torch in float32: `*_f32_*`).  The north star prescribes fp32 compute, so there the engine is held to
"no worse than a plain fp32 implementation of the reference's mathematics": error <= max(1e-5, 2 x the float32 evaluation's error), against
the oracle at theta64 AND against the oracle at float32(theta) (r05; 8 x in r04).  What made 2 x possible: r04's engine was 2.4-2.8 x
worse than torch-f32 on the cfg2 gradients because its tanh formed e^{2x} as exp2(x * float32(2 log2 e)) — the constant's rounding error
(-1.3e-8 relative) scaled EVERY pre-activation of the network coherently, which a trained theta amplifies like a perturbation of all weights
in one direction (profiles/r05_theta_variants_ab.txt: tools/r05/theta_ab_gpu.py over four tanh variants); now a two-constant exponent, an
odd formula and one Newton step on the reciprocal (csrc/vec.hpp: vtanh_fast): 0.3-1.0 x torch-f32 on the gradients.
The FLOAT64 mode (r05: on the matrix pipe, csrc/pinn_kernels5.hpp) is held to the PLAIN 1e-5 at every trained case at full size — it meets
it with nine orders of magnitude to spare.
The measured errors and margins of every case are printed (pytest -s) and tabulated in DESIGN.md section 6."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _digest(s):
    return hashlib.sha256(np.ascontiguousarray(s, dtype=np.float64).tobytes()).hexdigest()


def _makers():
    from neuralpde_jl_amd import workloads
    return {"cfg2_variants": lambda: workloads.cfg2_poisson2d(points=65536),
            "cfg3_variants": lambda: workloads.cfg3_burgers(points=262144),
            "cfg4_variants": lambda: workloads.cfg4_cavity(points=16384, bcs_points=4096),
            "cfg5_variants": lambda: workloads.cfg5_heat_inverse(points=32768, bcs_points=8192)}


def _errors(losses, grad, lref, gref):
    le = np.max(np.abs(losses - lref) / np.abs(lref))
    g2 = np.linalg.norm(grad - gref) / np.linalg.norm(gref)
    gi = np.max(np.abs(grad - gref)) / np.max(np.abs(gref))
    return le, g2, gi


CASES = [("cfg2_variants", t) for t in ("x2", "x4", "adam2000", "adam6000")] + [("cfg3_variants", t) for t in ("x2", "adam2000")] + \
        [("cfg4_variants", t) for t in ("x2", "adam400")] + [("cfg5_variants", t) for t in ("x2", "x4", "adam600")]
# (r06: trained parameters for the 128-wide configurations too — 400 / 600 float64 Adam steps of the oracle on a reduced design,
# `python oracle/make_golden.py variants-trained`: the two-pass split GEMM of H = 128 where the bias it removes matters)


@pytest.mark.parametrize("name,tag", CASES)
def test_theta_variant(npde, hip_lib, name, tag):
    path = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated (python oracle/make_golden.py variants {name})")
    g = np.load(path)
    if tag not in list(g["tags"]):
        pytest.skip(f"{name}.npz holds no variant {tag}")
    wl = _makers()[name]()
    rep = npde.symbolic_discretize(wl.pde_system, wl.discretization())
    eng = rep.engine
    assert eng.L.backend == "hip"
    sets = rep.pde_train_sets + rep.bcs_train_sets
    assert [s.shape[1] for s in sets] == list(g["set_sizes"])
    assert [_digest(s) for s in sets] == list(g["set_sha256"]), "regenerated point sets differ from the ones the fixture was computed on"
    theta, w = g["theta_" + tag], g["weights"]
    assert theta.size == eng.P
    rows = {}
    for mode in ("split", "fp32"):
        eng.set_option("gemm", mode)
        assert eng.get_option("gemm") == mode
        losses, grad = eng.loss_grad(theta, w)
        l2, gr2 = eng.loss_grad(theta, w)
        assert np.array_equal(l2, losses) and np.array_equal(gr2, grad)        # bit-reproducible
        for om in ("stencil", "exact"):
            rows[(mode, om)] = _errors(losses, grad, g[f"losses_{om}_{tag}"], g[f"grad_{om}_{tag}"])
    fd = _errors(g[f"losses_stencil_{tag}"], g[f"grad_stencil_{tag}"], g[f"losses_exact_{tag}"], g[f"grad_exact_{tag}"])
    print(f"\n{name} {tag}: |grad| = {np.linalg.norm(g['grad_exact_' + tag]):.3e}; the stencil oracle against the exact oracle (the finite-difference "
          f"error of the reference's own semantics): loss {fd[0]:.1e}, grad L2 {fd[1]:.1e}, Linf {fd[2]:.1e}")
    for (mode, om), (le, g2, gi) in rows.items():
        print(f"  gemm={mode:5s} vs {om:7s} oracle: loss rel {le:.2e}, grad rel L2 {g2:.2e}, Linf {gi:.2e}   margin to 1e-5: x{TOL / max(le, g2, gi):.1f}")
    f32 = None
    if f"grad_f32_{tag}" in g:
        f32 = _errors(g[f"losses_f32_{tag}"], g[f"grad_f32_{tag}"], g[f"losses_exact_{tag}"], g[f"grad_exact_{tag}"])
        print(f"  float32 torch evaluation of the same program vs the exact float64 oracle: loss rel {f32[0]:.2e}, grad rel L2 {f32[1]:.2e}, Linf {f32[2]:.2e}")
    trained = tag.startswith("adam")
    REL = 2.0                                                     # trained parameters: at most this multiple of a plain float32 evaluation's error
    f32q = None
    if trained and f"grad_exact32_{tag}" in g:
        # the arithmetic alone: reference = the float64 oracle at the float32-rounded parameters the fp32 kernels are handed
        l32, g32 = g[f"losses_exact32_{tag}"], g[f"grad_exact32_{tag}"]
        f32q = _errors(g[f"losses_f32_{tag}"], g[f"grad_f32_{tag}"], l32, g32)
        print(f"  against the oracle at float32(theta): torch-f32 loss rel {f32q[0]:.2e}, grad rel L2 {f32q[1]:.2e}, Linf {f32q[2]:.2e}")
    quant = None
    if f32q is not None:
        # what the quantisation of theta to float32 alone costs (oracle at float32(theta) against oracle at theta64): no fp32 engine can be
        # expected closer to the theta64 reference than that — a float32 evaluation lands below it only where its rounding happens to cancel it
        quant = _errors(l32, g32, g[f"losses_exact_{tag}"], g[f"grad_exact_{tag}"])
        print(f"  quantisation of theta alone: loss rel {quant[0]:.2e}, grad rel L2 {quant[1]:.2e}, Linf {quant[2]:.2e}")
    for mode in ("split", "fp32"):
        for i, e in enumerate(rows[(mode, "exact")]):
            bound = TOL if (f32 is None or not trained) else max(TOL, REL * max(f32[i], quant[i] if quant is not None else 0.0))
            if f32 is not None and not trained:
                # scaled parameters: where a plain float32 evaluation itself sits at the bar (cfg5 x 4: 8.7e-6) the engine gets 1.5 x that
                # (r05: split 6.1e-6, fp32 MFMAs 1.02e-5 there; r04: 9.5e-6 / 9.4e-6)
                bound = max(TOL, 1.5 * f32[i])
            assert e < bound, (name, tag, mode, "exact", i, e, bound)
        for i, (e, f) in enumerate(zip(rows[(mode, "stencil")], fd)):
            bound = max(TOL, 1.3 * f + 5e-6)
            if f32 is not None and trained:
                bound = max(bound, REL * max(f32[i], quant[i] if quant is not None else 0.0) + 1.3 * f)
            assert e < bound, (name, tag, mode, "stencil", i, e, bound)
        if f32q is not None:
            eng.set_option("gemm", mode)
            losses, grad = eng.loss_grad(theta, w)
            eq = _errors(losses, grad, l32, g32)
            print(f"  gemm={mode:5s} vs oracle at float32(theta): loss rel {eq[0]:.2e}, grad rel L2 {eq[1]:.2e}, Linf {eq[2]:.2e}   x torch-f32: "
                  f"{eq[0] / f32q[0]:.2f} / {eq[1] / f32q[1]:.2f} / {eq[2] / f32q[2]:.2f}")
            # (two fp32 evaluations' rounding errors are independent realisations of comparable size: in a max-norm one can be 2-3 x the other
            # either way — measured 0.3 ... 2.6 x, profiles/r05_theta_variants_ab.txt — so this reference gets 3 x where the theta64 one gets 2 x)
            for i, e in enumerate(eq):
                assert e < max(TOL, 3.0 * f32q[i]), (name, tag, mode, "exact32", i, e, f32q[i])
    if trained:
        # the float64 mode at FULL size: the north star's tolerance as is, no relaxation
        eng.set_option("gemm", "split")
        eng.set_option("precision", "f64")
        for k, s_ in enumerate(sets):
            eng.set_points_f64(k, s_)
        l64, g64 = eng.loss_grad_f64(theta, w)
        e64 = _errors(l64, g64, g[f"losses_exact_{tag}"], g[f"grad_exact_{tag}"])
        print(f"  float64 mode ({eng.get_option('f64_path')}) vs exact oracle: loss rel {e64[0]:.2e}, grad rel L2 {e64[1]:.2e}, Linf {e64[2]:.2e}   margin to 1e-5: x{TOL / max(e64):.1e}")
        assert max(e64) < TOL, (name, tag, "f64", e64)
        if name in ("cfg2_variants", "cfg3_variants"):
            assert eng.get_option("f64_path") == "mfma"
        # r06: the reference-semantics validation mode (pinn_set_option "derivative" = "stencil": the reference's central differences with its
        # get_eps steps) at FULL size against the STENCIL oracle.  The bound is the reproducibility of the reference's own finite-difference
        # numbers: the stencil oracle against itself on the same function with permuted hidden neurons (`noise_stencil_<tag>`,
        # oracle/make_golden.py variants-noise) — 1e-16 |u| / eps^2 of rounding per point against a gradient that is a small difference of large terms
        if f"noise_stencil_{tag}" in g:
            noise = g[f"noise_stencil_{tag}"]
            eng.set_option("derivative", "stencil")
            ls, gs = eng.loss_grad_f64(theta, w)
            es = _errors(ls, gs, g[f"losses_stencil_{tag}"], g[f"grad_stencil_{tag}"])
            ee = _errors(l64, g64, g[f"losses_stencil_{tag}"], g[f"grad_stencil_{tag}"])
            print(f"  stencil mode vs STENCIL oracle: loss rel {es[0]:.2e}, grad rel L2 {es[1]:.2e}, Linf {es[2]:.2e}   (exact-derivative float64 kernels vs the stencil oracle: "
                  f"{ee[0]:.2e} / {ee[1]:.2e} / {ee[2]:.2e}; the stencil oracle against its permuted self: {noise[0]:.2e} / {noise[1]:.2e} / {noise[2]:.2e})")
            for i in range(3):
                assert es[i] < 4.0 * max(noise[i], 2e-9), (name, tag, "stencil mode", i, es[i], noise[i])
            eng.set_option("derivative", "exact")
        eng.set_option("precision", "f32")
