#!/usr/bin/env python3
"""Generates tests/golden/*.npz — golden input/output vectors of the float64 oracle (oracle/pinn_oracle.py) for reduced
versions of the BASELINE.json configurations.  The reference (Julia) cannot run here, so these vectors are produced by the
restatement itself ("parity unpinned" beyond the reference's own numeric pins, see the oracle header); they freeze the
oracle's behaviour and give the GPU tests fixtures that do not need torch autograd at run time.

    python oracle/make_golden.py                 # the small cases (inputs stored in the fixture)
    python oracle/make_golden.py full [names]    # the full-size cases (inputs pinned by SHA-256, minutes of CPU time each)
    python oracle/make_golden.py variants [names]  # the same problems at SCALED and TRAINED parameters, both oracle modes (r04: the regime
                                                 # where the split-operand bf16 GEMMs of the 64- / 128-wide kernels have the least margin)
    python oracle/make_golden.py variants-trained [names]  # appends the trained variants an existing variants fixture lacks (r06: cfg4 / cfg5)
    python oracle/make_golden.py variants-noise [names]  # adds the stencil oracle's own reproducibility (permuted neurons) at the trained variants (r06)
    python oracle/make_golden.py variants-theta32 [names]  # adds the exact oracle at the float32-rounded trained parameters (r05)
    python oracle/make_golden.py variants-f32 [names]  # adds the float32 evaluation of the same program to the variants fixtures
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pinn_import  # noqa: E402
import pinn_oracle as po  # noqa: E402
import helpers  # noqa: E402

m = pinn_import.load()
from neuralpde_jl_amd import strategies, symbolic, workloads  # noqa: E402

CASES = {
    "cfg1_poisson1d_1024": lambda: workloads.cfg1_poisson1d(1024),
    "cfg2_poisson2d_512": lambda: workloads.cfg2_poisson2d(points=512, bcs_points=128),
    "cfg3_burgers_512": lambda: workloads.cfg3_burgers(points=512, bcs_points=128),
}


def point_sets(wl):
    vi = symbolic.get_vars(wl.pde_system.ivs, wl.pde_system.dvs)
    pde, bc, _ = wl.strategy.point_sets(wl.pde_system, vi, np.float64)
    return list(pde) + list(bc)


# Full-size cases (the benchmarked configuration and the larger BASELINE configs): the point sets are too large to commit, so the
# fixture pins them by SHA-256 (the test regenerates them with the same seeded generator and checks the digest) and stores
# theta, weights and the oracle's outputs.  The oracle runs in chunks of CHUNK points per term: every term's loss is
# sum_c (n_c/N) L_c and the gradient is the sum of the chunk gradients under the weights w_k n_ck/N_k.
FULL_CASES = {
    "cfg2_full": (lambda: workloads.cfg2_poisson2d(points=65536), dict(points=65536)),
    "cfg3_full": (lambda: workloads.cfg3_burgers(points=262144), dict(points=262144)),
    "cfg4_full": (lambda: workloads.cfg4_cavity(points=262144, bcs_points=32768), dict(points=262144, bcs_points=32768)),
    "cfg5_full": (lambda: workloads.cfg5_heat_inverse(points=1000000, bcs_points=65536), dict(points=1000000, bcs_points=65536)),
}
CHUNK = 16384


def set_digest(s):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(s, dtype=np.float64).tobytes()).hexdigest()


def chunked_loss_and_grad(prob, theta, sets, w, mode="stencil", chunk=CHUNK):
    K, N = len(sets), [s.shape[1] for s in sets]
    losses, grad = np.zeros(K), np.zeros(len(theta))
    nchunks = max((n + chunk - 1) // chunk for n in N)
    for c in range(nchunks):
        # every term contributes its c-th chunk (a term with fewer chunks contributes a single point at weight 0)
        part, wc, frac = [], [], []
        for k, s in enumerate(sets):
            lo, hi = c * chunk, min((c + 1) * chunk, N[k])
            if lo >= hi:
                part.append(s[:, :1]); wc.append(0.0); frac.append(0.0)
            else:
                part.append(s[:, lo:hi]); wc.append(w[k] * (hi - lo) / N[k]); frac.append((hi - lo) / N[k])
        ev = po.loss_and_grad(prob, theta, part, weights=wc, mode=mode)
        losses += np.array(frac) * ev.term_losses
        grad += ev.grad
        print(f"    chunk {c + 1}/{nchunks}", flush=True)
    return losses, grad


def make_full(out, only=None):
    import time
    for name, (make, kw) in FULL_CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        wl = make()
        sets = point_sets(wl)
        prob = helpers.oracle_problem(m, wl.pde_system, wl.chains, param_estim=wl.param_estim)
        K = len(sets)
        w = (np.concatenate([wl.adaptive_loss.pde_loss_weights * np.ones(len(wl.pde_system.eqs)),
                             wl.adaptive_loss.bc_loss_weights * np.ones(len(wl.pde_system.bcs))])
             if wl.adaptive_loss is not None else np.linspace(1.0, 2.0, K))
        print(name, "points per term", [s.shape[1] for s in sets], flush=True)
        theta = wl.theta
        if wl.param_estim:            # theta.p block appended as symbolic_discretize does (src/discretize.jl:457-462)
            theta = np.concatenate([theta, [float(wl.pde_system.defaults[p]) for p in wl.pde_system.ps]])
        losses, grad = chunked_loss_and_grad(prob, theta, sets, w)
        d = {"theta": theta, "weights": w, "losses_stencil": losses, "grad_stencil": grad.astype(np.float64), "nsets": np.array(K),
             "set_sha256": np.array([set_digest(s) for s in sets]), "set_sizes": np.array([s.shape[1] for s in sets]),
             "chunk": np.array(CHUNK)}
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        print(name, "losses", losses, f"({time.time() - t0:.0f} s)", flush=True)


# ---- parameter variants (r04) ----
# Every loss / gradient fixture above sits at glorot-initialised parameters, where pre-activations are O(1) and nothing saturates.  The
# variants freeze the oracle at parameters that stress the arithmetic: theta x 2 and x 4 (saturating tanh units, large higher-derivative
# jets) and TRAINED parameters (Adam in float64 on the oracle's own exact-derivative loss over a reduced point design: gradients that are
# small differences of large per-point terms).  Both oracle modes are stored: "stencil" = the reference's finite-difference semantics
# (src/pinn_types.jl:445-482), "exact" = the same program with exact derivatives (what the engine computes in fp32).
VARIANT_CASES = {
    # name: (full-size maker, maker of the reduced problem the training runs on, scale factors, Adam iteration counts)
    "cfg2_variants": (lambda: workloads.cfg2_poisson2d(points=65536), lambda: workloads.cfg2_poisson2d(points=2048, bcs_points=512), (2.0, 4.0), (2000, 6000)),
    "cfg3_variants": (lambda: workloads.cfg3_burgers(points=262144), lambda: workloads.cfg3_burgers(points=2048, bcs_points=512), (2.0,), (2000,)),
    # r06: trained parameters for the 128-wide configurations too (the two-pass split GEMM of H = 128 where its bias matters); the training runs
    # on a small design (`variants-trained` appends the tags to an existing fixture without recomputing the scaled ones)
    "cfg4_variants": (lambda: workloads.cfg4_cavity(points=16384, bcs_points=4096), lambda: workloads.cfg4_cavity(points=1024, bcs_points=256), (2.0,), (400,)),
    "cfg5_variants": (lambda: workloads.cfg5_heat_inverse(points=32768, bcs_points=8192), lambda: workloads.cfg5_heat_inverse(points=2048, bcs_points=512), (2.0, 4.0), (600,)),
}


def full_theta(wl):
    theta = np.asarray(wl.theta, dtype=np.float64)
    if wl.param_estim:
        theta = np.concatenate([theta, [float(wl.pde_system.defaults[p]) for p in wl.pde_system.ps]])
    return theta


def fixture_weights(wl, K):
    if wl.adaptive_loss is not None:
        return np.concatenate([wl.adaptive_loss.pde_loss_weights * np.ones(len(wl.pde_system.eqs)),
                               wl.adaptive_loss.bc_loss_weights * np.ones(len(wl.pde_system.bcs))])
    return np.linspace(1.0, 2.0, K)


def adam_train(prob, theta0, sets, w, iters, lr=3e-3, decay=0.6, report=500):
    """plain Adam (beta 0.9 / 0.999, eps 1e-8, bias-corrected) in float64 on the oracle's exact-derivative objective; the step size
    falls by `decay` every 1000 iterations (the schedule of examples/poisson2d_train.py).  Returns theta after each count in `iters`."""
    th = np.array(theta0, dtype=np.float64)
    m_, v_ = np.zeros_like(th), np.zeros_like(th)
    out = {}
    for t in range(1, max(iters) + 1):
        ev = po.loss_and_grad(prob, th, sets, weights=w, mode="exact")
        g = ev.grad
        m_ = 0.9 * m_ + 0.1 * g
        v_ = 0.999 * v_ + 0.001 * g * g
        step = lr * decay ** ((t - 1) // 1000)
        th = th - step * (m_ / (1 - 0.9 ** t)) / (np.sqrt(v_ / (1 - 0.999 ** t)) + 1e-8)
        if t % report == 0:
            print(f"    adam {t}: objective {ev.loss:.4e}", flush=True)
        if t in iters:
            out[t] = th.copy()
    return out


def make_variants(out, only=None):
    import time
    for name, (make, make_small, scales, adam_iters) in VARIANT_CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        wl = make()
        sets = point_sets(wl)
        prob = helpers.oracle_problem(m, wl.pde_system, wl.chains, param_estim=wl.param_estim)
        K = len(sets)
        w = fixture_weights(wl, K)
        theta0 = full_theta(wl)
        nnet = len(np.asarray(wl.theta))                     # network parameters (theta.p, if any, is not scaled)
        thetas = {}
        for sc in scales:
            th = theta0.copy()
            th[:nnet] *= sc
            thetas[f"x{sc:g}"] = th
        if adam_iters:
            wls = make_small()
            probs = helpers.oracle_problem(m, wls.pde_system, wls.chains, param_estim=wls.param_estim)
            trained = adam_train(probs, full_theta(wls), point_sets(wls), fixture_weights(wls, K), set(adam_iters))
            for it in adam_iters:
                thetas[f"adam{it}"] = trained[it]
        d = {"weights": w, "nsets": np.array(K), "tags": np.array(list(thetas.keys())),
             "set_sha256": np.array([set_digest(s) for s in sets]), "set_sizes": np.array([s.shape[1] for s in sets])}
        for tag, th in thetas.items():
            d["theta_" + tag] = th
            for mode in ("stencil", "exact"):
                losses, grad = chunked_loss_and_grad(prob, th, sets, w, mode=mode)
                d[f"losses_{mode}_{tag}"] = losses
                d[f"grad_{mode}_{tag}"] = grad.astype(np.float64)
            gs, ge = d[f"grad_stencil_{tag}"], d[f"grad_exact_{tag}"]
            print(name, tag, "losses", d[f"losses_stencil_{tag}"], "|grad|", np.linalg.norm(gs), "stencil-vs-exact grad rel L2",
                  np.linalg.norm(gs - ge) / np.linalg.norm(ge), f"({time.time() - t0:.0f} s)", flush=True)
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)


def add_trained(out, only=None):
    """appends the TRAINED variants of VARIANT_CASES that an existing fixture lacks (both oracle modes); run `variants-f32` and
    `variants-theta32` afterwards for the float32 evaluation / the oracle at float32(theta) of the new tags"""
    import time
    for name, (make, make_small, scales, adam_iters) in VARIANT_CASES.items():
        path = os.path.join(out, name + ".npz")
        if (only and name not in only) or not os.path.exists(path) or not adam_iters:
            continue
        d = dict(np.load(path))
        have = [str(t) for t in d["tags"]]
        todo = [it for it in adam_iters if f"adam{it}" not in have]
        if not todo:
            continue
        t0 = time.time()
        wl = make()
        sets = point_sets(wl)
        assert [set_digest(s) for s in sets] == list(d["set_sha256"])
        prob = helpers.oracle_problem(m, wl.pde_system, wl.chains, param_estim=wl.param_estim)
        K = len(sets)
        w = d["weights"]
        wls = make_small()
        probs = helpers.oracle_problem(m, wls.pde_system, wls.chains, param_estim=wls.param_estim)
        trained = adam_train(probs, full_theta(wls), point_sets(wls), fixture_weights(wls, K), set(todo), report=100)
        for it in todo:
            tag, th = f"adam{it}", trained[it]
            d["theta_" + tag] = th
            for mode in ("stencil", "exact"):
                losses, grad = chunked_loss_and_grad(prob, th, sets, w, mode=mode)
                d[f"losses_{mode}_{tag}"], d[f"grad_{mode}_{tag}"] = losses, grad.astype(np.float64)
            gs, ge = d[f"grad_stencil_{tag}"], d[f"grad_exact_{tag}"]
            print(name, tag, "losses", d[f"losses_stencil_{tag}"], "|grad|", np.linalg.norm(gs), "stencil-vs-exact grad rel L2",
                  np.linalg.norm(gs - ge) / np.linalg.norm(ge), f"({time.time() - t0:.0f} s)", flush=True)
            have.append(tag)
        d["tags"] = np.array(have)
        np.savez_compressed(path, **d)


def add_f32(out, only=None):
    """adds to every variants fixture the SAME program evaluated in float32 (torch CPU, exact-derivative mode: `losses_f32_<tag>`,
    `grad_f32_<tag>`): what a plain fp32 implementation of the reference's mathematics returns.  At trained parameters the residual is a
    small difference of O(1) terms and NO fp32 evaluation reaches 1e-5 relative to the float64 result; the parity tests bound the
    engine's error there by a small multiple of this evaluation's error (tests/test_gpu_theta_variants.py)."""
    import torch
    for name, (make, _, _, _) in VARIANT_CASES.items():
        path = os.path.join(out, name + ".npz")
        if (only and name not in only) or not os.path.exists(path):
            continue
        d = dict(np.load(path))
        wl = make()
        sets = point_sets(wl)
        assert [set_digest(s) for s in sets] == list(d["set_sha256"])
        prob = helpers.oracle_problem(m, wl.pde_system, wl.chains, param_estim=wl.param_estim)
        po.DT = torch.float32
        try:
            for tag in d["tags"]:
                if f"grad_f32_{tag}" in d:
                    continue                                 # (already there: `variants-trained` appends tags)
                losses, grad = chunked_loss_and_grad(prob, d["theta_" + str(tag)], sets, d["weights"], mode="exact")
                d[f"losses_f32_{tag}"], d[f"grad_f32_{tag}"] = losses, grad.astype(np.float64)
                ge = d[f"grad_exact_{tag}"]
                print(name, tag, "float32 evaluation vs float64: loss rel", np.max(np.abs(losses - d[f"losses_exact_{tag}"]) / np.abs(d[f"losses_exact_{tag}"])),
                      "grad rel L2", np.linalg.norm(grad - ge) / np.linalg.norm(ge), flush=True)
        finally:
            po.DT = torch.float64
        np.savez_compressed(path, **d)


def add_theta32(out, only=None):
    """adds to every variants fixture the exact-derivative float64 oracle AT THE FLOAT32-ROUNDED PARAMETERS of the trained variants
    (`losses_exact32_<tag>`, `grad_exact32_<tag>`): an fp32 engine is handed float32(theta), so at a trained theta — where the gradient is
    a small difference of large terms and H dtheta of the rounding alone is 4e-5 ... 1e-2 of it — this is the reference that isolates the
    engine's ARITHMETIC error from the quantisation of its input (r05; tools/r05/theta_ab_gpu.py, tests/test_gpu_theta_variants.py)."""
    for name, (make, _, _, _) in VARIANT_CASES.items():
        path = os.path.join(out, name + ".npz")
        if (only and name not in only) or not os.path.exists(path):
            continue
        d = dict(np.load(path))
        tags = [str(t) for t in d["tags"] if str(t).startswith("adam") and f"grad_exact32_{t}" not in d]
        if not tags:
            continue
        wl = make()
        sets = point_sets(wl)
        assert [set_digest(s) for s in sets] == list(d["set_sha256"])
        prob = helpers.oracle_problem(m, wl.pde_system, wl.chains, param_estim=wl.param_estim)
        for tag in tags:
            th32 = d["theta_" + tag].astype(np.float32).astype(np.float64)
            losses, grad = chunked_loss_and_grad(prob, th32, sets, d["weights"], mode="exact")
            d[f"losses_exact32_{tag}"], d[f"grad_exact32_{tag}"] = losses, grad.astype(np.float64)
            ge = d[f"grad_exact_{tag}"]
            print(name, tag, "oracle(theta64) vs oracle(float32(theta64)): loss rel", np.max(np.abs(losses - d[f"losses_exact_{tag}"]) / np.abs(losses)),
                  "grad rel L2", np.linalg.norm(grad - ge) / np.linalg.norm(grad), flush=True)
        np.savez_compressed(path, **d)


def hidden_permutation(chain, nparams_total, rng):
    """index vector idx: theta[idx] is the SAME network function with the neurons of every hidden layer permuted (another summation order
    in every layer); its gradient is grad[idx]"""
    ix = np.arange(nparams_total)
    sizes, out, o, perm_in = chain.sizes, [], 0, None
    for l in range(len(sizes) - 1):
        n_in, n_out = sizes[l], sizes[l + 1]
        W = ix[o:o + n_in * n_out].reshape(n_in, n_out).T.copy()
        b = ix[o + n_in * n_out:o + n_in * n_out + n_out].copy()
        o += n_in * n_out + n_out
        if perm_in is not None:
            W = W[:, perm_in]
        perm_out = rng.permutation(n_out) if l < len(sizes) - 2 else np.arange(n_out)
        W, b = W[perm_out], b[perm_out]
        out += [W.T.reshape(-1), b]
        perm_in = perm_out
    return np.concatenate(out + [ix[o:]])


def add_stencil_noise(out, only=None):
    """adds `noise_stencil_<tag>` = (loss rel, grad rel L2, grad rel Linf) of the STENCIL oracle against itself on the same function with
    permuted hidden neurons (first network), for the trained variants (r06): how reproducible the reference's finite-difference numbers are
    across summation orders — u(x +- eps) carries ~1e-16 relative rounding and the difference formulas multiply it by 1 / eps^2 ~ 7e7; at a
    trained theta, where the gradient is a small difference of large per-point terms, that is 1e-5 ... 1e-4 of the gradient.  The bound the
    engine's stencil mode is held to (tests/test_gpu_theta_variants.py)."""
    for name, (make, _, _, _) in VARIANT_CASES.items():
        path = os.path.join(out, name + ".npz")
        if (only and name not in only) or not os.path.exists(path):
            continue
        d = dict(np.load(path))
        tags = [str(t) for t in d["tags"] if str(t).startswith("adam") and f"noise_stencil_{t}" not in d]
        if not tags:
            continue
        wl = make()
        sets = point_sets(wl)
        assert [set_digest(s) for s in sets] == list(d["set_sha256"])
        prob = helpers.oracle_problem(m, wl.pde_system, wl.chains, param_estim=wl.param_estim)
        for tag in tags:
            th = d["theta_" + tag]
            idx = hidden_permutation(prob.chains[0], len(th), np.random.default_rng(7))
            losses, grad = chunked_loss_and_grad(prob, th[idx], sets, d["weights"], mode="stencil")
            l0, g0 = d[f"losses_stencil_{tag}"], d[f"grad_stencil_{tag}"][idx]
            noise = np.array([np.max(np.abs(losses - l0) / np.abs(l0)), np.linalg.norm(grad - g0) / np.linalg.norm(g0), np.max(np.abs(grad - g0)) / np.max(np.abs(g0))])
            d[f"noise_stencil_{tag}"] = noise
            print(name, tag, "stencil oracle against its permuted self: loss rel, grad rel L2, Linf =", noise, flush=True)
        np.savez_compressed(path, **d)


def main():
    out = os.path.join(ROOT, "tests", "golden")
    if len(sys.argv) > 1 and sys.argv[1] == "variants-noise":
        add_stencil_noise(out, sys.argv[2:])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants-theta32":
        add_theta32(out, sys.argv[2:])
        return
    os.makedirs(out, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "variants-f32":
        add_f32(out, sys.argv[2:])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        make_full(out, sys.argv[2:])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants-trained":
        add_trained(out, sys.argv[2:])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        make_variants(out, sys.argv[2:])
        return
    for name, make in CASES.items():
        wl = make()
        sets = point_sets(wl)
        prob = helpers.oracle_problem(m, wl.pde_system, wl.chains)
        K = len(sets)
        w = np.linspace(1.0, 2.0, K)
        st = po.loss_and_grad(prob, wl.theta, sets, weights=w, mode="stencil")
        ex = po.loss_and_grad(prob, wl.theta, sets, weights=w, mode="exact")
        d = {"theta": wl.theta, "weights": w, "losses_stencil": st.term_losses, "grad_stencil": st.grad,
             "losses_exact": ex.term_losses, "grad_exact": ex.grad, "nsets": np.array(K)}
        for k, s in enumerate(sets):
            d[f"set{k}"] = s
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        print(name, "losses", st.term_losses, "stencil-vs-exact grad rel", np.linalg.norm(st.grad - ex.grad) / np.linalg.norm(ex.grad))


if __name__ == "__main__":
    main()


def reference_tables():
    """Extract golden data the reference's own tests hold (needs /root/reference; the .npz is committed)."""
    import re
    src = open("/root/reference/test/DGM/dgm__burger_s_equation.jl").read()
    body = re.search(r"const BURGER_REF_U = \[(.*?)\n\]", src, re.S).group(1)
    rows = [r.strip().rstrip(";") for r in body.strip().split("\n")]
    U = np.array([[float(v) for v in r.split()] for r in rows])
    assert U.shape == (11, 21)
    out = os.path.join(ROOT, "tests", "golden", "reference")
    os.makedirs(out, exist_ok=True)
    np.savez(os.path.join(out, "dgm_burgers_mol_table.npz"), ts=np.linspace(0, 1, 11), xs=np.linspace(-1, 1, 21), u=U)
