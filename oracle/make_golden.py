#!/usr/bin/env python3
"""Generates tests/golden/*.npz — golden input/output vectors of the float64 oracle (oracle/pinn_oracle.py) for reduced
versions of the BASELINE.json configurations.  The reference (Julia) cannot run here, so these vectors are produced by the
restatement itself ("parity unpinned" beyond the reference's own numeric pins, see the oracle header); they freeze the
oracle's behaviour and give the GPU tests fixtures that do not need torch autograd at run time.

    python oracle/make_golden.py                 # the small cases (inputs stored in the fixture)
    python oracle/make_golden.py full [names]    # the full-size cases (inputs pinned by SHA-256, minutes of CPU time each)
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


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        make_full(out, sys.argv[2:])
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
