#!/usr/bin/env python3
"""Generates tests/golden/*.npz — golden input/output vectors of the float64 oracle (oracle/pinn_oracle.py) for reduced
versions of the BASELINE.json configurations.  The reference (Julia) cannot run here, so these vectors are produced by the
restatement itself ("parity unpinned" beyond the reference's own numeric pins, see the oracle header); they freeze the
oracle's behaviour and give the GPU tests fixtures that do not need torch autograd at run time.

    python oracle/make_golden.py
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


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
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
