"""Test helpers: restate a product-side PDESystem as an oracle Problem (oracle/pinn_oracle.py)."""
import numpy as np
import sympy as sp

import pinn_oracle as po


def oracle_problem(npde, pde_system, chains, param_estim=False):
    vi = npde.get_vars(pde_system.ivs, pde_system.dvs)
    sym_iv = {str(v): v for v in pde_system.ivs}
    ochains = [po.Chain(tuple(c.sizes), c.act) for c in chains]
    depfuncs = [d.func for d in pde_system.dvs]
    net_indvars = [tuple(sym_iv[n] for n in vi.dict_depvar_input[str(f)]) for f in depfuncs]

    def term(eq):
        from neuralpde_jl_amd.symbolic import term_indvars
        names = term_indvars(eq, vi)
        return po.TermSpec(eq.lhs, eq.rhs, tuple(sym_iv[n] for n in names))

    ps = tuple(pde_system.ps)
    default_p = np.array([float(pde_system.defaults[p]) for p in ps]) if ps else None
    return po.Problem(chains=ochains, depvars=depfuncs, net_indvars=net_indvars,
                      pde_terms=[term(e) for e in pde_system.eqs], bc_terms=[term(b) for b in pde_system.bcs],
                      params=ps, param_estim=param_estim, default_p=default_p)


def rel_errors(losses, grad, ref):
    le = np.abs(np.asarray(losses) - ref.term_losses) / np.maximum(np.abs(ref.term_losses), 1e-300)
    g = np.asarray(grad, dtype=np.float64)
    g2 = np.linalg.norm(g - ref.grad) / np.linalg.norm(ref.grad)
    gi = np.max(np.abs(g - ref.grad)) / np.max(np.abs(ref.grad))
    return le, g2, gi
