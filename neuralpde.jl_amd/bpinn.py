"""Physics log-likelihood of a Bayesian PINN on top of the engine's per-term sums.

The reference's BPINN (ext/bpinn/PDE_BPINN.jl:425) sums, over the PDE and boundary terms, the closures that
`get_points_loss_functions` builds for `GridTraining` (src/training_strategies.jl:113-127):

    l_k(theta, sigma_k) = logpdf(MvNormal(r_k(X_k; theta), sigma_k^2 I), 0)
                        = -N_k/2 log(2 pi) - N_k log(sigma_k) - SSE_k / (2 sigma_k^2),      SSE_k = sum_i r_k(x_i; theta)^2

i.e. an affine function of the per-term sums of squares the engine already returns (L_k = SSE_k / N_k), and

    grad_theta sum_k l_k = - grad_theta sum_k w_k L_k        with  w_k = N_k / (2 sigma_k^2),

which is one `pinn_loss_grad` call with those term weights (reverse mode over all P parameters, where the reference
uses forward-mode ForwardDiff over every parameter of the network).  Priors, the data likelihood (`L2LossData`) and the
HMC/NUTS sampler itself stay on the host as in the reference.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np


def physics_loglikelihood(engine, theta, stds: Sequence[float], term_sizes: Sequence[int], want_grad: bool = True
                          ) -> Tuple[float, np.ndarray]:
    """sum_k logpdf(MvNormal(r_k, sigma_k^2 I), 0) and its gradient w.r.t. theta.
    stds: sigma_k per term (PDE terms first, then boundary terms: `allstd[1:2]` of the reference, flattened);
    term_sizes: N_k, the number of points of every term's (fixed) set."""
    sig = np.asarray(stds, dtype=np.float64)
    n = np.asarray(term_sizes, dtype=np.float64)
    if sig.shape != (engine.K,) or n.shape != (engine.K,):
        raise ValueError(f"need one std and one set size per loss term ({engine.K})")
    if np.any(sig <= 0):
        raise ValueError("standard deviations must be positive")
    w = n / (2.0 * sig ** 2)
    losses, grad = engine.loss_grad_f64(theta, w) if want_grad else (engine.loss_grad(theta, w, want_grad=False)[0], None)
    ll = float(np.sum(-0.5 * n * math.log(2.0 * math.pi) - n * np.log(sig) - w * np.asarray(losses, dtype=np.float64)))
    return ll, (None if grad is None else -np.asarray(grad, dtype=np.float64))


def loglikelihood(engine, theta, stds: Sequence[float], want_grad: bool = True):
    """The same log-likelihood through the engine's own entry point `pinn_loglik_grad`: returns (loglik, d/dtheta, d/dstds) — the stds
    as trailing sampler parameters (`allstd` of ext/bpinn/PDE_BPINN.jl:16-26 when they are not fixed).  Terms may include DataLoss
    terms: with their own std they are the L2LossData term (ext/bpinn/PDE_BPINN.jl:148-183), evaluated in the same fused call."""
    ll, g, gs = engine.loglik_grad(theta, stds, want_grad=want_grad)
    return ll, (None if g is None else g.astype(np.float64)), gs
