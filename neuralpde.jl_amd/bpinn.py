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
    ll, g, gs = engine.loglik_grad_f64(theta, stds, want_grad=want_grad)      # (double at the ABI: native on a handle in float64 mode, fp32 kernels + widening otherwise)
    return ll, (None if g is None else g.astype(np.float64)), gs


# ------------------------------------------------------------------------------------------------
# host-side sampler: the mirror of `ahmc_bayesian_pinn_pde` (ext/bpinn/PDE_BPINN.jl:371-640)
# ------------------------------------------------------------------------------------------------
class BPINNsolution:
    """ext/bpinn/PDE_BPINN.jl `BPINNsolution`: samples, per-dependent-variable ensemble prediction (mean, std over the last `numensemble`
    draws) on the `saveats` grid, and that grid."""

    def __init__(self, samples, ensemblesol, ensemblestd, timepoints, stats):
        self.samples, self.ensemblesol, self.ensemblestd, self.timepoints, self.stats = samples, ensemblesol, ensemblestd, timepoints, stats


def _hmc(logp_grad, theta0, draw_samples, n_leapfrog, eps0, target, rng, n_adapts=None):
    """HMC with `n_leapfrog` leapfrog steps per draw ([3P] AdvancedHMC.HMC(eps, n_leapfrog)), Stan-style adaptation during the first
    n_adapts = min(draw_samples / 10, 1000) draws (AdvancedHMC's default): dual averaging of the step size towards `target` acceptance
    and a diagonal metric from the windowed sample variance (StanHMCAdaptor + DiagEuclideanMetric)."""
    n = theta0.size
    n_adapts = min(draw_samples // 10, 1000) if n_adapts is None else n_adapts
    th = theta0.astype(np.float64).copy()
    lp, g = logp_grad(th)
    minv = np.ones(n)                                # inverse mass (diagonal metric)

    def leap(th, r, g, eps, steps):
        r = r + 0.5 * eps * g
        for s in range(steps):
            th = th + eps * minv * r
            lp, g = logp_grad(th)
            r = r + (eps if s < steps - 1 else 0.5 * eps) * g
        return th, r, lp, g

    # find_good_stepsize: double / halve until the one-step acceptance crosses 0.8 (AdvancedHMC.find_good_stepsize)
    eps = eps0
    r0 = rng.standard_normal(n) / np.sqrt(minv)
    h0 = -lp + 0.5 * np.sum(minv * r0 * r0)
    t1, r1, lp1, _ = leap(th, r0, g, eps, 1)
    d = h0 - (-lp1 + 0.5 * np.sum(minv * r1 * r1))
    direction = 1.0 if (np.isfinite(d) and d > np.log(0.8)) else -1.0
    for _ in range(40):
        eps *= 2.0 ** direction
        t1, r1, lp1, _ = leap(th, r0, g, eps, 1)
        d = h0 - (-lp1 + 0.5 * np.sum(minv * r1 * r1))
        if not np.isfinite(d):
            d = -np.inf
        if (direction > 0) == (d <= np.log(0.8)):
            break
    mu, hbar, log_eps_bar, t0, gamma, kappa = np.log(10 * eps), 0.0, 0.0, 10.0, 0.05, 0.75
    w0, w1 = int(0.15 * n_adapts), int(0.9 * n_adapts)            # variance window of the metric adaptation
    win = []
    samples, accs = [], []
    m_count = 0
    for it in range(draw_samples):
        r = rng.standard_normal(n) / np.sqrt(minv)
        h_old = -lp + 0.5 * np.sum(minv * r * r)
        tn, rn, lpn, gn = leap(th, r, g, eps, n_leapfrog)
        h_new = -lpn + 0.5 * np.sum(minv * rn * rn)
        a = float(np.exp(min(0.0, h_old - h_new))) if np.isfinite(h_new) else 0.0
        if rng.random() < a:
            th, lp, g = tn, lpn, gn
        samples.append(th.copy())
        accs.append(a)
        if it < n_adapts:
            m_count += 1
            hbar = (1 - 1 / (m_count + t0)) * hbar + (target - a) / (m_count + t0)
            log_eps = mu - np.sqrt(m_count) / gamma * hbar
            eta = m_count ** (-kappa)
            log_eps_bar = eta * log_eps + (1 - eta) * log_eps_bar
            eps = float(np.exp(log_eps))
            if w0 <= it < w1:
                win.append(th.copy())
            if it == w1 - 1 and len(win) >= 8 and n_adapts >= 100:       # (a metric from fewer draws is noise; short runs keep the unit metric)
                var = np.var(np.asarray(win), axis=0)
                k = len(win)
                minv = (k / (k + 5.0)) * var + 1e-3 * (5.0 / (k + 5.0))
                mu, hbar, log_eps_bar, m_count = np.log(10 * eps), 0.0, 0.0, 0          # restart the step-size search on the new metric
            if it == n_adapts - 1:
                eps = float(np.exp(log_eps_bar)) if m_count > 0 else eps
    return np.asarray(samples), {"acceptance": np.asarray(accs), "step_size": eps, "inv_metric": minv, "n_adapts": n_adapts}


class LogNormal:
    """[3P] Distributions.LogNormal(mu, sigma) as a parameter prior (`param = [LogNormal(6.0, 0.5)]`, test/PDEBPINN/bpinn_pde__bpinn_pde_inv_i_*.jl:58)."""

    def __init__(self, mu, sigma):
        self.mu, self.sigma = float(mu), float(sigma)

    def params(self):
        return (self.mu, self.sigma)

    def logpdf_grad(self, x):
        if not x > 0.0:
            return -np.inf, 0.0
        z = (np.log(x) - self.mu) / self.sigma
        return float(-np.log(x * self.sigma * np.sqrt(2 * np.pi)) - 0.5 * z * z), float(-1.0 / x - z / (self.sigma * x))


class Normal:
    """[3P] Distributions.Normal(mu, sigma) as a parameter prior."""

    def __init__(self, mu, sigma):
        self.mu, self.sigma = float(mu), float(sigma)

    def params(self):
        return (self.mu, self.sigma)

    def logpdf_grad(self, x):
        z = (x - self.mu) / self.sigma
        return float(-np.log(self.sigma * np.sqrt(2 * np.pi)) - 0.5 * z * z), float(-z / self.sigma)


def ahmc_bayesian_pinn_pde(npde, pde_system, discretization, draw_samples=1000, bcstd=(0.01,), l2std=(0.05,), phystd=(0.05,), priorsNNw=(0.0, 2.0),
                           param=(), n_leapfrog=30, step_size=0.1, targetacceptancerate=0.8, saveats=(0.1,), numensemble=None, rng=None):
    """`ahmc_bayesian_pinn_pde(pde_system, discretization; draw_samples, bcstd, phystd, priorsNNw, Kernel = HMC(0.1, 30), saveats,
    numensemble)` — the forward-problem form of ext/bpinn/PDE_BPINN.jl:371-640: the posterior over the network parameters is
    prior N(priorsNNw[1], priorsNNw[2]^2 I) x physics likelihood (`pinn_loglik_grad`: every leapfrog step is ONE fused device evaluation,
    reverse mode over all parameters where the reference differentiates forward over each of them); the HMC sampler, its adaptation and
    the ensemble statistics run on the host, as in the reference.  `discretization`: a PhysicsInformedNN with fixed point sets (the
    reference's BayesianPINN takes GridTraining).  Inverse problems: `param = [prior of every PDE parameter]` (the chain starts at the
    priors' first parameter, as in the reference), the discretization built with `param_estim = True` and the observations as `data_loss`
    terms (the reference's `dataset`), whose standard deviations are `l2std` — the L2 data term then sits in the same fused device call."""
    rng = np.random.default_rng() if rng is None else rng
    rep = npde.symbolic_discretize(pde_system, discretization)
    eng = rep.engine
    n_pde, n_bc = len(rep.eqs), len(rep.bcs)
    bro = lambda s, n: list(np.broadcast_to(np.asarray(s, dtype=np.float64).reshape(-1), (n,))) if np.size(s) in (1, n) else None
    ps, bs = bro(phystd, n_pde), bro(bcstd, n_bc)
    if ps is None or bs is None:
        raise ValueError("phystd / bcstd: one standard deviation per equation / boundary condition (or one for all)")
    n_data = eng.K - n_pde - n_bc
    ls = bro(l2std, n_data) if n_data else []
    if ls is None:
        raise ValueError("l2std: one standard deviation per data term (or one for all)")
    stds = np.asarray(ps + bs + ls, dtype=np.float64)
    mu0, sd0 = float(priorsNNw[0]), float(priorsNNw[1])
    ninv = len(param)
    if ninv and (not discretization.param_estim or ninv != int(eng.P - sum(c.nparams for c in discretization.chain))):
        raise ValueError("param: one prior per estimated PDE parameter, and the discretization needs param_estim = True")
    nn = eng.P - ninv

    def logp_grad(th):
        ll, g, _ = eng.loglik_grad_f64(th, stds)
        g = g.astype(np.float64)
        w = th[:nn]
        lp = ll - 0.5 * np.sum(((w - mu0) / sd0) ** 2) - nn * (np.log(sd0) + 0.5 * np.log(2 * np.pi))
        g[:nn] -= (w - mu0) / sd0 ** 2
        for j, pr in enumerate(param):
            l, d = pr.logpdf_grad(float(th[nn + j]))
            lp += l
            g[nn + j] += d
        return lp, g
    theta0 = np.asarray(rep.flat_init_params, dtype=np.float64).copy()
    for j, pr in enumerate(param):
        theta0[nn + j] = pr.params()[0]
    samples, stats = _hmc(logp_grad, theta0, int(draw_samples), int(n_leapfrog), float(step_size), float(targetacceptancerate), rng)
    numensemble = int(draw_samples // 3) if numensemble is None else int(numensemble)
    # inference: the last `numensemble` draws on the saveats grid (one spacing per independent variable), per dependent variable
    doms = {str(d.variable): (float(d.domain.lo), float(d.domain.hi)) for d in pde_system.domain}
    # the reference slices samples[(end - numensemble):end] — numensemble + 1 draws (ext/bpinn/PDE_BPINN.jl:254) — estimates the PDE
    # parameters over all of them (:264-269) and builds the ensemble curves from the first numensemble of the slice (:302); restated as is
    ens_samples = samples[-(numensemble + 1):]
    ens, ens_std, tps = [], [], []
    for i, name in enumerate(rep.depvars):
        ins = list(rep.dict_depvar_input[name])
        if len(saveats) not in (1, len(pde_system.ivs)):
            raise ValueError("saveats: one grid spacing per independent variable")
        axes = []
        for v in ins:
            k = [str(q) for q in pde_system.ivs].index(str(v))
            lo, hi = doms[str(v)]
            step = float(saveats[k if len(saveats) > 1 else 0])
            axes.append(np.arange(lo, hi + 0.5 * step, step))
        mesh = np.meshgrid(*axes, indexing="ij")
        pts = np.stack([m.ravel() for m in mesh])
        preds = np.stack([rep.phi[i](pts, npde.depvar_params(rep, th, name))[0] if isinstance(rep.phi, (list, tuple)) else rep.phi(pts, th)[0]
                          for th in ens_samples[:numensemble]])     # the reference predicts with the FIRST numensemble of its slice (:302)
        ens.append(preds.mean(axis=0)); ens_std.append(preds.std(axis=0)); tps.append(pts)
    sol = BPINNsolution(samples, ens, ens_std, tps, stats)
    sol.estimated_de_params = [float(ens_samples[:, nn + j].mean()) for j in range(ninv)]
    return sol
