"""Adaptive loss weighting — host-side mirror of src/adaptive_losses.jl on top of the engine's per-term outputs.

The update rules stay on the host exactly as in the reference (they run outside AD, `@ignore_derivatives`,
src/discretize.jl:574-580); what they consume comes from the engine:
  * per-term losses                 <- pinn_loss_grad (`term_losses`)
  * per-term gradients d L_k/d theta <- pinn_term_grads   (GradientScaleAdaptiveLoss, src/adaptive_losses.jl:112-123)
Each scheme mutates `pde_loss_weights` / `bc_loss_weights`; the weighted sum and its gradient then use the new weights
(src/discretize.jl:582-588).  `reweight(theta, pde_losses, bc_losses, iteration, term_grads)` is called once per
full_loss_function evaluation, like the closure returned by `generate_adaptive_loss_function`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np


NONZERO_DIVISOR_EPS = 1.0e-7          # src/adaptive_losses.jl:125 (effective value, see GradientScaleAdaptiveLoss.reweight)


def _vectorify(w, n: Optional[int] = None) -> np.ndarray:
    a = np.atleast_1d(np.asarray(w, dtype=np.float64)).copy()
    return a


class AbstractAdaptiveLoss:
    """src/adaptive_losses.jl: AbstractAdaptiveLoss."""
    needs_term_grads = False
    reweight_every = 0          # 0: never

    def broadcast(self, n_pde: int, n_bc: int):
        """scalars are broadcast to the number of terms (src/discretize.jl:553-559); a wrong-length vector errors"""
        self.pde_loss_weights = np.ones(n_pde) * self.pde_loss_weights
        self.bc_loss_weights = np.ones(n_bc) * self.bc_loss_weights

    def fires(self, iteration: int) -> bool:
        return self.reweight_every > 0 and iteration % self.reweight_every == 0

    def reweight(self, theta, pde_losses, bc_losses, iteration: int, term_grads: Optional[Callable] = None):
        return None


class NonAdaptiveLoss(AbstractAdaptiveLoss):
    """NonAdaptiveLoss(; pde_loss_weights = 1, bc_loss_weights = 1, additional_loss_weights = 1) — :22-42."""

    def __init__(self, pde_loss_weights=1.0, bc_loss_weights=1.0, additional_loss_weights=1.0):
        self.pde_loss_weights = _vectorify(pde_loss_weights)
        self.bc_loss_weights = _vectorify(bc_loss_weights)
        self.additional_loss_weights = _vectorify(additional_loss_weights)


class GradientScaleAdaptiveLoss(AbstractAdaptiveLoss):
    """GradientScaleAdaptiveLoss(reweight_every; weight_change_inertia = 0.9, ...) — :75-151.
    bc weights <- inertia * w + (1 - inertia) * max_k max|grad L_pde_k| / (mean|grad L_bc_j| + eps)."""
    needs_term_grads = True

    def __init__(self, reweight_every: int, weight_change_inertia=0.9, pde_loss_weights=1.0, bc_loss_weights=1.0,
                 additional_loss_weights=1.0):
        self.reweight_every = int(reweight_every)
        self.weight_change_inertia = float(weight_change_inertia)
        self.pde_loss_weights = _vectorify(pde_loss_weights)
        self.bc_loss_weights = _vectorify(bc_loss_weights)
        self.additional_loss_weights = _vectorify(additional_loss_weights)

    def reweight(self, theta, pde_losses, bc_losses, iteration, term_grads=None):
        if not self.fires(iteration):
            return
        tg = np.asarray(term_grads(), dtype=np.float64)           # K x P, pde terms first
        n_pde = len(pde_losses)
        pde_grads_max = max(np.max(np.abs(tg[k])) for k in range(n_pde))
        bc_grads_mean = np.array([np.mean(np.abs(tg[n_pde + j])) for j in range(len(bc_losses))])
        # nonzero_divisor_eps (:125): `adaloss_T isa Float64 ? 1e-11 : convert(adaloss_T, 1e-7)` — `adaloss_T` is a TYPE, and a type is
        # never an instance of Float64, so the reference always takes the second branch: 1e-7, also for Float64 weights (restated as is)
        proposed = pde_grads_max / (bc_grads_mean + NONZERO_DIVISOR_EPS)
        a = self.weight_change_inertia
        self.bc_loss_weights = a * self.bc_loss_weights + (1 - a) * proposed


class _Adam:
    """[3P] Optimisers.Adam on a small vector (state for MiniMaxAdaptiveLoss)."""

    def __init__(self, eta, n, beta=(0.9, 0.999), eps=1e-8):
        self.eta, self.beta, self.eps = eta, beta, eps
        self.m, self.v, self.t = np.zeros(n), np.zeros(n), 0

    def update(self, x, dx):
        self.t += 1
        b1, b2 = self.beta
        self.m = b1 * self.m + (1 - b1) * dx
        self.v = b2 * self.v + (1 - b2) * dx * dx
        return x - self.eta * (self.m / (1 - b1 ** self.t)) / (np.sqrt(self.v / (1 - b2 ** self.t)) + self.eps)


class MiniMaxAdaptiveLoss(AbstractAdaptiveLoss):
    """MiniMaxAdaptiveLoss(reweight_every; pde_max_optimiser = Adam(1e-4), bc_max_optimiser = Adam(0.5), ...) — :183-239:
    gradient ASCENT of the weights on the losses (Optimisers.update! with -losses)."""

    def __init__(self, reweight_every: int, pde_max_eta=1.0e-4, bc_max_eta=0.5, pde_loss_weights=1.0, bc_loss_weights=1.0,
                 additional_loss_weights=1.0):
        self.reweight_every = int(reweight_every)
        self.pde_max_eta, self.bc_max_eta = pde_max_eta, bc_max_eta
        self.pde_loss_weights = _vectorify(pde_loss_weights)
        self.bc_loss_weights = _vectorify(bc_loss_weights)
        self.additional_loss_weights = _vectorify(additional_loss_weights)
        self._opt = None

    def reweight(self, theta, pde_losses, bc_losses, iteration, term_grads=None):
        if self._opt is None:
            self._opt = (_Adam(self.pde_max_eta, len(self.pde_loss_weights)), _Adam(self.bc_max_eta, len(self.bc_loss_weights)))
        if not self.fires(iteration):
            return
        self.pde_loss_weights = self._opt[0].update(self.pde_loss_weights, -np.asarray(pde_losses, dtype=np.float64))
        self.bc_loss_weights = self._opt[1].update(self.bc_loss_weights, -np.asarray(bc_losses, dtype=np.float64))


def _softmax(x):
    e = np.exp(x - np.max(x))
    return e / np.sum(e)


class SoftAdaptAdaptiveLoss(AbstractAdaptiveLoss):
    """SoftAdaptAdaptiveLoss(reweight_every; alpha = 0.1, ...) — :284-364: lambda = softmax(alpha * relative loss rate) * N."""

    def __init__(self, reweight_every: int, alpha=0.1, pde_loss_weights=1.0, bc_loss_weights=1.0, additional_loss_weights=1.0):
        self.reweight_every = int(reweight_every)
        self.alpha = float(alpha)
        self.pde_loss_weights = _vectorify(pde_loss_weights)
        self.bc_loss_weights = _vectorify(bc_loss_weights)
        self.additional_loss_weights = _vectorify(additional_loss_weights)
        self._prev = None

    def reweight(self, theta, pde_losses, bc_losses, iteration, term_grads=None):
        allv = np.concatenate([np.asarray(pde_losses, dtype=np.float64), np.asarray(bc_losses, dtype=np.float64)])
        if self._prev is None:                                   # seeded on the very first call (:322-326)
            self._prev = allv.copy()
        if not self.fires(iteration):
            return
        rates = (allv - self._prev) / (self._prev + 1.0e-8)
        w = _softmax(self.alpha * rates) * len(allv)
        n_pde = len(pde_losses)
        self.pde_loss_weights, self.bc_loss_weights = w[:n_pde].copy(), w[n_pde:].copy()
        self._prev = allv.copy()


class ReLoBRaLoAdaptiveLoss(AbstractAdaptiveLoss):
    """ReLoBRaLoAdaptiveLoss(reweight_every; alpha = 1.0, beta = 0.9, ...) — :408-491: softmax of the loss ratio against the
    previous (prob. beta) or the initial checkpoint (random look-back)."""

    def __init__(self, reweight_every: int, alpha=1.0, beta=0.9, pde_loss_weights=1.0, bc_loss_weights=1.0,
                 additional_loss_weights=1.0, rng: Optional[np.random.Generator] = None):
        self.reweight_every = int(reweight_every)
        self.alpha, self.beta = float(alpha), float(beta)
        self.pde_loss_weights = _vectorify(pde_loss_weights)
        self.bc_loss_weights = _vectorify(bc_loss_weights)
        self.additional_loss_weights = _vectorify(additional_loss_weights)
        self.rng = rng or np.random.default_rng()
        self._init = self._prev = None

    def reweight(self, theta, pde_losses, bc_losses, iteration, term_grads=None):
        allv = np.concatenate([np.asarray(pde_losses, dtype=np.float64), np.asarray(bc_losses, dtype=np.float64)])
        if self._init is None:
            self._init, self._prev = allv.copy(), allv.copy()
        if not self.fires(iteration):
            return
        ref = self._prev if self.rng.random() < self.beta else self._init
        w = _softmax(self.alpha * allv / (ref + 1.0e-8)) * len(allv)
        n_pde = len(pde_losses)
        self.pde_loss_weights, self.bc_loss_weights = w[:n_pde].copy(), w[n_pde:].copy()
        self._prev = allv.copy()
