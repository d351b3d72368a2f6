"""Training strategies = collocation point sets (host side).

Mirror of src/training_strategies.jl (GridTraining :13, StochasticTraining :235, QuasiRandomTraining :311)
and of the set / bound builders in src/discretize.jl (generate_training_sets :185-241, get_bounds :299-324).
The reduction `theta -> mean(abs2, residual(set, theta))` (training_strategies.jl:220, 280, 380) itself is
done by the HIP engine; these classes only decide WHICH points it runs on and when they are redrawn.
QuadratureTraining (adaptive CPU cubature, "not GPU compatible", docs/src/manual/training_strategies.md:10-11)
and WeightedIntervalTraining (NNODE only) are out of scope.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .symbolic import PDESystem, VarInfo, get_argument, get_variables


class AbstractTrainingStrategy:
    """Extension point, as `AbstractTrainingStrategy` in src/NeuralPDE.jl:92-134: a strategy provides
    `point_sets(pinnrep) -> (pde_sets, bc_sets, resample)`."""

    def point_sets(self, pde_system: PDESystem, vi: VarInfo, dtype):
        raise NotImplementedError


def _julia_range(lo: float, step: float, hi: float) -> np.ndarray:
    """`lo:step:hi` with Julia's length rule floor((hi-lo)/step + eps) + 1."""
    n = int(np.floor((hi - lo) / step + 1e-10)) + 1
    return lo + step * np.arange(n)


def generate_training_sets(domains, dx, eqs, bcs, dtype, vi: VarInfo):
    """src/discretize.jl:185-241.  Returns [pde_train_sets, bcs_train_sets]; each set is (d_k x N_k),
    columns enumerate Iterators.product with the first variable fastest."""
    dxs = list(dx) if isinstance(dx, (list, tuple, np.ndarray)) else [dx] * len(domains)
    dict_var_span = {str(d.variable): _julia_range(d.domain.lo, s, d.domain.hi) for d, s in zip(domains, dxs)}
    bound_args = get_argument(bcs, vi)
    # Reference quirk, restated faithfully (:202-214): `dif` is filled from `bound_vars = get_variables(bcs)`,
    # which holds only Symbols, so `x isa Number && push!(dif[i], x)` never fires, `dif` stays empty and
    # `setdiff(c, d)` is the identity: in v6.2.2 the pde sets are the FULL grids, boundary values included
    # (the removal shown in docs/src/developer/debugging.md:160-192 is from an older version; that page is
    # marked "not current").
    dict_var_span_ = dict(dict_var_span)

    def product(spans):
        # Iterators.product: first iterator fastest
        cols = list(itertools.product(*[list(s) for s in reversed(spans)]))
        arr = np.array([c[::-1] for c in cols], dtype=dtype).T
        return arr.reshape(len(spans), -1)

    bcs_train_sets = []
    for bt in bound_args:
        spans = [dict_var_span[str(a)] if not isinstance(a, float) else np.array([a]) for a in bt]
        bcs_train_sets.append(product(spans))
    pde_args = get_argument(eqs, vi)
    pde_train_sets = []
    for bt in pde_args:
        spans = [dict_var_span_[str(a)] if not isinstance(a, float) else np.array([a]) for a in bt]
        pde_train_sets.append(product(spans))
    return [pde_train_sets, bcs_train_sets]


def get_bounds(domains, eqs, bcs, dtype, vi: VarInfo, points: int):
    """Non-quadrature bounds, src/discretize.jl:299-324: variables -> [inf + 1/points, sup - 1/points]
    (absolute shrink regardless of domain length), numeric bc arguments -> [c, c]."""
    dx = 1.0 / points
    span = {str(d.variable): (d.domain.lo + dx, d.domain.hi - dx) for d in domains}

    def bounds(args_list):
        res = []
        for args in args_list:
            lb = np.array([span[str(a)][0] if not isinstance(a, float) else a for a in args], dtype=dtype)
            ub = np.array([span[str(a)][1] if not isinstance(a, float) else a for a in args], dtype=dtype)
            res.append((lb, ub))
        return res

    return bounds(get_argument(eqs, vi)), bounds(get_argument(bcs, vi))


@dataclass
class GridTraining(AbstractTrainingStrategy):
    """GridTraining(dx) — src/training_strategies.jl:13; fixed sets built once (:131-160, :215-221)."""
    dx: object

    def point_sets(self, pde_system, vi, dtype):
        pde, bc = generate_training_sets(pde_system.domain, self.dx, pde_system.eqs, pde_system.bcs, dtype, vi)
        return pde, bc, None


def generate_random_points(points: int, bound, dtype, rng: np.random.Generator) -> np.ndarray:
    """src/training_strategies.jl:242-245: rand(T, d, N) .* (ub .- lb) .+ lb."""
    lb, ub = bound
    return (rng.random((len(lb), points)).astype(dtype) * (ub - lb)[:, None] + lb[:, None]).astype(dtype)


@dataclass
class StochasticTraining(AbstractTrainingStrategy):
    """StochasticTraining(points; bcs_points = points) — src/training_strategies.jl:235-240.  A fresh uniform
    sample per loss call (:277-281)."""
    points: int
    bcs_points: Optional[int] = None
    rng: np.random.Generator = field(default_factory=lambda: np.random.default_rng())

    def __post_init__(self):
        if self.bcs_points is None:
            self.bcs_points = self.points

    def point_sets(self, pde_system, vi, dtype):
        pb, bb = get_bounds(pde_system.domain, pde_system.eqs, pde_system.bcs, dtype, vi, self.points)

        def draw():
            return ([generate_random_points(self.points, b, dtype, self.rng) for b in pb],
                    [generate_random_points(self.bcs_points, b, dtype, self.rng) for b in bb])

        pde, bc = draw()
        return pde, bc, draw


class SobolSample:
    """[3P] QuasiMonteCarlo.SobolSample stand-in (scipy.stats.qmc.Sobol, scrambled, seeded)."""

    def __init__(self, seed: int = 0, scramble: bool = True):
        self.seed, self.scramble = seed, scramble
        self._calls = 0

    def sample(self, n, lb, ub, dtype):
        from scipy.stats import qmc
        eng = qmc.Sobol(len(lb), scramble=self.scramble, seed=self.seed + self._calls)
        self._calls += 1
        if not self.scramble:
            # un-randomised sequence: elements 1..n — Sobol.jl (behind QuasiMonteCarlo.SobolSample) never returns the all-zero
            # element 0, and neither does the device sampler (pinn_set_sampler kind 3)
            eng.fast_forward(1)
            u = eng.random(n)
        else:
            m = int(np.ceil(np.log2(max(n, 1))))
            u = eng.random_base2(m)[:n] if (1 << m) >= n else eng.random(n)
        return (lb[:, None] + (ub - lb)[:, None] * u.T).astype(dtype)


class LatinHypercubeSample:
    """[3P] QuasiMonteCarlo.LatinHypercubeSample stand-in — the reference's default sampling_alg
    (src/training_strategies.jl:321)."""

    def __init__(self, seed: Optional[int] = None):
        self.rng = np.random.default_rng(seed)

    def sample(self, n, lb, ub, dtype):
        d = len(lb)
        u = np.empty((d, n))
        for i in range(d):
            u[i] = (self.rng.permutation(n) + self.rng.random(n)) / n
        return (lb[:, None] + (ub - lb)[:, None] * u).astype(dtype)


@dataclass
class QuasiRandomTraining(AbstractTrainingStrategy):
    """QuasiRandomTraining(points; bcs_points, sampling_alg, resampling = true, minibatch = 0) —
    src/training_strategies.jl:311-334.  resampling: new design every call (:375-381); otherwise `minibatch`
    designs generated up front and one picked at random per call (:383-387)."""
    points: int
    bcs_points: Optional[int] = None
    sampling_alg: object = field(default_factory=LatinHypercubeSample)
    resampling: bool = True
    minibatch: int = 0
    rng: np.random.Generator = field(default_factory=lambda: np.random.default_rng())

    def __post_init__(self):
        if self.bcs_points is None:
            self.bcs_points = self.points

    def point_sets(self, pde_system, vi, dtype):
        pb, bb = get_bounds(pde_system.domain, pde_system.eqs, pde_system.bcs, dtype, vi, self.points)
        alg = self.sampling_alg

        def design():
            return ([alg.sample(self.points, lb, ub, dtype) for lb, ub in pb],
                    [alg.sample(self.bcs_points, lb, ub, dtype) for lb, ub in bb])

        if self.resampling:
            pde, bc = design()
            return pde, bc, design
        nb = max(int(self.minibatch), 1)
        batches = [design() for _ in range(nb)]
        if nb == 1:
            return batches[0][0], batches[0][1], None

        def pick():
            return batches[int(self.rng.integers(nb))]

        return batches[0][0], batches[0][1], pick


class QuadratureTraining(AbstractTrainingStrategy):
    """`QuadratureTraining(; quadrature_alg, reltol, abstol, maxiters, batch)` (src/training_strategies.jl:396-481): every term's loss is
    (1/area) * integral of residual^2 over the term's domain.  STAND-IN: the reference integrates adaptively (Integrals.jl cubature,
    HCubatureJL by default) to `reltol`/`abstol`; here the integral is a FIXED tensor Gauss-Legendre rule with `nodes` points per free
    axis (default: 32 in 1-D, 24 in 2-D, 12 in 3-D, 8 in 4-D), evaluated with the other terms in one fused device call
    (`pinn_set_point_weights`).  Same objective, different (non-adaptive) quadrature error; `quadrature_alg`, `reltol`, `abstol`,
    `maxiters`, `batch` are accepted for source compatibility and ignored."""

    def __init__(self, quadrature_alg=None, reltol=1e-6, abstol=1e-3, maxiters=1000, batch=100, nodes: int = None):
        self.quadrature_alg, self.reltol, self.abstol, self.maxiters, self.batch = quadrature_alg, reltol, abstol, maxiters, batch
        self.nodes = nodes
        self._weights = None

    def _rule(self, lb, ub, dtype):
        free = [i for i in range(len(lb)) if ub[i] > lb[i]]
        n = self.nodes or {0: 1, 1: 32, 2: 24, 3: 12}.get(len(free), 8)
        xs, ws = np.polynomial.legendre.leggauss(n)
        grids, wts = [], []
        for i in range(len(lb)):
            if i in free:
                grids.append(0.5 * (ub[i] - lb[i]) * xs + 0.5 * (ub[i] + lb[i]))
                wts.append(0.5 * ws)                                   # weights of the MEAN over the axis (sum to 1)
            else:
                grids.append(np.array([lb[i]]))
                wts.append(np.array([1.0]))
        mesh = np.meshgrid(*grids, indexing="ij")
        wmesh = np.meshgrid(*wts, indexing="ij")
        pts = np.stack([m.reshape(-1, order="F") for m in mesh]).astype(dtype)          # first variable fastest, like the grids
        w = np.prod(np.stack([m.reshape(-1, order="F") for m in wmesh]), axis=0)
        return pts, w / w.sum()

    def point_sets(self, pde_system: PDESystem, vi: VarInfo, dtype):
        # integration domains = the full variable ranges (no 1/points inset: src/discretize.jl get_bounds for QuadratureTraining)
        dom = {str(d.variable): (float(d.domain.lo), float(d.domain.hi)) for d in pde_system.domain}
        args = get_argument(list(pde_system.eqs) + list(pde_system.bcs), vi)
        sets, weights = [], []
        for a in args:
            lb = np.array([float(v) if isinstance(v, (int, float)) else dom[str(v)][0] for v in a])
            ub = np.array([float(v) if isinstance(v, (int, float)) else dom[str(v)][1] for v in a])
            p, w = self._rule(lb, ub, dtype)
            sets.append(p); weights.append(w)
        n_pde = len(pde_system.eqs)
        self._weights = weights
        return sets[:n_pde], sets[n_pde:], None

    def point_weights(self):
        """per term: quadrature weights (sum to 1) of the sets returned by the last point_sets call"""
        return self._weights


# ------------------------------------------------------------------------------------------------
# the reference's strategy plug-in points, by name
# ------------------------------------------------------------------------------------------------
def get_loss_function(init_params_or_pinnrep, loss_function, train_set, eltype=np.float64, strategy=None):
    """`get_loss_function(init_params, loss_function, train_set, eltypeθ, strategy)` (src/training_strategies.jl:213-221):
    the per-term closure `θ -> mean(abs2, loss_function(train_set, θ))` of a datafree residual function `(cord, θ) -> 1 x N`
    (contract: test/Interface/interface__abstract_contracts.jl:53-64)."""
    train_set = np.asarray(train_set, dtype=eltype)

    def loss(theta):
        r = np.asarray(loss_function(train_set, theta), dtype=np.float64)
        return float(np.mean(np.abs(r) ** 2))
    return loss


def merge_strategy_with_loss_function(pinnrep, strategy: AbstractTrainingStrategy, datafree_pde_loss_function, datafree_bc_loss_function):
    """`merge_strategy_with_loss_function(pinnrep, strategy, datafree_pde, datafree_bc)` (src/training_strategies.jl:131-160 Grid,
    247-269 Stochastic, 336-363 QuasiRandom; called at src/discretize.jl:541-545): zips every datafree residual function with the
    point set the strategy provides for it and returns `(pde_loss_functions, bc_loss_functions)`, lists of `θ -> scalar`.
    A custom strategy only has to subclass AbstractTrainingStrategy and implement `point_sets`.  (`symbolic_discretize` itself
    evaluates all terms and the gradient in one fused engine call; these closures are the per-term, value-only view of it.)"""
    pde_sets, bc_sets, resample = strategy.point_sets(pinnrep._pde_system, pinnrep._vi, np.float64)
    if len(pde_sets) != len(datafree_pde_loss_function) or len(bc_sets) != len(datafree_bc_loss_function):
        raise ValueError("the strategy must provide one point set per equation and per boundary condition")

    def wrap(fns, which):
        out = []
        for k, fn in enumerate(fns):
            def loss(theta, fn=fn, k=k):
                sets = resample()[which] if resample is not None else (pde_sets, bc_sets)[which]    # fresh sets every call (:277-281, :375-381)
                r = np.asarray(fn(sets[k], theta), dtype=np.float64)
                return float(np.mean(np.abs(r) ** 2))
            out.append(loss)
        return out
    return wrap(datafree_pde_loss_function, 0), wrap(datafree_bc_loss_function, 1)
