"""The BASELINE.json configurations as concrete problem statements + synthetic inputs.

Used by bench.py (config 2 is the bench workload) and by the parity tests (all configs, possibly at
reduced point counts).  Synthetic data only: theta ~ glorot-uniform weights / U(+-0.1) biases from
numpy.random.default_rng(1000 + cfg), Sobol / grid / uniform collocation sets as listed below.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import sympy as sp

from .adaptive import NonAdaptiveLoss
from .pinn import Chain, Dense, PhysicsInformedNN
from .strategies import GridTraining, QuasiRandomTraining, SobolSample, StochasticTraining
from .symbolic import Differential, Eq, In, Interval, PDESystem, parameters, variables


@dataclass
class Workload:
    name: str
    pde_system: PDESystem
    chains: List[Chain]
    strategy: object
    theta: np.ndarray
    adaptive_loss: Optional[NonAdaptiveLoss] = None
    param_estim: bool = False
    n_interior: int = 0

    def discretization(self, precision: str = "f32") -> PhysicsInformedNN:
        """the BASELINE configurations are the north star's fp32 workloads: the benchmark harness opts into the fp32 kernels explicitly
        (`precision = "f32"`); pass "auto" (the API default: compute dtype = eltype(theta), i.e. the float64 kernels for these Float64
        parameter vectors) or "f64" for the reference's default eltype"""
        chain = self.chains if len(self.chains) > 1 else self.chains[0]
        return PhysicsInformedNN(chain, self.strategy, init_params=self.theta, adaptive_loss=self.adaptive_loss,
                                 param_estim=self.param_estim, precision=precision)


def mlp(n_in: int, width: int, hidden_layers: int, act: str = "tanh") -> Chain:
    layers = [Dense(n_in, width, act)] + [Dense(width, width, act) for _ in range(hidden_layers - 1)] + [Dense(width, 1)]
    return Chain(*layers)


def synthetic_theta(chains: List[Chain], seed: int, extra: int = 0, dtype=np.float64) -> np.ndarray:
    """weights ~ U(+-sqrt(6/(fan_in+fan_out))), biases ~ U(+-0.1)  (SURVEY.md §8d)."""
    rng = np.random.default_rng(seed)
    parts = []
    for ch in chains:
        for l in ch.layers:
            lim = math.sqrt(6.0 / (l.n_in + l.n_out))
            W = rng.uniform(-lim, lim, size=(l.n_out, l.n_in))
            b = rng.uniform(-0.1, 0.1, size=(l.n_out,))
            parts += [W.T.reshape(-1), b]
    if extra:
        parts.append(np.ones(extra))
    return np.concatenate(parts).astype(dtype)


def cfg1_poisson1d(points: int = 1024, dtype=np.float64) -> Workload:
    """1-D Poisson u'' = -pi^2 sin(pi x), u(0)=u(1)=0, 3x32 tanh MLP, GridTraining with `points` grid nodes
    (dx = 1/(points-1): the reference's grid includes both end points, discretize.jl:202-214)."""
    (x,) = parameters("x")
    (u,) = variables("u")
    Dxx = Differential(x) ** 2
    eq = Eq(Dxx(u(x)), -sp.pi ** 2 * sp.sin(sp.pi * x))
    bcs = [Eq(u(0.0), 0.0), Eq(u(1.0), 0.0)]
    sysm = PDESystem([eq], bcs, [In(x, Interval(0.0, 1.0))], [x], [u(x)])
    chain = mlp(1, 32, 3)
    return Workload("cfg1_poisson1d_3x32_grid", sysm, [chain], GridTraining(1.0 / (points - 1)),
                    synthetic_theta([chain], 1001, dtype=dtype), n_interior=points)


def cfg2_poisson2d(points: int = 65536, bcs_points: Optional[int] = None, width: int = 64, hidden: int = 4,
                   dtype=np.float64) -> Workload:
    """2-D Poisson on the unit square exactly as test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:65-76, 4x64 tanh MLP,
    QuasiRandomTraining(points) with a fixed scrambled-Sobol design (resampling = false, as the reference's own
    deterministic tests do, test/NNPDE1/nnpde__pde_vi_pde_with_mixed_derivative.jl:78-80)."""
    x, y = parameters("x y")
    (u,) = variables("u")
    Dxx, Dyy = Differential(x) ** 2, Differential(y) ** 2
    eq = Eq(Dxx(u(x, y)) + Dyy(u(x, y)), -sp.sin(sp.pi * x) * sp.sin(sp.pi * y))
    bcs = [Eq(u(0, y), 0.0), Eq(u(1, y), 0.0), Eq(u(x, 0), 0.0), Eq(u(x, 1), 0.0)]
    dom = [In(x, Interval(0.0, 1.0)), In(y, Interval(0.0, 1.0))]
    sysm = PDESystem([eq], bcs, dom, [x, y], [u(x, y)])
    chain = mlp(2, width, hidden)
    strat = QuasiRandomTraining(points, bcs_points=bcs_points, sampling_alg=SobolSample(seed=1002), resampling=False, minibatch=1)
    return Workload("cfg2_poisson2d_4x64_quasirandom", sysm, [chain], strat, synthetic_theta([chain], 1002, dtype=dtype),
                    n_interior=points)


def cfg3_burgers(points: int = 262144, bcs_points: Optional[int] = None, dtype=np.float64) -> Workload:
    """Burgers u_t + u u_x - (0.01/pi) u_xx = 0 (docs/src/tutorials/low_level.md:27-32), (t,x) in [0,1]x[-1,1],
    u(0,x) = -sin(pi x), u(t,-1) = u(t,1) = 0; 4x64 tanh MLP."""
    t, x = parameters("t x")
    (u,) = variables("u")
    Dt, Dx, Dxx = Differential(t), Differential(x), Differential(x) ** 2
    eq = Eq(Dt(u(t, x)) + u(t, x) * Dx(u(t, x)) - (0.01 / sp.pi) * Dxx(u(t, x)), 0)
    bcs = [Eq(u(0, x), -sp.sin(sp.pi * x)), Eq(u(t, -1), 0.0), Eq(u(t, 1), 0.0)]
    dom = [In(t, Interval(0.0, 1.0)), In(x, Interval(-1.0, 1.0))]
    sysm = PDESystem([eq], bcs, dom, [t, x], [u(t, x)])
    chain = mlp(2, 64, 4)
    strat = QuasiRandomTraining(points, bcs_points=bcs_points, sampling_alg=SobolSample(seed=1003), resampling=False, minibatch=1)
    return Workload("cfg3_burgers_4x64_quasirandom", sysm, [chain], strat, synthetic_theta([chain], 1003, dtype=dtype),
                    n_interior=points)


def cfg4_cavity(points: int = 262144, bcs_points: int = 32768, width: int = 128, hidden: int = 5, dtype=np.float64) -> Workload:
    """Steady lid-driven cavity, Re = 100 (nu = 0.01): u u_x + v u_y + p_x - nu (u_xx + u_yy) = 0, same for v with p_y,
    u_x + v_y = 0; 8 Dirichlet terms (u, v on four walls, lid u(x,1) = 1); three single-output nets (u, v, p) — the only
    way the reference expresses several dependent variables; bc weights 10 (NonAdaptiveLoss)."""
    x, y = parameters("x y")
    u, v, p = variables("u v p")
    Dx, Dy = Differential(x), Differential(y)
    Dxx, Dyy = Dx ** 2, Dy ** 2
    nu = 0.01
    U, V, Pp = u(x, y), v(x, y), p(x, y)
    eqs = [Eq(U * Dx(U) + V * Dy(U) + Dx(Pp) - nu * (Dxx(U) + Dyy(U)), 0),
           Eq(U * Dx(V) + V * Dy(V) + Dy(Pp) - nu * (Dxx(V) + Dyy(V)), 0),
           Eq(Dx(U) + Dy(V), 0)]
    bcs = [Eq(u(0, y), 0.0), Eq(u(1, y), 0.0), Eq(u(x, 0), 0.0), Eq(u(x, 1), 1.0),
           Eq(v(0, y), 0.0), Eq(v(1, y), 0.0), Eq(v(x, 0), 0.0), Eq(v(x, 1), 0.0)]
    dom = [In(x, Interval(0.0, 1.0)), In(y, Interval(0.0, 1.0))]
    sysm = PDESystem(eqs, bcs, dom, [x, y], [U, V, Pp])
    chains = [mlp(2, width, hidden) for _ in range(3)]
    strat = QuasiRandomTraining(points, bcs_points=bcs_points, sampling_alg=SobolSample(seed=1004), resampling=False, minibatch=1)
    return Workload(f"cfg4_cavity_3x({hidden}x{width})_quasirandom", sysm, chains, strat, synthetic_theta(chains, 1004, dtype=dtype),
                    adaptive_loss=NonAdaptiveLoss(pde_loss_weights=1.0, bc_loss_weights=10.0), n_interior=points)


def cfg5_heat_inverse(points: int = 1000000, bcs_points: int = 65536, width: int = 128, hidden: int = 6, seed: int = 1005,
                      dtype=np.float64) -> Workload:
    """3-D heat equation inverse problem u_t = kappa (u_xx + u_yy + u_zz) on (t,x,y,z) in [0,1]^4 with kappa estimated
    (param_estim = true, initial 1.0; SURVEY.md N2: the reference expresses it as PhysicsInformedNN(...; param_estim) +
    StochasticTraining, docs/src/tutorials/param_estim.md:97-102), IC u(0,x,y,z) = sin(pi x) sin(pi y) sin(pi z), six
    homogeneous wall conditions; 6x128 tanh MLP; StochasticTraining(points; bcs_points) redrawn on every call.
    The data-misfit term of the reference set-up is a host-side `additional_loss` and is not part of this workload."""
    t, x, y, z = parameters("t x y z")
    (u,) = variables("u")
    (kappa,) = parameters("kappa")
    Dt = Differential(t)
    Dxx, Dyy, Dzz = Differential(x) ** 2, Differential(y) ** 2, Differential(z) ** 2
    U = u(t, x, y, z)
    eq = Eq(Dt(U), kappa * (Dxx(U) + Dyy(U) + Dzz(U)))
    bcs = [Eq(u(0, x, y, z), sp.sin(sp.pi * x) * sp.sin(sp.pi * y) * sp.sin(sp.pi * z)),
           Eq(u(t, 0, y, z), 0.0), Eq(u(t, 1, y, z), 0.0), Eq(u(t, x, 0, z), 0.0), Eq(u(t, x, 1, z), 0.0),
           Eq(u(t, x, y, 0), 0.0), Eq(u(t, x, y, 1), 0.0)]
    dom = [In(v, Interval(0.0, 1.0)) for v in (t, x, y, z)]
    sysm = PDESystem([eq], bcs, dom, [t, x, y, z], [U], ps=[kappa], defaults={kappa: 1.0})
    chain = mlp(4, width, hidden)
    strat = StochasticTraining(points, bcs_points=bcs_points, rng=np.random.default_rng(seed))
    return Workload(f"cfg5_heat_inverse_{hidden}x{width}_stochastic", sysm, [chain], strat, synthetic_theta([chain], seed, dtype=dtype),
                    param_estim=True, n_interior=points)


CONFIGS = {"cfg1": cfg1_poisson1d, "cfg2": cfg2_poisson2d, "cfg3": cfg3_burgers, "cfg4": cfg4_cavity, "cfg5": cfg5_heat_inverse}
