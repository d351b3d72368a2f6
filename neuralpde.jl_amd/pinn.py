"""PhysicsInformedNN / discretize / symbolic_discretize — host-side mirror of the reference's discretizer
API for the PINN hot path, on top of the HIP engine.

Reference surface mirrored (same names, argument meaning, error behaviour):
  Chain / Dense                       [3P] Lux (only Dense chains; DGM etc. are out of scope)
  Phi                                 src/pinn_types.jl:79-90
  PhysicsInformedNN(chain, strategy; init_params, param_estim, additional_loss, adaptive_loss, logger,
                    log_options, iteration)                                 src/pinn_types.jl:147-211
  NonAdaptiveLoss                     src/adaptive_losses.jl:22-42
  symbolic_discretize -> PINNRepresentation   src/discretize.jl:413-767, src/pinn_types.jl:257-440
  discretize -> OptimizationProblem   src/discretize.jl:776-780
What the reference does with RuntimeGeneratedFunctions + Zygote per iteration is one `pinn_loss_grad`
call into libpinn_hip.so here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .ir import NetIR, ProblemIR, TermIR
from .strategies import AbstractTrainingStrategy
from .symbolic import Equation, PDESystem, VarInfo, get_vars, lower_equation


# ------------------------------------------------------------------------------------------------
# chains
# ------------------------------------------------------------------------------------------------
@dataclass
class Dense:
    """Lux.Dense(in => out, activation)."""
    n_in: int
    n_out: int
    activation: str = "identity"


@dataclass
class PeriodicEmbedding:
    """[3P] Boltz.Layers.PeriodicEmbedding(idxs, periods) as the FIRST layer of a Chain (reference:
    test/CUDA/nnpde_cuda__1d_pde_dirichlet_bc_cuda.jl:26,48): inputs `idxs` (1-based, as in Julia) leave the input list and come back at
    its end as sin(2 pi x / period) for all of them, then cos(2 pi x / period) for all of them; the other inputs pass through in order."""
    idxs: Sequence[int]
    periods: Sequence[float]


class Chain:
    """Lux.Chain of Dense layers, optionally behind a PeriodicEmbedding.  The engine supports the shape every PINN chain in the
    reference's PDE tests has: tanh / sigmoid (per layer) or sin (all layers) on the hidden layers, identity on the last, single output."""

    def __init__(self, *layers):
        self.embed = ()
        if layers and isinstance(layers[0], PeriodicEmbedding):
            emb, layers = layers[0], layers[1:]
            if len(emb.idxs) != len(emb.periods) or not emb.idxs:
                raise ValueError("PeriodicEmbedding needs one period per embedded input")
            if len(set(emb.idxs)) != len(emb.idxs) or min(emb.idxs) < 1 or any(not (p > 0) for p in emb.periods):
                raise ValueError("PeriodicEmbedding: distinct 1-based input indices and positive periods")
            self.embed = tuple((int(i) - 1, float(p)) for i, p in zip(emb.idxs, emb.periods))
        if any(isinstance(l, PeriodicEmbedding) for l in layers):
            raise ValueError("PeriodicEmbedding is supported as the first layer of a Chain only")
        if not layers:
            raise ValueError("Chain needs at least one layer")
        for a, b in zip(layers[:-1], layers[1:]):
            if a.n_out != b.n_in:
                raise ValueError("DimensionMismatch: consecutive Dense layers do not chain")
        self.layers = list(layers)
        self.sizes = tuple([layers[0].n_in] + [l.n_out for l in layers])
        self.n_inputs = self.sizes[0] - len(self.embed)          # arguments of the dependent variable (features = sizes[0])
        if self.embed and (self.n_inputs < len(self.embed) or max(i for i, _ in self.embed) >= self.n_inputs):
            raise ValueError("DimensionMismatch: the first Dense layer must take (inputs + number of embedded inputs) features")
        if len(layers) < 2:
            raise ValueError("the HIP engine needs at least one hidden layer")
        if layers[-1].activation != "identity":
            raise ValueError("the last layer must have identity activation")
        alias = {"σ": "sigmoid", "sigmoid_fast": "sigmoid", "tanh_fast": "tanh"}
        acts = [alias.get(l.activation, l.activation) for l in layers[:-1]]
        if len(set(acts)) == 1:
            self.act = acts[0]
        else:
            # per-layer activations (e.g. the reference's Dense(1, n, tanh), Dense(n, n, σ), Dense(n, 1)): tanh and sigmoid may be
            # mixed (kernel variant ACT_MIXED, compiled for the small-net shapes); sin has kernels of its own and cannot be mixed
            if not set(acts) <= {"tanh", "sigmoid"}:
                raise ValueError("the HIP engine mixes only tanh and sigmoid inside one chain (got " + ", ".join(acts) + ")")
            self.act = ",".join(acts)

    @property
    def nparams(self) -> int:
        return sum(l.n_in * l.n_out + l.n_out for l in self.layers)


class DGM:
    """DGM(in_dims, out_dims, modes, layers, activation1, activation2, out_activation) — the Deep Galerkin architecture of the reference
    (src/dgm.jl:97-115): S1 = a1(W1 x + b1); `layers` gated layers Z, G, R = a1(U x + W S + b), H = a2(Uh x + Wh (S .* R) + bh),
    S' = (1 - G) .* H + Z .* S (DGMLSTMLayer, :40-48); f = W S + b.  The engine runs it with its own kernel family (csrc/pinn_kernels3.hpp):
    single output, identity output activation, activations tanh / sigmoid / sin, up to 64 modes."""

    def __init__(self, in_dims: int, out_dims: int, modes: int, layers: int, activation1="tanh", activation2="tanh", out_activation="identity"):
        alias = {"σ": "sigmoid", "sigmoid_fast": "sigmoid", "tanh_fast": "tanh"}
        a1, a2 = alias.get(activation1, activation1), alias.get(activation2, activation2)
        if out_dims != 1:
            raise ValueError("each network must have a single output (one network per dependent variable, src/pinn_types.jl:106-108)")
        if out_activation != "identity":
            raise ValueError("the HIP engine runs DGM networks with the identity output activation (the reference's examples use it)")
        if not {a1, a2} <= {"tanh", "sigmoid", "sin"}:
            raise ValueError(f"unsupported DGM activations ({a1}, {a2}); supported: tanh, sigmoid, sin")
        if not (1 <= modes <= 64 and 1 <= layers <= 8):
            raise ValueError("DGM networks are supported with 1..64 modes and 1..8 gated layers")
        self.in_dims, self.modes, self.dgm_layers = in_dims, modes, layers
        self.sizes = (in_dims, modes, 1)
        self.act = f"dgm,{a1},{a2},{layers}"
        self.layers = None

    @property
    def nparams(self) -> int:
        d, M = self.in_dims, self.modes
        return M * d + M + self.dgm_layers * (4 * M * d + 4 * M * M + 4 * M) + M + 1


def DeepGalerkin(in_dims: int, out_dims: int, modes: int, L: int, activation1, activation2, out_activation, strategy, **kwargs):
    """DeepGalerkin(in_dims, out_dims, modes, L, activation1, activation2, out_activation, strategy; kwargs...) — src/dgm.jl:143-152:
    a PhysicsInformedNN over the DGM architecture."""
    return PhysicsInformedNN(DGM(in_dims, out_dims, modes, L, activation1, activation2, out_activation), strategy, **kwargs)


def initialparameters(rng: np.random.Generator, chain: Chain, dtype=np.float64, init: str = "lux1") -> np.ndarray:
    """[3P] Lux.initialparameters for a Chain of Dense layers, flattened in ComponentArrays order [W1 (out x in, column-major) | b1 | ...].
    init = "lux1" (default): the Dense defaults of the Lux 1.x line the reference pins (Project.toml: Lux 1.31) —
    `init_weight = kaiming_uniform(gain = 1/sqrt(3))` and `init_bias = U(-1/sqrt(fan_in), 1/sqrt(fan_in))`, i.e. both drawn from
    U(+-1/sqrt(fan_in)) (the PyTorch convention Lux adopted in 1.0; the Lux sources are not part of the reference tree, so this
    is restated from its documentation);
    init = "glorot": glorot_uniform weights and zero bias (Lux < 1.0, and what `dgm.jl:12` still passes explicitly).
    Only the distribution of the random start matters here: a user-supplied `init_params` bypasses this function."""
    parts = []
    if isinstance(chain, DGM):
        # Dense layers: the Lux Dense defaults; gated layers: glorot_uniform weights, zero biases (src/dgm.jl:10-13)
        d, M = chain.in_dims, chain.modes
        lim = 1.0 / math.sqrt(d)
        parts += [rng.uniform(-lim, lim, size=(M, d)).T.reshape(-1), rng.uniform(-lim, lim, size=M)]
        for _ in range(chain.dgm_layers):
            for nin in (d, d, d, d, M, M, M, M):
                g = math.sqrt(6.0 / (nin + M))
                parts.append(rng.uniform(-g, g, size=(M, nin)).T.reshape(-1))
            parts.append(np.zeros(4 * M))
        lim = 1.0 / math.sqrt(M)
        parts += [rng.uniform(-lim, lim, size=(1, M)).reshape(-1), rng.uniform(-lim, lim, size=1)]
        return np.concatenate(parts).astype(dtype)
    for l in chain.layers:
        if init == "glorot":
            lim = math.sqrt(6.0 / (l.n_in + l.n_out))
            W, b = rng.uniform(-lim, lim, size=(l.n_out, l.n_in)), np.zeros(l.n_out)
        elif init == "lux1":
            lim = 1.0 / math.sqrt(l.n_in)
            W, b = rng.uniform(-lim, lim, size=(l.n_out, l.n_in)), rng.uniform(-lim, lim, size=l.n_out)
        else:
            raise ValueError("init must be 'lux1' or 'glorot'")
        parts += [W.T.reshape(-1), b]
    return np.concatenate(parts).astype(dtype)


# ------------------------------------------------------------------------------------------------
# discretizer types
# ------------------------------------------------------------------------------------------------
@dataclass
class LogOptions:
    """src/pinn_types.jl:7-17."""
    log_frequency: int = 50


from .adaptive import (AbstractAdaptiveLoss, GradientScaleAdaptiveLoss, MiniMaxAdaptiveLoss, NonAdaptiveLoss,  # noqa: E402
                       ReLoBRaLoAdaptiveLoss, SoftAdaptAdaptiveLoss)


@dataclass
class DataLoss:
    """EXTENSION (not in the reference's API): a data-misfit term  weight * mean(abs2, u(points) - values)  evaluated ON THE DEVICE
    with the physics terms, gradient included.  It is what the reference's users write as `additional_loss(phi, theta, p)` for
    inverse problems (docs/src/tutorials/param_estim.md:79-95, test/NNPDE2/additional_loss__lorenz_system.jl:66-77); keeping it
    in the fused evaluation lets `solve(prob, Adam)` run the whole inverse problem without leaving HBM.
    depvar: the dependent variable (e.g. `u` or `u(t, x)`); points: (d x N) in the order of that variable's arguments; values: N."""
    depvar: object
    points: np.ndarray
    values: np.ndarray
    weight: float = 1.0


class PhysicsInformedNN:
    """PhysicsInformedNN(chain, strategy; ...) — src/pinn_types.jl:165-211.
    `chain` is one Chain or a list with one single-output Chain per dependent variable (:106-108)."""

    def __init__(self, chain, strategy: AbstractTrainingStrategy, *, init_params=None, phi=None, derivative=None,
                 param_estim: bool = False, additional_loss: Optional[Callable] = None, adaptive_loss=None,
                 logger=None, log_options: LogOptions = LogOptions(), iteration=None, data_loss: Sequence[DataLoss] = (),
                 precision: str = "auto", **kwargs):
        if phi is not None or derivative is not None:
            raise ValueError("custom `phi` / `derivative` closures are per-call Julia hooks (src/pinn_types.jl:166-167) "
                             "and cannot be fused into the HIP kernels; they are not supported by this backend")
        self.chain = list(chain) if isinstance(chain, (list, tuple)) else [chain]
        self.multioutput = isinstance(chain, (list, tuple))
        if precision not in ("auto", "f32", "f64"):
            raise ValueError('precision must be "auto" (compute dtype = eltype(theta), the reference\'s contract, src/eltype_matching.jl:8-10), "f32" or "f64"')
        # PRECISION POLICY (r06) = the reference's: the compute dtype follows eltype(theta) (src/eltype_matching.jl:8-10,
        # src/discretize.jl:432-449: Float64 unless the user passes Float32 init_params).
        #   "auto" (default): Float64 parameters (incl. init_params = None) -> the engine's float64 evaluation mode (pinn_set_option(h,
        #           "precision", "f64"): objective, gradient, every public closure and the BFGS / L-BFGS stages in double, points handed
        #           over in double); Float32 parameters -> the fp32 kernels.  A problem the float64 kernels do not cover (DGM networks,
        #           general mixed derivatives of order >= 3; periodic embeddings are covered since r06) FAILS at discretize time with the reason — never a
        #           silent narrowing;
        #   "f32":  the explicit fast opt-in — fp32 kernels (7-8x faster on the matrix pipe) whatever eltype(theta); parameters and
        #           results are converted at the boundary;
        #   "f64":  the float64 mode whatever eltype(theta).
        self.precision = precision
        self.strategy = strategy
        self.init_params = init_params
        self.param_estim = param_estim
        self.additional_loss = additional_loss
        self.data_loss = list(data_loss)
        self.adaptive_loss = adaptive_loss
        self.logger = logger
        self.log_options = log_options
        if iteration is None:                      # src/pinn_types.jl:195-204
            self.iteration, self.self_increment = [1], True
        else:
            self.iteration, self.self_increment = iteration, False
        self.kwargs = kwargs                       # stored, never used — as in the reference (:162, :209)
        self.phi = None                            # filled by symbolic_discretize


class Phi:
    """Trial function handle: `phi(x, theta)` evaluates the network (src/pinn_types.jl:88-90).  x: (d,) or (d x N)."""

    def __init__(self, engine: "_lib.Engine", net: int, theta_slice: slice, d: int):
        self.engine, self.net, self.d, self.theta_slice = engine, net, d, theta_slice

    def __call__(self, x, theta):
        """theta: the whole flat vector, or only this network's own parameters — the reference's `phi[i](x, res.u.depvar.u_i)`
        (docs/src/tutorials/systems.md:112-121)."""
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        pts = x.reshape(self.d, -1) if not single else x.reshape(self.d, 1)
        th = np.asarray(theta).reshape(-1)
        n_own = self.theta_slice.stop - self.theta_slice.start
        if th.size == n_own and th.size != self.engine.P:
            full = np.zeros(self.engine.P, dtype=th.dtype)
            full[self.theta_slice] = th
            th = full
        # a handle in float64 mode evaluates the trial function in double end to end (pinn_phi_f64; src/pinn_types.jl:88-90 computes in eltype(theta))
        if self.engine.get_option("precision") == "f64":
            out = self.engine.phi_f64(self.net, th, pts).reshape(1, -1)
        else:
            out = self.engine.phi(self.net, th, pts).astype(np.float64).reshape(1, -1)
        return out[:, 0] if single else out


def depvar_params(rep, theta, name):
    """`theta.depvar.<name>` of the reference's ComponentArray (src/discretize.jl:451-465): the parameters of one dependent variable's
    network out of the flat vector."""
    return np.asarray(theta)[rep.net_slices[rep.depvars.index(str(name))]]


@dataclass
class PINNLossFunctions:
    """src/pinn_types.jl:414-440."""
    bc_loss_functions: List[Callable]
    pde_loss_functions: List[Callable]
    full_loss_function: Callable
    additional_loss_function: Optional[Callable]
    datafree_pde_loss_functions: List[Callable]
    datafree_bc_loss_functions: List[Callable]
    data_loss_functions: List[Callable] = field(default_factory=list)      # DataLoss extension


@dataclass
class PINNRepresentation:
    """src/pinn_types.jl:257-403 (the fields the hot path uses)."""
    eqs: list
    bcs: list
    domains: list
    eq_params: tuple
    defaults: Optional[dict]
    default_p: Optional[np.ndarray]
    param_estim: bool
    additional_loss: Optional[Callable]
    adaloss: NonAdaptiveLoss
    depvars: list
    indvars: list
    dict_indvars: dict
    dict_depvars: dict
    dict_depvar_input: dict
    logger: object
    multioutput: bool
    iteration: list
    init_params: np.ndarray
    flat_init_params: np.ndarray
    phi: object
    strategy: object
    pde_indvars: list
    bc_indvars: list
    symbolic_pde_loss_functions: List[TermIR]
    symbolic_bc_loss_functions: List[TermIR]
    loss_functions: Optional[PINNLossFunctions] = None
    ir: Optional[ProblemIR] = None
    net_slices: Optional[list] = None               # per dependent variable: its slice of the flat theta (theta.depvar.<name>)
    engine: Optional[object] = None
    pde_train_sets: Optional[list] = None
    bcs_train_sets: Optional[list] = None


class OptimizationFunction:
    """[3P] SciMLBase.OptimizationFunction(f, AutoZygote()) stand-in with an explicit gradient."""

    def __init__(self, f: Callable, value_and_grad: Callable):
        self.f, self.value_and_grad = f, value_and_grad

    def __call__(self, theta, p=None):
        return self.f(theta, p)

    def grad(self, theta, p=None):
        return self.value_and_grad(theta)[1]


@dataclass
class OptimizationProblem:
    """[3P] SciMLBase.OptimizationProblem(f, u0)."""
    f: OptimizationFunction
    u0: np.ndarray
    p: object = None


@dataclass
class Adam:
    """[3P] Optimisers.Adam(eta, (beta1, beta2), eps)."""
    eta: float = 0.001
    beta: tuple = (0.9, 0.999)
    epsilon: float = 1e-8


@dataclass
class BFGS:
    """[3P] OptimizationOptimJL.BFGS() — the quasi-Newton finisher of the reference's tests (e.g. test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:86).
    Mirror: scipy's BFGS on the HOST over the engine's fused `value_and_grad` (one device evaluation per objective call, float64 iterates,
    fp32 evaluation); needs a fixed objective, i.e. fixed point sets — as the reference's own comments require (`resampling = false`)."""
    gtol: float = 1e-8


@dataclass
class LBFGS:
    """[3P] OptimizationOptimJL.LBFGS(): scipy's L-BFGS-B on the host, see BFGS."""
    m: int = 10
    gtol: float = 1e-8


@dataclass
class OptimizationSolution:
    u: np.ndarray
    objective: float
    losses: np.ndarray


def solve(prob: OptimizationProblem, alg: Adam, maxiters: int = 1000, callback: Optional[Callable] = None) -> OptimizationSolution:
    """`solve(prob, Adam(eta); maxiters, callback)` ([3P] Optimization.jl, test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:83-85) on
    the engine's resident-theta Adam loop (pinn_adam_steps).  StochasticTraining redraws its points on the device before
    every step; fixed sets stay installed.  Without a callback the whole loop runs without host synchronisation; with a
    callback `(state, loss) -> stop::Bool` it runs in chunks of 50 steps."""
    rep = prob.pinnrep
    eng = rep.engine
    eng_resample = getattr(rep, "_device_samplers", None)
    if isinstance(alg, (BFGS, LBFGS)):
        return _host_quasi_newton(prob, alg, maxiters, callback)
    if eng_resample and not rep._state.get("samplers_installed"):
        # installed ONCE per discretisation: a second solve (the resume idiom solve(remake(prob, u0 = res.u))) continues the
        # device draw counters instead of replaying the first run's point sequence
        for k, (lb, ub, n, seed, kind) in eng_resample.items():
            eng.set_sampler(k, lb, ub, n, seed, kind)
        rep._state["samplers_installed"] = True
    if rep.additional_loss is not None or (not eng_resample and rep._state.get("resample") is not None):
        # host-side pieces per iteration — a Python `additional_loss(phi, theta, p)` (src/discretize.jl:590-598) or pre-generated
        # designs picked at random per call (`resampling = false, minibatch > 1`, src/training_strategies.jl:383-387): the optimiser
        # loop runs on the host, every iteration is still ONE fused device evaluation (value_and_grad)
        return _host_adam(prob, alg, maxiters, callback)
    theta, losses, done, init = np.asarray(prob.u0, dtype=np.float64), [], 0, True
    ada = rep.adaloss
    n_pde = len(rep.eqs)
    chunk = maxiters if callback is None else 50
    if ada.reweight_every > 0:                     # adaptive weights: reweight on the host between chunks of device steps
        chunk = min(chunk, ada.reweight_every)
    # precision = "f64": theta crosses the ABI in double (pinn_adam_init_f64 / pinn_adam_get_f64), so a chunked run continues from the
    # exact state; otherwise float32, the device dtype of the north star
    f64 = eng.get_option("precision") == "f64"
    step = eng.adam_f64 if f64 else eng.adam
    th32 = theta.astype(np.float64 if f64 else np.float32)
    while done < maxiters:
        n = min(chunk, maxiters - done)
        if ada.reweight_every > 0:                 # end the chunk where the iteration counter hits a multiple of reweight_every
            it = rep.iteration[0] + done
            n = min(n, ada.reweight_every - (it % ada.reweight_every))
        th32, hist = step(th32, n, alg.eta, rep._weights_now(), alg.beta[0], alg.beta[1], alg.epsilon, init=init)
        init = False
        losses.append(hist)
        done += n
        if ada.reweight_every > 0:
            tl, _ = (eng.loss_grad_f64 if f64 else eng.loss_grad)(th32, None, want_grad=False)
            ada.reweight(th32, tl[:n_pde], tl[n_pde:n_pde + len(rep.bcs)], rep.iteration[0] + done,
                         term_grads=lambda: (eng.term_grads_f64 if f64 else eng.term_grads)(th32)[1])
        if callback is not None and callback({"iter": done, "u": th32.astype(np.float64)}, float(hist[-1])):
            break
    losses = np.concatenate(losses)
    rep.iteration[0] += done
    return OptimizationSolution(th32.astype(prob.u0.dtype), float(losses[-1]), losses)


def _host_adam(prob: OptimizationProblem, alg: Adam, maxiters: int, callback) -> OptimizationSolution:
    """[3P] Optimisers.Adam on the host over `prob.f.value_and_grad` (the objective of src/discretize.jl:567-598 incl. a host-side
    `additional_loss`, which must return `(value, gradient)` here: the Python mirror has no AD to differentiate a bare value)."""
    rep = prob.pinnrep
    theta = np.asarray(prob.u0, dtype=np.float64).copy()
    m, v, losses = np.zeros_like(theta), np.zeros_like(theta), []
    b1, b2 = alg.beta
    for it in range(1, maxiters + 1):
        val, g = prob.f.value_and_grad(theta)
        if rep.additional_loss is not None and rep._state.get("add_grad_missing"):
            raise TypeError("solve(): additional_loss must return (value, gradient w.r.t. the flat theta) for training on this backend")
        losses.append(val)
        g = np.asarray(g, dtype=np.float64)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        theta = theta - alg.eta * (m / (1 - b1 ** it)) / (np.sqrt(v / (1 - b2 ** it)) + alg.epsilon)
        if callback is not None and callback({"iter": it, "u": theta.copy()}, float(val)):
            break
    return OptimizationSolution(theta.astype(prob.u0.dtype), float(losses[-1]), np.array(losses))


def _host_quasi_newton(prob: OptimizationProblem, alg, maxiters: int, callback) -> OptimizationSolution:
    """BFGS / LBFGS of the reference's test scripts: scipy.optimize.minimize on the host, every objective call = ONE fused device
    evaluation through `prob.f.value_and_grad`.  The objective must not change between calls (fixed point sets)."""
    from scipy.optimize import minimize
    rep = prob.pinnrep
    if getattr(rep, "_device_samplers", None) or rep._state.get("resample") is not None:
        raise ValueError("BFGS / LBFGS need a fixed objective: use GridTraining, QuadratureTraining or QuasiRandomTraining(...; resampling = false, "
                         "minibatch = 1) (the reference's tests say the same, e.g. test/NNPDE1/nnpde__pde_vi_pde_with_mixed_derivative.jl:76)")
    if isinstance(alg, LBFGS) and callback is None and rep.additional_loss is None and rep.adaloss.reweight_every <= 0:
        # the library's own L-BFGS (pinn_lbfgs): same recursion, no Python in the loop
        theta, hist = rep.engine.lbfgs(prob.u0, int(maxiters), rep._weights_now(), history=alg.m, gtol=alg.gtol)
        rep.iteration[0] += len(hist)
        final = float(hist[-1]) if len(hist) else float(prob.f.value_and_grad(np.asarray(prob.u0, dtype=np.float64))[0])
        return OptimizationSolution(theta.astype(prob.u0.dtype), final, hist if len(hist) else np.array([final]))
    losses, it, last = [], [0], [None]

    def fun(th):
        val, g = prob.f.value_and_grad(th)
        return float(val), np.asarray(g, dtype=np.float64)

    class _Stop(Exception):
        pass

    def cb(th):
        it[0] += 1
        last[0] = np.asarray(th, dtype=np.float64).copy()
        val, _ = fun(th) if callback is not None else (np.nan, None)
        losses.append(val)
        if callback is not None and callback({"iter": it[0], "u": np.asarray(th).copy()}, float(val)):
            raise _Stop
    x0 = np.asarray(prob.u0, dtype=np.float64).copy()
    method, opts = ("BFGS", {"maxiter": int(maxiters), "gtol": alg.gtol}) if isinstance(alg, BFGS) else \
        ("L-BFGS-B", {"maxiter": int(maxiters), "maxcor": alg.m, "gtol": alg.gtol, "ftol": 0.0})
    try:
        out = minimize(fun, x0, jac=True, method=method, callback=cb, options=opts)
        theta, final = out.x, float(out.fun)
    except _Stop:
        theta, final = None, None
    if theta is None:                                   # stopped by the callback: the last iterate it saw
        theta = last[0] if last[0] is not None else x0
    if final is None:
        final = fun(theta)[0]
    rep.iteration[0] += it[0]
    return OptimizationSolution(np.asarray(theta).astype(prob.u0.dtype), final, np.array(losses if callback is not None else [final]))


def remake(prob: OptimizationProblem, u0=None) -> OptimizationProblem:
    """`remake(prob, u0 = res.u)` — the reference's resume idiom (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:84-85)."""
    new = OptimizationProblem(prob.f, prob.u0 if u0 is None else np.asarray(u0), prob.p)
    if hasattr(prob, "pinnrep"):
        new.pinnrep = prob.pinnrep          # the discretisation (engine, point sets, weights) is shared
    return new


# ------------------------------------------------------------------------------------------------
# symbolic_discretize / discretize
# ------------------------------------------------------------------------------------------------
def _broadcast_weights(w, n: int) -> np.ndarray:
    a = np.ones(n) * np.asarray(w, dtype=np.float64)      # errors like the reference if lengths mismatch
    return a


def symbolic_discretize(pde_system: PDESystem, discretization: PhysicsInformedNN) -> PINNRepresentation:
    """src/discretize.jl:413-767."""
    eqs, bcs = list(pde_system.eqs), list(pde_system.bcs)
    if not bcs:
        raise TypeError("MethodError: PDESystem without boundary conditions (the reference fails at solve time, "
                        "test/direct_function__empty_boundary_condition_fails_in_solve_phase.jl:24)")
    vi = get_vars(pde_system.ivs, pde_system.dvs)
    chains = discretization.chain
    if len(chains) != len(vi.depvars):
        raise ValueError(f"{len(vi.depvars)} dependent variables need {len(vi.depvars)} single-output chains "
                         f"(src/pinn_types.jl:106-108); got {len(chains)}")
    for ch in chains:
        if ch.sizes[-1] != 1:
            raise ValueError("each chain must have a single output (one chain per dependent variable)")
    eq_params = tuple(pde_system.ps)
    defaults = pde_system.defaults
    param_estim = discretization.param_estim
    default_p = None
    if eq_params:
        if defaults is None:
            raise ValueError("PDESystem with parameters needs `defaults`")
        default_p = np.array([float(defaults[p]) for p in eq_params], dtype=np.float64)

    # ---- flat init params (src/discretize.jl:432-465): Float64 unless the user passed Float32 ----
    net_offs, o = [], 0
    for ch in chains:
        net_offs.append(o)
        o += ch.nparams
    nnet = o
    if discretization.init_params is None:
        rng = np.random.default_rng()
        flat = np.concatenate([initialparameters(rng, ch, np.float64) for ch in chains])
    else:
        ip = discretization.init_params
        flat = np.concatenate([np.asarray(a).reshape(-1) for a in ip]) if isinstance(ip, (list, tuple)) else np.asarray(ip).reshape(-1)
        if flat.size != nnet:
            raise ValueError(f"init_params has {flat.size} entries, the chains need {nnet}")
    if param_estim and eq_params:
        flat = np.concatenate([flat, default_p.astype(flat.dtype)])      # theta.p block (:457-462)
    dtype = flat.dtype if flat.dtype in (np.float32, np.float64) else np.float64

    # ---- residual IR per equation / bc (src/discretize.jl:505-525) ----
    sym_pde = [lower_equation(eq, vi, eq_params, "pde") for eq in eqs]
    sym_bc = []
    for bc in bcs:
        if bc.lhs.is_Number and bc.rhs.is_Number:
            raise ValueError("ArgumentError: boundary condition without a dependent variable "
                             "(test/direct_function__trivial_bc_0_0_*.jl:44)")
        sym_bc.append(lower_equation(bc, vi, eq_params, "bc"))
    NP = len(eq_params)
    NE = NP if (param_estim and NP) else 0
    # data-misfit terms (DataLoss extension): residual u(x_i) - d_i with d a per-point data channel (descriptor op DATA)
    from .ir import Instr, Slot
    sym_data = []
    for dl in discretization.data_loss:
        name = str(getattr(dl.depvar, "func", dl.depvar))
        if name not in vi.dict_depvars:
            raise ValueError(f"DataLoss: {name} is not a dependent variable of the system")
        inputs = vi.dict_depvar_input[name]
        d_net = len(inputs)
        pts, vals = np.asarray(dl.points, dtype=np.float64), np.asarray(dl.values, dtype=np.float64).reshape(-1)
        if pts.ndim != 2 or pts.shape[0] != d_net or pts.shape[1] != vals.size or vals.size == 0:
            raise ValueError(f"DataLoss for {name}: points must be ({d_net} x N) and values N (N > 0)")
        slot_row = d_net + NP
        sym_data.append(TermIR(dim=d_net, slots=[Slot(vi.dict_depvars[name] - 1, ())],
                               ops=[Instr("DATA", 0, 0, 0.0), Instr("SUB", slot_row, slot_row + 1, 0.0)], out_row=slot_row + 2,
                               indvars=tuple(inputs), kind="data", source=f"{name}(points) - values"))
    terms = sym_pde + sym_bc + sym_data
    ir = ProblemIR(
        ntheta=int(flat.size),
        nets=[NetIR(tuple(ch.sizes), ch.act, off, getattr(ch, "embed", ())) for ch, off in zip(chains, net_offs)],
        terms=terms, nparams=NP, nparams_estim=NE, p_theta_off=nnet,
        p_defaults=list(default_p) if default_p is not None else [],
        param_names=[str(p) for p in eq_params], depvar_names=list(vi.depvars),
        depvar_inputs=[list(vi.dict_depvar_input[n]) for n in vi.depvars])
    # PINN_DESCRIPTOR=2: hand the equations to the library as s-expressions (the form the Julia glue emits; lowered by csrc/sexpr.cpp)
    # instead of the tapes lowered by symbolic.py — both front ends must give the same engine (tests/test_sexpr_frontend.py)
    import os as _os
    # ---- strategy: point sets (src/discretize.jl:541-545); their sizes go into the descriptor as planner hints ----
    strategy = discretization.strategy
    pde_sets, bc_sets, resample = strategy.point_sets(pde_system, vi, dtype)
    n_pde, n_bc = len(eqs), len(bcs)
    hints = [s.shape[1] for s in list(pde_sets) + list(bc_sets)] + [np.asarray(dl.values).size for dl in discretization.data_loss]
    engine = _lib.Engine(ir.to_descriptor2(hints) if _os.environ.get("PINN_DESCRIPTOR") == "2" else ir.to_descriptor(hints))

    def install(pde_sets, bc_sets):
        for k, s in enumerate(list(pde_sets) + list(bc_sets)):
            if s.shape[0] != terms[k].dim:
                raise ValueError(f"point set of term {k} has {s.shape[0]} rows, the term binds {terms[k].dim} variables")
            if f64:
                engine.set_points_f64(k, s)
            else:
                engine.set_points(k, s)

    prec = getattr(discretization, "precision", "auto")
    if prec == "auto":                                  # compute dtype = eltype(theta) (src/eltype_matching.jl:8-10)
        prec = "f32" if dtype == np.float32 else "f64"
    f64 = prec == "f64"
    if f64:
        try:
            engine.set_option("precision", "f64")       # (fails with the reason for what the mode does not cover: DGM, embeddings)
        except _lib.EngineError as e:
            if getattr(discretization, "precision", "auto") != "auto":
                raise
            raise _lib.EngineError(f"{e}\n(the parameters are Float64, so precision = \"auto\" selected the float64 kernels — the reference's contract, "
                                   "src/eltype_matching.jl:8-10.  Pass PhysicsInformedNN(..., precision = \"f32\") to run this problem on the fp32 kernels, "
                                   "or Float32 init_params.)") from None
    install(pde_sets, bc_sets)
    if getattr(strategy, "point_weights", None) is not None and strategy.point_weights() is not None:
        for k, w in enumerate(strategy.point_weights()):           # quadrature strategies: loss_k = sum_i w_i r_i^2
            engine.set_point_weights(k, w)
    n_data = len(sym_data)
    data_sets = [np.asarray(dl.points, dtype=np.float64) for dl in discretization.data_loss]
    for j, dl in enumerate(discretization.data_loss):                    # fixed sets + their observations, installed once
        (engine.set_points_f64 if f64 else engine.set_points)(n_pde + n_bc + j, data_sets[j])
        (engine.set_point_data_f64 if f64 else engine.set_point_data)(n_pde + n_bc + j, np.asarray(dl.values, dtype=np.float64).reshape(1, -1))
    state = {"pde_sets": pde_sets, "bc_sets": bc_sets, "cache_theta": None, "cache": None, "resample": resample}

    adaloss = discretization.adaptive_loss or NonAdaptiveLoss()
    if not isinstance(adaloss, AbstractAdaptiveLoss):
        raise TypeError("adaptive_loss must be an AbstractAdaptiveLoss (src/adaptive_losses.jl)")
    adaloss.broadcast(n_pde, n_bc)                      # scalars -> one weight per term (src/discretize.jl:553-559)
    iteration = discretization.iteration

    def weights_now():
        wd = np.array([float(adaloss.additional_loss_weights[0]) * float(dl.weight) for dl in discretization.data_loss])
        return np.concatenate([adaloss.pde_loss_weights, adaloss.bc_loss_weights, wd]).astype(np.float64)

    def evaluate(theta, want_grad=True, weights=None, redraw=True):
        """one fused engine call; memoised on (theta, weights) so the per-term closures and the full loss share it"""
        th = np.asarray(theta)
        w = weights_now() if weights is None else np.asarray(weights, dtype=np.float64)
        key = (th.tobytes(), w.tobytes())
        # memo: valid for fixed sets, and for resampling strategies when the caller asks for the SAME draw again (redraw = False)
        if (resample is None or not redraw) and state["cache_theta"] == key and (state["cache"][1] is not None or not want_grad):
            return state["cache"]
        if resample is not None and redraw:
            ps, bs = resample()
            state["pde_sets"], state["bc_sets"] = ps, bs
            install(ps, bs)
        losses, grad = engine.loss_grad_f64(th, w, want_grad=want_grad) if f64 else engine.loss_grad(th, w, want_grad=want_grad)
        state["cache_theta"], state["cache"] = key, (losses, grad)
        return losses, grad

    def term_loss(k):
        def f(theta):
            return float(evaluate(theta, want_grad=False)[0][k])
        return f

    def datafree(k):
        def f(cord, theta):
            """(cord, theta) -> 1 x N residual (src/discretize.jl:174)."""
            cord = np.asarray(cord)
            if f64:                                     # Float64 closure (src/pinn_types.jl:435-439; rtol 1e-8 in test/Forward/forward__ode.jl:46-47)
                engine.set_points_f64(k, cord)
                r = engine.residual_f64(k, np.asarray(theta), cord.shape[1]).reshape(1, -1)
                engine.set_points_f64(k, (state["pde_sets"] + state["bc_sets"])[k])
            else:
                engine.set_points(k, cord)
                r = engine.residual(k, np.asarray(theta), cord.shape[1]).astype(np.float64).reshape(1, -1)
                engine.set_points(k, (state["pde_sets"] + state["bc_sets"])[k])
            if getattr(strategy, "point_weights", None) is not None and strategy.point_weights() is not None:
                engine.set_point_weights(k, strategy.point_weights()[k])
            state["cache_theta"] = None
            return r
        return f

    additional_loss = discretization.additional_loss

    def losses_and_reweight(theta, want_grad=False):
        """the part of full_loss_function that runs outside AD (src/discretize.jl:569-580): term losses, iteration
        counter, adaptive reweighting (which may ask the engine for per-term gradients)"""
        losses, _ = evaluate(theta, want_grad=want_grad)
        if discretization.self_increment:
            iteration[0] += 1
        adaloss.reweight(theta, losses[:n_pde], losses[n_pde:n_pde + n_bc], iteration[0],
                         term_grads=lambda: (engine.term_grads_f64 if f64 else engine.term_grads)(np.asarray(theta))[1])
        return losses

    def add_term(theta):
        if additional_loss is None:
            return 0.0, None
        th = np.asarray(theta)
        add = additional_loss(phi, th[:nnet] if param_estim else theta, th[nnet:] if param_estim else None)
        wa = float(np.asarray(adaloss.additional_loss_weights).reshape(-1)[0])
        if isinstance(add, tuple):
            return wa * float(add[0]), wa * np.asarray(add[1])
        state["add_grad_missing"] = True
        return wa * float(add), None

    def value_and_grad(theta):
        # ONE device evaluation per optimiser step: losses and gradient together (the engine always runs the reverse sweep); the
        # memo in `evaluate` serves the gradient below unless the adaptive rule has just changed the weights
        losses = losses_and_reweight(theta, want_grad=True)
        w = weights_now()
        _, grad = evaluate(theta, want_grad=True, weights=w, redraw=False)      # cached, or recomputed under the new weights
        total = float(np.dot(w, losses))
        dt = np.asarray(theta).dtype
        g = grad.astype(dt if dt in (np.float32, np.float64) else np.float64)
        av, ag = add_term(theta)
        if ag is not None:
            g = g + ag
        return total + av, g

    def full_loss_function(theta, p=None):
        """src/discretize.jl:567-598."""
        losses = losses_and_reweight(theta)
        return float(np.dot(weights_now(), losses)) + add_term(theta)[0]

    phis = [Phi(engine, i, slice(net_offs[i], net_offs[i] + chains[i].nparams), getattr(chains[i], "n_inputs", chains[i].sizes[0])) for i in range(len(chains))]
    phi = phis if discretization.multioutput else phis[0]
    discretization.phi = phi

    rep = PINNRepresentation(
        eqs=eqs, bcs=bcs, domains=list(pde_system.domain), eq_params=eq_params, defaults=defaults, default_p=default_p,
        param_estim=param_estim, additional_loss=additional_loss, adaloss=adaloss, depvars=vi.depvars, indvars=vi.indvars,
        dict_indvars=vi.dict_indvars, dict_depvars=vi.dict_depvars, dict_depvar_input=vi.dict_depvar_input,
        logger=discretization.logger, multioutput=discretization.multioutput, iteration=iteration,
        init_params=flat[:nnet], flat_init_params=flat, phi=phi, strategy=strategy,
        pde_indvars=[list(t.indvars) for t in sym_pde], bc_indvars=[list(t.indvars) for t in sym_bc],
        symbolic_pde_loss_functions=sym_pde, symbolic_bc_loss_functions=sym_bc, ir=ir, engine=engine,
        pde_train_sets=pde_sets, bcs_train_sets=bc_sets,
        net_slices=[slice(net_offs[i], net_offs[i] + chains[i].nparams) for i in range(len(chains))])
    rep.loss_functions = PINNLossFunctions(
        bc_loss_functions=[term_loss(n_pde + j) for j in range(n_bc)],
        pde_loss_functions=[term_loss(i) for i in range(n_pde)],
        full_loss_function=full_loss_function, additional_loss_function=additional_loss,
        datafree_pde_loss_functions=[datafree(i) for i in range(n_pde)],
        datafree_bc_loss_functions=[datafree(n_pde + j) for j in range(n_bc)],
        data_loss_functions=[term_loss(n_pde + n_bc + j) for j in range(n_data)])
    rep._value_and_grad = value_and_grad
    rep._weights_now = weights_now
    rep._weights = weights_now()
    # resampling strategies with an on-device counterpart for the resident-theta loop: StochasticTraining (uniform redraw in the
    # same bounds) and QuasiRandomTraining(resampling = true) with LatinHypercubeSample (its default) or SobolSample
    from .strategies import LatinHypercubeSample, QuasiRandomTraining, StochasticTraining, get_bounds
    rep._device_samplers = None
    from .strategies import SobolSample
    kind = 0
    if isinstance(strategy, StochasticTraining):
        kind = 1
    elif isinstance(strategy, QuasiRandomTraining) and strategy.resampling:
        kind = 2 if isinstance(strategy.sampling_alg, LatinHypercubeSample) else (3 if isinstance(strategy.sampling_alg, SobolSample) else 0)
    if kind:
        pb, bb = get_bounds(pde_system.domain, eqs, bcs, np.float64, vi, strategy.points)
        # the device samplers' seeds come from the strategy's own generator (StochasticTraining.rng, LatinHypercubeSample.rng) or seed
        # (SobolSample.seed), so a seeded strategy gives a reproducible run; an unseeded one draws fresh seeds
        alg = getattr(strategy, "sampling_alg", None)
        rng = getattr(strategy, "rng", None) or getattr(alg, "rng", None)
        if rng is None:
            rng = np.random.default_rng(getattr(alg, "seed", None))
        rep._device_samplers = {}
        for k, (lb, ub) in enumerate(list(pb) + list(bb)):
            seed = int(rng.integers(1, 1 << 31))
            if kind == 3 and not strategy.sampling_alg.scramble:
                seed = 0                                    # un-randomised Sobol: the same design on every draw, as in the reference
            rep._device_samplers[k] = (lb, ub, strategy.points if k < n_pde else strategy.bcs_points, seed, kind)
    rep._state = state
    rep._pde_system, rep._vi = pde_system, vi
    return rep


def discretize(pde_system: PDESystem, discretization: PhysicsInformedNN) -> OptimizationProblem:
    """src/discretize.jl:776-780: OptimizationProblem(OptimizationFunction(full_loss_function, AutoZygote()),
    flat_init_params) — with the engine's reverse-mode gradient in place of Zygote."""
    rep = symbolic_discretize(pde_system, discretization)
    f = OptimizationFunction(rep.loss_functions.full_loss_function, rep._value_and_grad)
    prob = OptimizationProblem(f, rep.flat_init_params)
    prob.pinnrep = rep
    return prob
