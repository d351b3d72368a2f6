"""neuralpde.jl_amd — MI355X-native engine for NeuralPDE.jl's PhysicsInformedNN/discretize hot path.

Python host mirror of the reference's discretizer API over the C ABI of libpinn_hip.so
(include/pinn_hip.h).  The directory name contains a dot, so import it through the repo-root shim:

    import pinn_import; npde = pinn_import.load()

See DESIGN.md for the kernel design and INTEGRATION.md for the Julia `ccall` binding.
"""
from . import _lib
from ._lib import Engine, EngineError, Library, adam_steps_sharded, comm_init_all, comm_unique_id, loss_grad_sharded, loss_grad_sharded_f64
from .ir import Instr, NetIR, ProblemIR, Slot, TermIR
from .bpinn import BPINNsolution, loglikelihood, physics_loglikelihood
from . import bpinn as _bpinn


def ahmc_bayesian_pinn_pde(pde_system, discretization, **kw):
    """ext/bpinn/PDE_BPINN.jl:371 — see bpinn.ahmc_bayesian_pinn_pde."""
    import sys as _sys
    return _bpinn.ahmc_bayesian_pinn_pde(_sys.modules[__name__], pde_system, discretization, **kw)
from .adaptive import (AbstractAdaptiveLoss, GradientScaleAdaptiveLoss, MiniMaxAdaptiveLoss, ReLoBRaLoAdaptiveLoss,
                       SoftAdaptAdaptiveLoss)
from .pinn import (PeriodicEmbedding, DGM, DeepGalerkin, DataLoss, depvar_params, Adam, BFGS, LBFGS, OptimizationSolution, solve, Chain, Dense, LogOptions, NonAdaptiveLoss, OptimizationFunction, OptimizationProblem, Phi,
                   PhysicsInformedNN, PINNLossFunctions, PINNRepresentation, discretize, initialparameters, remake,
                   symbolic_discretize)
from .strategies import (AbstractTrainingStrategy, QuadratureTraining, GridTraining, LatinHypercubeSample, QuasiRandomTraining,
                         SobolSample, StochasticTraining, generate_random_points, generate_training_sets, get_bounds,
                         get_loss_function, merge_strategy_with_loss_function)
from .symbolic import (Differential, Eq, Equation, In, Interval, LoweringError, PDESystem, get_argument, get_variables,
                       get_vars, lower_equation, parameters, variables)

__all__ = [n for n in dir() if not n.startswith("_")]
