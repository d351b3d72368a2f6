/*
 * pinn_hip.h — C ABI of the MI355X-native PINN residual/loss engine (libpinn_hip.so).
 *
 * Drop-in boundary for NeuralPDE.jl's PhysicsInformedNN/discretize hot path.  The Julia glue
 *  `ccall`s these from the per-term loss closures that
 * `merge_strategy_with_loss_function` returns (reference: src/training_strategies.jl:131-160,
 * 247-269, 336-363) and from the `grad` of the OptimizationFunction built in
 * src/discretize.jl:776-780.  Plain pointers and sizes only; all matrices use the reference's
 * own memory layout (Julia column-major):
 *     points of a term      : d x N Float32, point-major (each point's d coordinates contiguous)
 *                             == the `cord` matrix of the generated loss function, src/discretize.jl:126-131
 *     theta / gradient      : flat vector in ComponentArrays order
 *                             [net(depvar 1): W1 (out x in, col-major) | b1 | W2 | b2 ... | net(depvar 2) ... | p]
 *                             == pinnrep.flat_init_params, src/discretize.jl:451-465
 * Every function returns 0 on success, non-zero on error; pinn_last_error() gives the message
 * (thread-local).  The engine computes with exact (Taylor-mode) derivatives, in fp32 on the device by default
 * (it reproduces the reference's Float64 finite-difference semantics to ~1e-7 relative at initialisation,
 * SURVEY.md §8c) or in float64 (pinn_set_option "precision"; every entry point has a double twin `_f64`).
 * There is NO CPU fallback: without a gfx950 device pinn_create fails.
 *
 * One handle = one caller thread at a time (the reference's loss closures are not re-entrant
 * either: `iteration[] += 1`, src/discretize.jl:574-576).
 */
#ifndef PINN_HIP_H
#define PINN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pinn_engine* pinn_handle;

/* ABI / build identification ("hip" for the product library). */
const char* pinn_backend(void);
int pinn_abi_version(void);
const char* pinn_last_error(void);

/*
 * Build an engine from a problem descriptor (text; "pinnir 1": residual tapes, or "pinnir 2": the equations as s-expressions, lowered
 * inside the library — grammar of both in DESIGN.md §2; optional trailing `hint <term> <points>` lines announce the sizes of the point
 * sets the caller is going to install, so that the planner can put small terms onto a launch their network already has).
 * The descriptor carries what symbolic_discretize extracts from the PDESystem:
 * chains (sizes, activation, offset in theta)   <- pinnrep.phi / flat_init_params   (src/discretize.jl:432-465)
 * per term: dimension, jet slots, residual tape <- symbolic_pde/bc_loss_functions   (src/discretize.jl:505-525)
 * Replaces: build_loss_function + RuntimeGeneratedFunction (src/discretize.jl:163-175).
 * Unsupported expressions / network shapes fail HERE (never a silent fallback).
 */
int pinn_create(const char* descriptor, pinn_handle* out);          /* on the caller's current HIP device */
int pinn_create_on(const char* descriptor, int device, pinn_handle* out);   /* on HIP device `device` (single process, several GPUs) */
int pinn_destroy(pinn_handle h);

/* Number of loss terms K (pde terms first, then bcs: the order of src/discretize.jl:569-570) and length of theta. */
int pinn_num_terms(pinn_handle h);
int64_t pinn_num_theta(pinn_handle h);

/*
 * Install the collocation set of one term (host pointer; copied to HBM and kept resident).
 * Replaces the `train_set` captured by get_loss_function (src/training_strategies.jl:215-221) or the
 * per-call sample of StochasticTraining / QuasiRandomTraining (:271-282, :365-389).
 * n_norm: the N of mean(abs2, .) — the GLOBAL point count when the set is one shard of a
 * multi-GPU partition; pass 0 for n_norm == n.
 */
int pinn_set_points(pinn_handle h, int term, const float* pts, int64_t n, int64_t n_norm);
/* Same, `pts` already in device memory (adopted by copy). */
int pinn_set_points_device(pinn_handle h, int term, const float* d_pts, int64_t n, int64_t n_norm);

/*
 * One evaluation of full_loss_function and its reverse-mode gradient (src/discretize.jl:567-598, :778):
 *   term_losses[k] = mean(abs2, residual_k(set_k, theta))            (K doubles, unweighted)
 *   grad           = d/dtheta sum_k term_w[k] * term_losses[k]       (P floats, nullable)
 * theta: P floats (host).  term_w: K floats (adaptive-loss weights, src/adaptive_losses.jl; NULL = ones).
 * Deterministic: identical inputs give bit-identical outputs.
 * grad == NULL selects the LOSS-ONLY evaluation: forward pass + residual + sums of squares, no records, no reverse sweep, no
 * gradient reduction (about a third of the full evaluation) — the cost class of the reference's per-term closures when they are only
 * evaluated, not differentiated (src/training_strategies.jl:215-221: callbacks, MiniMax / SoftAdapt reweighting, rejected line-search
 * trials).  The per-point residuals are those of the full evaluation bit for bit and their squares are summed in double precision, so
 * the term losses agree with the full evaluation's to the order of those sums (1e-13 relative; identical as floats).
 */
int pinn_loss_grad(pinn_handle h, const float* theta, int64_t p, const float* term_w,
                   double* term_losses, float* grad);
/* Float64 convenience for the reference's default eltype (src/discretize.jl:432-449): converts at the boundary. */
int pinn_loss_grad_f64(pinn_handle h, const double* theta, int64_t p, const double* term_w,
                       double* term_losses, double* grad);
/*
 * Per-term gradients for GradientScaleAdaptiveLoss (src/adaptive_losses.jl:112-123):
 * term_grads is K x P (row k = d term_losses[k] / d theta), row-major.
 */
int pinn_term_grads(pinn_handle h, const float* theta, int64_t p, double* term_losses, float* term_grads);
/* The same with theta and the K x P gradients in double (r06): native on a handle in float64 mode, fp32 kernels + widening otherwise. */
int pinn_term_grads_f64(pinn_handle h, const double* theta, int64_t p, double* term_losses, double* term_grads);

/*
 * BPINN physics (+ data) log-likelihood and its gradients in one evaluation:
 *   loglik = sum_k logpdf(MvNormal(residual_k(set_k, theta), stds[k]^2 I), 0)
 *          = sum_k [ -N_k/2 log(2 pi) - N_k log stds[k] - SSE_k / (2 stds[k]^2) ]                     (SSE, not MSE)
 * i.e. the sum the reference's BayesianPINN builds from get_points_loss_functions (src/training_strategies.jl:113-127,
 * src/discretize.jl:678-754, ext/bpinn/PDE_BPINN.jl:16-26) over the pde and bc terms; a data-misfit term (descriptor op DATA) with
 * its own std is the L2LossData term (ext/bpinn/PDE_BPINN.jl:148-183).  grad_theta (P floats, nullable) = d loglik / d theta by the
 * engine's reverse sweep (the reference differentiates with ForwardDiff over all P parameters, ext/bpinn/PDE_BPINN.jl:519);
 * grad_std (K doubles, nullable) = d loglik / d stds[k] = -N_k / s + SSE_k / s^3 for samplers that treat the stds as parameters.
 * N_k is the number of points INSTALLED for term k (pinn_set_points' n), not its n_norm: with sharded point sets (n < n_norm) every
 * returned quantity is this shard's additive part — loglik, grad_theta and grad_std summed over the ranks are the global values; a caller
 * that passes n_norm != n for another reason (a re-normalised mean) gets the constants of n points here.
 */
int pinn_loglik_grad(pinn_handle h, const float* theta, int64_t p, const double* stds, double* loglik, float* grad_theta, double* grad_std);
/* theta and grad_theta in double (r06; the reference's BPINN samples Float64 parameters, ext/bpinn/PDE_BPINN.jl:519): native on a handle in
 * float64 mode, fp32 kernels + widening otherwise. */
int pinn_loglik_grad_f64(pinn_handle h, const double* theta, int64_t p, const double* stds, double* loglik, double* grad_theta, double* grad_std);
/*
 * Device-resident variant for multi-GPU data parallelism (one process per GPU; the caller
 * all-reduces d_out with RCCL): d_theta = P floats in HBM; d_out = P + K floats in HBM:
 *   d_out[0..P)   = this shard's gradient contribution (already scaled by term_w[k]/n_norm_k)
 *   d_out[P..P+K) = this shard's sum of squared residuals per term (divide by n_norm_k after the all-reduce)
 * Asynchronous on `stream` (a hipStream_t; NULL = the default stream, e.g. torch's current stream).
 * d_out may be any device-accessible address, including pinned device-mapped host memory (hipHostMalloc): the last
 * reduction kernel then delivers the result to the host without a copy command (what pinn_loss_grad does internally).
 */
int pinn_loss_grad_device(pinn_handle h, const float* d_theta, const float* term_w, float* d_out, void* stream);
/* The same for a handle in the float64 evaluation mode (pinn_set_option(h, "precision", "f64")) with DOUBLE device buffers: d_theta = P doubles,
 * d_out = P + K doubles — the Float64 caller's residual + gradient without a narrowing at the boundary and without PCIe (the device-array form
 * of src/discretize.jl:541-545 with Float64 parameters).  Fails on a handle that is not in the mode. */
int pinn_loss_grad_device_f64(pinn_handle h, const double* d_theta, const float* term_w, double* d_out, void* stream);
/* Loss-only counterpart: d_sums = K floats in device-accessible memory = this shard's sum of squared residuals per term
 * (divide by n_norm_k; the same numbers pinn_loss_grad_device writes to d_out[P..P+K)).  Asynchronous on `stream`. */
int pinn_loss_device(pinn_handle h, const float* d_theta, float* d_sums, void* stream);

/*
 * Engine-owned data parallelism over the GPUs of one node: every term's point set is split into contiguous column
 * blocks (install each block with pinn_set_points(..., n_norm = GLOBAL N)), theta is replicated, and ONE RCCL all-reduce (sum, fp32) of
 * the packed vector [gradient (P) | per-term sums of squares (K)] — 51 KB for 4x64 — completes the evaluation on the evaluation's own
 * stream.  The reference aggregates the K term losses on the host (src/discretize.jl:568-588); this is that aggregation across devices.
 *   one process per GPU : rank 0: pinn_comm_unique_id(id); distribute the PINN_COMM_ID_BYTES bytes; all ranks: pinn_comm_init_rank;
 *                         per evaluation pinn_loss_grad_sharded_device (as pinn_loss_grad_device, d_out all-reduced in place; d_out
 *                         must be device memory; divide the K sums by n_norm_k afterwards).
 *   one process, G GPUs : pinn_create_on(desc, g) for g = 0..G-1; pinn_comm_init_all(handles, G) (ncclCommInitAll); per evaluation
 *                         pinn_loss_grad_sharded(handles, G, theta, ...) = pinn_loss_grad over all shards (host pointers in and out).
 * RCCL is loaded on first use; single-GPU work never needs it.
 *   bring your own collective : pinn_comm_init_custom(h, nranks, rank, fn, ctx) makes `fn` the transport of this handle's communicator
 *                         (one process per GPU): the engine calls  fn(ctx, buf, count, dtype, stream)  to sum `count` elements of `buf`
 *                         (dtype 0: float, 1: double; device memory of this rank) over the ranks IN PLACE, ordered after the work already
 *                         queued on `stream` and before whatever the engine queues next (an MPI caller synchronises the stream and calls
 *                         MPI_Allreduce on the device pointers; a torch caller wraps them and calls dist.all_reduce).  Non-zero = failure.
 *
 * The RESIDENT training loop over a communicator (r04): pinn_adam_steps on a handle that belongs to a one-process-per-GPU communicator
 * (pinn_comm_init_rank / _custom), or pinn_adam_steps_sharded over the handles of a pinn_comm_init_all communicator, runs per iteration
 *     [redraw this rank's sampled sets] -> evaluate the local shards -> ONE in-stream all-reduce of [gradient | sums] (+ the K sums as
 *     doubles) -> fused Adam update + weight-image scatter on EVERY rank
 * with nothing crossing PCIe and no host synchronisation inside the loop: every rank applies the identical update to its identical copy
 * of theta (pinn_adam_init with the same theta on every rank).  This is `solve(prob, Adam(lr); maxiters)` over full_loss_function
 * (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:83-85, src/discretize.jl:567-598) with the K-term aggregation done across devices.
 * Device samplers draw rank-specific points (the rank is mixed into the seed); an un-randomised Sobol design (seed 0) cannot be sharded.
 */
#define PINN_COMM_ID_BYTES 128
typedef int (*pinn_allreduce_fn)(void* ctx, void* buf, int64_t count, int dtype, void* stream);
int pinn_comm_init_custom(pinn_handle h, int nranks, int rank, pinn_allreduce_fn fn, void* ctx);
int pinn_adam_steps_sharded(pinn_handle* hs, int ndev, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w,
                            double* loss_history);
/* One Adam update of the resident state from a caller-supplied vector [gradient (P) | raw per-term sums (K)] in host memory: the
 * host-optimiser form of the sharded loop (evaluate with pinn_loss_grad_sharded, apply here); *loss (nullable) = sum_k w_k sums_k / n_norm_k. */
int pinn_adam_apply(pinn_handle h, const float* grad_and_sums, int64_t n, float lr, float beta1, float beta2, float eps, const float* term_w,
                    double* loss);
int pinn_comm_unique_id(void* id, int64_t nbytes);
int pinn_comm_init_rank(pinn_handle h, int nranks, int rank, const void* id, int64_t nbytes);
int pinn_comm_init_all(pinn_handle* hs, int ndev);
int pinn_comm_size(pinn_handle h);
int pinn_comm_rank(pinn_handle h);
int pinn_comm_destroy(pinn_handle h);
int pinn_loss_grad_sharded_device(pinn_handle h, const float* d_theta, const float* term_w, float* d_out, void* stream);
int pinn_loss_grad_sharded(pinn_handle* hs, int ndev, const float* theta, int64_t p, const float* term_w,
                           double* term_losses, float* grad);
/* The same over handles / buffers in the FLOAT64 evaluation mode (r06): this rank's shards by the double kernels, ONE all-reduce of [P + K] doubles
 * (ncclDouble; a caller's transport is called with dtype 1).  pinn_adam_steps on a float64-mode handle with a one-process-per-GPU communicator and
 * pinn_adam_steps_sharded over float64-mode handles run the resident double loop with that all-reduce inside every iteration. */
int pinn_loss_grad_sharded_device_f64(pinn_handle h, const double* d_theta, const float* term_w, double* d_out, void* stream);
int pinn_loss_grad_sharded_f64(pinn_handle* hs, int ndev, const double* theta, int64_t p, const double* term_w,
                               double* term_losses, double* grad);

/* residual_k(set_k, theta): the datafree loss function of src/discretize.jl:174 on the installed set; r has n_k floats. */
int pinn_residual(pinn_handle h, int term, const float* theta, int64_t p, float* r);
/* Trial function phi(x, theta) of one net (src/pinn_types.jl:88-90): pts = d x n point-major, out = n floats. */
int pinn_phi(pinn_handle h, int net, const float* theta, int64_t p, const float* pts, int64_t n, float* out);
/*
 * derivative(phi, u, x, eps, order, theta) of one net at n points (numeric_derivative, src/pinn_types.jl:445-482; its epsilons come from
 * get_eps, src/symbolic_utilities.jl:98-103): the derivative of the trial function of order `order` (0..4) along `axes[0..order)`
 * (network-input axes, 0-based; order 2 may be mixed, orders 3-4 are along one axis).  The engine returns the EXACT derivative the
 * Taylor-jet kernels carry (the value the reference's central differences approximate to ~1e-8, test/Forward/forward__derivatives.jl:22-44).
 */
int pinn_derivative(pinn_handle h, int net, const float* theta, int64_t p, const float* pts, int64_t n, int order, const int* axes, float* out);
/*
 * The three per-point closures in DOUBLE (r06).  The reference evaluates them in eltype(theta), Float64 by default (src/discretize.jl:432-449,
 * src/eltype_matching.jl:8-10), and pins them at Float64 tolerances: datafree_pde_loss_functions[i](cord, theta) at rtol 1e-8
 * (src/pinn_types.jl:435-439, test/Forward/forward__ode.jl:46-47), numeric_derivative against automatic differentiation at atol 1e-8 / 4e-5
 * (src/pinn_types.jl:445-482, test/Forward/forward__derivatives.jl:29-44), phi(x, theta) (src/pinn_types.jl:88-90).  On a handle in float64
 * mode (pinn_set_option(h, "precision", "f64")) theta, points and results cross the boundary in double and the double kernels evaluate them
 * (matrix-pipe tile kernels where instantiated, one lane per point elsewhere): results equal a Float64 evaluation to rounding.  On an fp32
 * handle they narrow / widen at the boundary (the fp32 kernels run).  pinn_derivative_f64 covers the derivatives the float64 jet sets carry
 * (orders <= 2 in 1-3 inputs, pure orders 3-4, first + pure second in 4 inputs); anything else fails with a message.
 * pinn_residual_f64 reads the term's set as installed: pinn_set_points_f64 for Float64 coordinates.
 */
int pinn_residual_f64(pinn_handle h, int term, const double* theta, int64_t p, double* r);
int pinn_phi_f64(pinn_handle h, int net, const double* theta, int64_t p, const double* pts, int64_t n, double* out);
int pinn_derivative_f64(pinn_handle h, int net, const double* theta, int64_t p, const double* pts, int64_t n, int order, const int* axes, double* out);

/*
 * Resident-theta training loop: theta, the Adam moments and the collocation sets stay in HBM; no
 * per-iteration PCIe traffic.  Mirrors `solve(prob, Adam(lr); maxiters)` ([3P] OptimizationOptimisers, used by every
 * reference test, e.g. test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:83) on the objective of pinn_loss_grad.
 *   pinn_set_sampler : kind 1 = redraw the term's points uniformly in [lb, ub] on the device before every step
 *                      (StochasticTraining, src/training_strategies.jl:242-245, 277-281); kind 2 = Latin-hypercube redraw
 *                      (QuasiRandomTraining with its default LatinHypercubeSample and resampling = true, :321, 375-381);
 *                      kind 3 = Sobol' design (QuasiRandomTraining with SobolSample: Gray-code sequence, Joe-Kuo direction numbers,
 *                      first element skipped as in Sobol.jl; seed 0 = the plain sequence on every draw, otherwise a fresh digital shift per draw);
 *                      kind 0 = keep the installed set.
 *   pinn_adam_init   : upload theta (P floats), zero the moments.
 *   pinn_adam_steps  : nsteps updates m = b1 m + (1-b1) g, v = b2 v + (1-b2) g^2, theta -= lr m^/(sqrt(v^) + eps);
 *                      loss_history (nullable) receives the weighted total loss of every step.
 *   pinn_adam_get    : download theta.
 */
int pinn_set_sampler(pinn_handle h, int term, int kind, const float* lb, const float* ub, int64_t n, uint64_t seed);
/*
 * Per-point data of a term whose residual contains DATA channels (descriptor op `DATA j`): ndata x N floats, channel-major, for the
 * point set installed last (observations d_i of a data-misfit term  u(x_i) - d_i, the usual content of the reference's
 * `additional_loss`, e.g. docs/src/tutorials/param_estim.md:79-95, evaluated inside the fused loss + gradient instead of on the host).
 */
int pinn_set_point_data(pinn_handle h, int term, const float* data, int ndata, int64_t n);
/* the same with the observations in double: the float64 evaluation mode reads them as given, the fp32 kernels their float conversion */
int pinn_set_point_data_f64(pinn_handle h, int term, const double* data, int ndata, int64_t n);
/*
 * Quadrature weights for the term's current point set: the term's loss becomes  sum_i w[i] * r_i^2  (with sum_i w[i] = 1: a weighted mean,
 * e.g. a tensor Gauss-Legendre rule — (1/area) * integral of r^2, the objective of the reference's QuadratureTraining,
 * src/training_strategies.jl:451-481) instead of mean(abs2, r).  NULL restores the plain mean.  Sharded sets: pass the shard's
 * weights of the GLOBAL rule (n_norm of pinn_set_points is then only a scale that cancels).
 */
int pinn_set_point_weights(pinn_handle h, int term, const float* w, int64_t n);
/* Copy the term's current collocation set (d x N, point-major, as installed or as last drawn by the device sampler) to the host. */
int pinn_get_points(pinn_handle h, int term, float* pts, int64_t n);
int pinn_adam_init(pinn_handle h, const float* theta, int64_t p);
int pinn_adam_steps(pinn_handle h, int nsteps, float lr, float beta1, float beta2, float eps, const float* term_w, double* loss_history);
int pinn_adam_get(pinn_handle h, float* theta, int64_t p);
/* The same in double (r05).  On a handle in float64 mode (pinn_set_option(h, "precision", "f64")) the WHOLE loop runs in double on the
 * device — parameters, moments, the redrawn point sets (the fp32 samplers' points widened on the device: StochasticTraining /
 * QuasiRandomTraining(resampling = true) with the reference's default Float64 parameters, src/discretize.jl:432-449,
 * src/training_strategies.jl:271-282, 365-389), residual + gradient kernels, Adam — and pinn_adam_init / pinn_adam_get convert at the
 * boundary; in float32 mode these two narrow / widen at the boundary. */
int pinn_adam_init_f64(pinn_handle h, const double* theta, int64_t p);
int pinn_adam_get_f64(pinn_handle h, double* theta, int64_t p);
/*
 * L-BFGS on the weighted objective sum_k w_k L_k over FIXED point sets (the quasi-Newton finisher of the reference's scripts,
 * `solve(prob, BFGS() / LBFGS(); maxiters)`, e.g. test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:86 — there [3P] OptimizationOptimJL on the
 * host; here inside the library so that C / Julia / Python callers share it).  theta (double, in/out) is iterated in double precision
 * on the host; every objective / gradient evaluation is one fused device evaluation in fp32.  Two-loop recursion with `history` pairs,
 * backtracking line search (Armijo), stops after `maxiters` iterations, when the gradient's max-norm falls below `gtol`, or when the line
 * search cannot decrease the objective any more (the noise floor of the evaluation; a handle in "split" GEMM mode first switches itself to
 * "fp32" there and goes on — the split products' floor is 2-4 x higher — and is switched back before the call returns;
 * $PINN_LBFGS_KEEP_GEMM=1 disables that).  loss_history (nullable): objective after every iteration,
 * `maxiters` entries; *iters_done: iterations performed.  Terms with device samplers are refused (the objective must not change).
 */
int pinn_lbfgs(pinn_handle h, double* theta, int64_t p, int maxiters, int history, double gtol, const float* term_w, double* loss_history,
               int* iters_done);

/*
 * Run-time options of a handle.  "gemm" = arithmetic of the hidden-layer GEMMs of the 64- / 128-wide (neuron-split) kernels:
 *   "split" (default) — every fp32 product rebuilt from three bf16 pieces per operand on the bf16 matrix pipe (6 MFMAs, fp32 accumulation):
 *                       2-4 x the rounding error of an fp32 fmaf chain, ~1.35 x faster (error budget: DESIGN.md section 6);
 *   "fp32"            — v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain per product, for callers that need the last bit (a quasi-Newton
 *                       finisher at its noise floor: the reference's `solve(prob, BFGS())` stage, test/NNPDE1/nnpde__pde_iii_3rd_order_ode.jl:89-93,
 *                       runs on Float64 CPU arithmetic there).
 *   "auto" (r06)      — the engine chooses between the two by MEASUREMENT: at the current parameters the gradient is evaluated with both arithmetics,
 *                       delta = |grad(split) - grad(fp32)|_2 / |grad(fp32)|_2  (the split products' arithmetic error at this iterate); "split" runs
 *                       while delta <= 1e-5 (the north star's tolerance), "fp32" above, back under 3e-6.  ~2e-7 at initialisation, 1e-5 / 4e-3 at the
 *                       trained fixtures (cfg2 after 2,000 / 6,000 Adam steps).  Checked at the end of a pinn_adam_steps call (loop path) and inside
 *                       pinn_loss_grad with a gradient, at most once per 1,000 optimiser steps / evaluations (two evaluations + at most two re-plans
 *                       per check), never inside a resident loop.  pinn_get_option(h, "gemm") reports "auto(split)" / "auto(fp32)",
 *                       "gemm_delta" the last delta, "grad_health" rho = |grad|_2 / sqrt(sum_k w_k L_k) of the last measured evaluation: rho against
 *                       its value at initialisation is the cancellation factor of the gradient sum — once it has fallen by ~1e3 no fp32 evaluation holds
 *                       1e-3 any more and "precision" = "f64" is the remedy (the glue's precision policy puts Float64 parameters there from the start).
 * Both kernel sets are in the library; switching rebuilds the handle's kernel plan in place (milliseconds) and keeps point sets, samplers,
 * per-point data / weights and the optimiser state.  Narrower nets (one-wave-per-tile kernels) and DGM nets compute on fp32 MFMAs / the
 * VALU in either mode.  $PINN_GEMM = split | fp32 sets the mode new handles start in.  pinn_get_option writes the current value.
 * "precision" = "f32" (default) | "f64": the FLOAT64 evaluation mode — the reference's default eltype (src/discretize.jl:432-449).  With
 *   "f64", pinn_loss_grad_f64 evaluates natively in double (pinn_loss_grad converts at the boundary) and pinn_lbfgs iterates on the double
 *   objective: what a quasi-Newton stage needs to take an objective below ~1e-7 (test/NNPDE1/nnpde__pde_iii_3rd_order_ode.jl:89-93) and
 *   what parity at TRAINED parameters needs (fp32 cannot hold 1e-5 there).  Kernels: wave-private tiles on
 *   v_mfma_f64_16x16x4_f64 (csrc/pinn_kernels5.hpp) for tanh / sigmoid nets with hidden layers up to 64 wide and the instantiated jet sets,
 *   one lane per point on the fp64 VALU (csrc/pinn_kernels4.hpp) for everything else the mode covers; pinn_get_option(h, "f64_path") reports
 *   what the last evaluation ran ("mfma" | "lanes" | "mfma+lanes").  7-8x the time of the fp32 kernels on the matrix pipe.
 *   Covers equations of up to 6 dependent variables (same argument count), Dense chains with tanh / sigmoid / sin, derivative orders <= 2
 *   in 1-3 inputs (1-D, and mixed / pure in 2-D and 3-D where instantiated: <= 4; 4-D: first and pure second), PDE parameters, quadrature
 *   weights, per-point DATA channels, device samplers, periodic input embeddings (r06); anything else (DGM) fails HERE with a message and leaves the fp32
 *   plan usable.  In this mode EVERY evaluating entry point runs the double kernels (r06): pinn_loss_grad / pinn_loss_grad_device /
 *   pinn_loss_device / pinn_term_grads / pinn_loglik_grad / pinn_residual / pinn_phi / pinn_derivative convert at the boundary, their _f64
 *   twins hand everything over in double; pinn_adam_init / _steps / _get / _apply keep theta and the moments in double on the device, in
 *   a buffer of their own (evaluations between two pinn_adam_steps calls — adaptive reweighting, callbacks — leave the iterate alone);
 *   samplers redraw in float and the double copy follows.  pinn_set_points_f64 / pinn_set_point_data_f64 install a point set / its
 *   observations in double (the fp32 kernels get the float conversion).  Communicators (r06): pinn_loss_grad_sharded_device_f64 /
 *   pinn_loss_grad_sharded_f64 and the resident loops (pinn_adam_steps over a one-process-per-GPU communicator, pinn_adam_steps_sharded) all-reduce
 *   [P + K] DOUBLES; the float entry points pinn_loss_grad_sharded(_device) run the fp32 kernels whatever the mode.
 *   Small problems (r06: the reference's own regime, a few hundred points per term): row-tile counts of 1 / 2 for 16- / 32-wide nets, and the terms
 *   that share an input binding and an instantiated jet set are evaluated by ONE launch sequence (<= 6 members, <= 8,192 points together) —
 *   same results to rounding; pinn_get_option(h, "f64_merged") = the number of such sequences in the last evaluation, $PINN_F64_NO_MERGE=1
 *   switches them off (A/B, tests).  Terms whose residual is affine in the trial function(s) with constant coefficients (boundary conditions, linear
 *   PDEs with forcing terms) skip the tape interpreter in the tile kernel: the coordinate-only part is evaluated once per point set
 *   (pinn_get_option(h, "f64_affine") = the number of such terms; $PINN_F64_NO_LIN=1: off).  The weight-gradient kernel of a small launch works on
 *   short point blocks (64 ... 256 instead of 512 points per workgroup row, chosen so that the launch has ~512 workgroups; $PINN_F64_NO_SHORT_BLOCKS=1: off).
 *   PRECISION POLICY of the glue (Julia: HIPStrategy / hip_discretize `precision = :auto`; Python mirror: PhysicsInformedNN(precision = "auto")):
 *   the reference's contract compute dtype = eltype(theta) (src/eltype_matching.jl:8-10) — Float64 parameters select "f64", Float32
 *   parameters "f32"; "f32" on Float64 parameters is the explicit fast opt-in.
 * "derivative" = "exact" (default) | "stencil" (r06; float64 mode only): a VALIDATION mode, not a performance path.  "stencil" evaluates every
 *   derivative slot as the reference's central differences — numeric_derivative (src/pinn_types.jl:445-482: the order-1..4 formulas, the
 *   recursion for mixed derivatives and orders > 4) with the steps of get_eps (src/symbolic_utilities.jl:98-103: eps(Float64)^(1/(2+order)),
 *   the TOTAL order's step on every axis, :185), in the reference's order of operations — as value-only forward passes at shifted copies of the
 *   point set, the difference formulas as tape ops, and a reverse sweep through the same combination (one seeded launch per shifted set;
 *   csrc/f64.cpp: f64_stencil_term).  pinn_loss_grad*, pinn_term_grads*, pinn_loglik_grad*, pinn_residual* then return the reference's
 *   finite-difference numbers instead of exact derivatives: for digit-by-digit comparisons with the reference's generated loss functions
 *   (julia: NeuralPDEHIP.selftest) and with the stencil oracle.  How close two correct implementations of these formulas can be is bounded by the
 *   formulas themselves: u(x +- eps) carries ~1e-16 relative rounding and 1 / eps^2 ~ 7e7 multiplies it — 1e-8 at initialisation, 1e-5 ... 1e-4
 *   of the gradient at trained parameters (tests/test_f64_mode.py measures it as the stencil oracle against itself with permuted neurons).
 * "persistent" = "on" (default) | "off": pinn_adam_steps runs a SMALL problem — one network of the one-wave-per-tile kernel family, at most
 *   32 workgroups (~2,000 points of a 3 x 32 net), fixed or device-redrawn point sets (pinn_set_sampler), no estimated PDE parameters, no communicator — as ONE persistent launch
 *   per call (csrc/pinn_train.hpp: evaluation, fixed-order reduction, Adam and the weight-image update of every iteration inside the kernel,
 *   two grid barriers per iteration) instead of three launches per iteration: the reference's own test regime,
 *   solve(prob, Adam; maxiters = 4000) on 100-1,000 points (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:83-85).  Bit-identical to the loop.
 *   A launch whose workgroups are not all resident (a device shared with another process) ends by its barrier's time-out: the call then
 *   restores the optimiser state, runs the loop instead and keeps the loop for the handle.  $PINN_PERSISTENT=0 switches it off for every handle.  pinn_get_option(h, "adam_path") reports what the last pinn_adam_steps call ran:
 *   "persistent" | "loop" | "none".  The host-entry evaluations of such a problem (pinn_loss_grad with a gradient, the objective calls of
 *   pinn_lbfgs) take the same kernel in its evaluation-only form — residual kernel, grid barrier and fixed-order sums in ONE launch instead of
 *   two, same numbers — unless HIP events were requested (pinn_set_timing); pinn_get_option(h, "eval_path"): "one launch" | "stand-alone kernels".
 */
int pinn_set_points_f64(pinn_handle h, int term, const double* pts, int64_t n, int64_t n_norm);
int pinn_set_option(pinn_handle h, const char* name, const char* value);
int pinn_get_option(pinn_handle h, const char* name, char* buf, int64_t buflen);

/* Timing of the last pinn_loss_grad*: HIP-event milliseconds of the fused residual kernels / of the whole device section. */
int pinn_last_timing(pinn_handle h, float* kernel_ms, float* total_ms);
/* How many HIP events an evaluation records: level 0 (default) none, 1 a start/stop pair around launch group `group` (-1: every
 * group), 2 additionally the phase events behind pinn_last_timing.  Every recorded event costs a few us of dispatch gap between
 * kernels — 25 us per pinn_loss_grad call at level 2, measured on the reference's 1-D Poisson test (profiles/r04_train_kernel.txt) —
 * so events are opt-in (the default changed from level 2 to level 0 in r04: callers of pinn_last_timing must call pinn_set_timing(h, 2, -1)
 * first): with the events off pinn_last_timing FAILS (non-zero return, pinn_last_error says how to switch them on) and
 * pinn_group_timing reports ms = -1. */
int pinn_set_timing(pinn_handle h, int level, int group);
/* Kernel plan: number of launch groups (terms that share one fused kernel) and the HIP-event duration of group g's
 * fused residual kernel in the last evaluation (-1 if that group was not timed, see pinn_set_timing), with the
 * points / jet channels / wave tiles it processed. */
int pinn_num_groups(pinn_handle h);
int pinn_group_timing(pinn_handle h, int group, float* ms, int64_t* points, int* channels, int* tiles);
/* Launch group whose launch carried group g's tiles in the last evaluation: g itself, or the head group of a MERGED launch (two
 * kernel-family members of one network — e.g. the interior jet set and the value-only boundary set — walked by one persistent
 * kernel; the head's pinn_group_timing then covers both).  -1: not launched yet / bad index. */
int pinn_group_launched_by(pinn_handle h, int group);
/* Introspection used by tests and bench: writes a short human-readable description of the kernel plan. */
int pinn_describe(pinn_handle h, char* buf, int64_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* PINN_HIP_H */
