# NeuralPDEHIP.jl — the Julia side of the drop-in boundary: NeuralPDE.jl's `PhysicsInformedNN` / `discretize` hot path on the MI355X
# engine (libpinn_hip.so, C ABI in include/pinn_hip.h).
#
# Everything symbolic stays where it is (ModelingToolkit / Symbolics / `symbolic_discretize`); this module
#   1. PRINTS the expression trees the reference itself walks — `toexpr(expand_derivatives(eq.lhs))`, `toexpr(...rhs)`
#      (src/symbolic_utilities.jl:360-370) — as prefix s-expressions (`sexpr`).  The lowering (`_transform_expression`,
#      src/symbolic_utilities.jl:132-331: dependent-variable calls -> `u(cord, θ, phi)`, nested Differentials -> one
#      `derivative(...)` call, everything else broadcast arithmetic) is done INSIDE the library (csrc/sexpr.cpp), the same code the
#      Python mirror drives, so this file contains no lowering logic that could drift;
#   2. writes the problem descriptor "pinnir 2" (`descriptor(pinnrep)`): chains, the flat-θ (ComponentArrays) layout, per equation the
#      coordinate row order `this_eq_indvars` exactly as `build_symbolic_loss_function` computes it (src/discretize.jl:41-43);
#   3. plugs into the reference at its two existing plug-in points (SURVEY.md §8b):
#        * `HIPStrategy(inner)  <: NeuralPDE.AbstractTrainingStrategy` + `merge_strategy_with_loss_function` (called at
#          src/discretize.jl:541-545): per-term closures `θ -> mean(abs2, residual)` with `ChainRulesCore.rrule`s, so `discretize`,
#          adaptive losses, logging, `additional_loss`, `AutoZygote` and user callbacks keep working unchanged;
#        * `hip_discretize(pde_system, discretization)`: the fast path — `OptimizationFunction(f; grad = ...)` whose gradient is ONE
#          fused `pinn_loss_grad` call (all terms, current adaptive weights), bypassing Zygote for the physics terms;
#   4. checks the θ layout once per engine (`verify_layout`: `pinn_phi` against the Lux chain at random points) and rethrows every
#      non-zero status of the C ABI as a Julia exception (`HIPEngineError`).
#
# Status: written against NeuralPDE v6.2.2 / Lux 1.x / ComponentArrays 0.15 / Symbolics 7 from their sources and docs; Julia is not
# installed in the build container, so this file has NOT been executed there.  What can be pinned without Julia is pinned:
# tests/golden/descriptors/*.pinnir2 hold the descriptors of the five BASELINE configurations in exactly the format `descriptor`
# emits, tests/test_sexpr_frontend.py feeds them (and Julia-style spellings of the same equations) through the library against the
# float64 oracle, and `NeuralPDEHIP.selftest()` below re-checks every descriptor against the reference's own generated loss functions
# on the first machine that has both Julia and the library.

module NeuralPDEHIP

using Libdl
using Random
using Statistics
using ComponentArrays
using ChainRulesCore
using SciMLBase
using Symbolics
using Symbolics: toexpr, expand_derivatives, Differential
using SymbolicUtils
import Lux
import NeuralPDE
import NeuralPDE: AbstractTrainingStrategy, PINNRepresentation, GridTraining, StochasticTraining, QuasiRandomTraining,
                  PhysicsInformedNN, merge_strategy_with_loss_function
import QuasiMonteCarlo
import Optimization

export HIPStrategy, hip_discretize, descriptor, sexpr, HIPEngine, HIPEngineError, state_of

# ------------------------------------------------------------------------------------------------
# library + error convention (include/pinn_hip.h: every function returns 0 or sets pinn_last_error)
# ------------------------------------------------------------------------------------------------
const LIBPATH = Ref{String}(get(ENV, "PINN_HIP_LIB", joinpath(@__DIR__, "..", "neuralpde.jl_amd", "csrc", "libpinn_hip.so")))
const LIB = Ref{Ptr{Cvoid}}(C_NULL)

struct HIPEngineError <: Exception
    msg::String
end
Base.showerror(io::IO, e::HIPEngineError) = print(io, "HIPEngineError: ", e.msg)

function lib()
    if LIB[] == C_NULL
        isfile(LIBPATH[]) || throw(HIPEngineError("$(LIBPATH[]) not found: build it (python -c 'import __graft_entry__ as g; g.build()') " *
                                                  "or set ENV[\"PINN_HIP_LIB\"]; the engine has no CPU fallback"))
        LIB[] = Libdl.dlopen(LIBPATH[])
    end
    return LIB[]
end
sym(name::Symbol) = Libdl.dlsym(lib(), name)
last_error() = unsafe_string(ccall(sym(:pinn_last_error), Cstring, ()))
check(rc::Integer, what::AbstractString) = rc == 0 ? nothing : throw(HIPEngineError("$what: $(last_error())"))

# ------------------------------------------------------------------------------------------------
# one pinn_handle
# ------------------------------------------------------------------------------------------------
mutable struct HIPEngine
    h::Ptr{Cvoid}
    K::Int          # number of loss terms (pde terms first, then bcs: src/discretize.jl:569-570)
    P::Int          # length(θ)
    precision::Symbol   # :f32 (fp32 kernels) | :f64 (float64 evaluation mode); follows `set_precision!`
    function HIPEngine(desc::AbstractString; device::Integer = -1)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall(sym(:pinn_create_on), Cint, (Cstring, Cint, Ref{Ptr{Cvoid}}), desc, device, out), "pinn_create")
        e = new(out[], 0, 0, :f32)
        e.K = ccall(sym(:pinn_num_terms), Cint, (Ptr{Cvoid},), e.h)
        e.P = ccall(sym(:pinn_num_theta), Int64, (Ptr{Cvoid},), e.h)
        # (no HIP events around the kernels of an evaluation — the library's default; profiling callers switch them on with pinn_set_timing)
        finalizer(x -> (x.h != C_NULL && ccall(sym(:pinn_destroy), Cint, (Ptr{Cvoid},), x.h); x.h = C_NULL), e)
        return e
    end
end

"Install the collocation set of term `k` (1-based): `pts` is the reference's `d × N` matrix (column-major == point-major, no transpose)."
function set_points!(e::HIPEngine, k::Integer, pts::AbstractMatrix; n_norm::Integer = 0)
    # EltypeAdaptor of the reference (src/eltype_matching.jl:8-10): the points take the compute dtype — Float64 in float64 mode
    # (pinn_set_points_f64: the double kernels read them as given), Float32 for the fp32 kernels
    if e.precision === :f64
        p64 = Matrix{Float64}(pts)
        GC.@preserve p64 check(ccall(sym(:pinn_set_points_f64), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Int64),
                                     e.h, k - 1, p64, size(p64, 2), n_norm), "pinn_set_points_f64")
    else
        p32 = Matrix{Float32}(pts)
        GC.@preserve p32 check(ccall(sym(:pinn_set_points), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Int64, Int64),
                                     e.h, k - 1, p32, size(p32, 2), n_norm), "pinn_set_points")
    end
    return nothing
end

"""
`(term_losses::Vector{Float64}, grad::Vector{Float64})` of `Σ_k w[k] * mean(abs2, residual_k)` — one fused device evaluation.
`want_grad = false` is the engine's LOSS-ONLY evaluation (`grad = NULL` at the ABI: forward pass + residuals + sums of squares, no reverse
sweep, about 2.6x cheaper): the same term losses, `grad` comes back empty.
"""
function loss_grad(e::HIPEngine, θ::AbstractVector{<:Real}, w::AbstractVector{<:Real}; want_grad::Bool = true)
    θ64 = Vector{Float64}(θ); w64 = Vector{Float64}(w)
    length(θ64) == e.P || throw(DimensionMismatch("θ has $(length(θ64)) entries, the engine expects $(e.P)"))
    length(w64) == e.K || throw(DimensionMismatch("need one weight per loss term ($(e.K))"))
    losses = zeros(Float64, e.K)
    grad = zeros(Float64, want_grad ? e.P : 0)
    GC.@preserve θ64 w64 losses grad check(ccall(sym(:pinn_loss_grad_f64), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        e.h, θ64, e.P, w64, losses, want_grad ? pointer(grad) : Ptr{Float64}(C_NULL)), "pinn_loss_grad_f64")
    return losses, grad
end

"""
    set_gemm!(e, :split | :fp32 | :auto)

Arithmetic of the hidden-layer GEMMs of the 64- / 128-wide kernels on a live engine (`pinn_set_option(h, "gemm", …)`): `:split` (default) =
three bf16 pieces per operand on the bf16 matrix pipe; `:fp32` = fp32 MFMAs, an fmaf chain per product — for a quasi-Newton stage that
runs into the split products' noise floor (the reference runs `BFGS()` in Float64, test/NNPDE1/nnpde__pde_iii_3rd_order_ode.jl:89-93).
"""
function set_gemm!(e::HIPEngine, mode::Symbol)
    # :auto (r06): the engine MEASURES when to leave the split products — at the current iterate the gradient is evaluated with both arithmetics,
    # δ = ‖g_split − g_fp32‖ / ‖g_fp32‖; split while δ ≤ 1e-5, fp32 above (include/pinn_hip.h, "gemm")
    mode in (:split, :fp32, :auto) || throw(ArgumentError("gemm mode must be :split, :fp32 or :auto"))
    check(ccall(sym(:pinn_set_option), Cint, (Ptr{Cvoid}, Cstring, Cstring), e.h, "gemm", String(mode)), "pinn_set_option")
    return nothing
end

"""
    adam!(e::HIPEngine, θ0, nsteps, η; w = ones(K), init = true) -> (θ, loss_history)

`solve(prob, Adam(η); maxiters = nsteps)` with θ, the moments and the point sets resident on the device (`pinn_adam_steps`).
"""
function adam!(e::HIPEngine, θ0::AbstractVector{<:Real}, nsteps::Integer, η::Real; w::AbstractVector{<:Real} = ones(e.K),
               β1::Real = 0.9, β2::Real = 0.999, ϵ::Real = 1.0e-8, init::Bool = true)
    # `solve(prob, Adam(η); maxiters = nsteps)` with θ, the moments and the point sets resident on the device (`pinn_adam_steps`): the
    # persistent kernel where the problem is small enough, the launch-per-step loop otherwise; `init = false` continues from the device state
    # The parameters cross the boundary in DOUBLE (pinn_adam_init_f64 / pinn_adam_get_f64, r05): a handle in float64 mode
    # (`set_precision!(e, :f64)`, the reference's default eltype, src/discretize.jl:432-449) keeps θ, the moments, the redrawn point sets and
    # every kernel of the iteration in double on the device; a float32 handle narrows / widens at the boundary.
    θ64 = Vector{Float64}(θ0); w32 = Vector{Float32}(w)
    if init
        GC.@preserve θ64 check(ccall(sym(:pinn_adam_init_f64), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), e.h, θ64, e.P), "pinn_adam_init_f64")
    end
    hist = zeros(Float64, nsteps)
    GC.@preserve w32 hist check(ccall(sym(:pinn_adam_steps), Cint,
        (Ptr{Cvoid}, Cint, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float64}), e.h, nsteps, η, β1, β2, ϵ, w32, hist), "pinn_adam_steps")
    out = zeros(Float64, e.P)
    GC.@preserve out check(ccall(sym(:pinn_adam_get_f64), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), e.h, out, e.P), "pinn_adam_get_f64")
    return out, hist
end

"""
    set_persistent!(e, on::Bool)

`adam!` on a SMALL problem (one network of at most 32-wide layers, up to ~2,000 collocation points, fixed or device-redrawn point sets)
runs all its iterations inside ONE persistent launch (`pinn_set_option(h, "persistent", …)`, DESIGN.md section 4.6: 2x fewer microseconds
per iteration in the regime of the reference's own tests, `solve(prob, Adam(0.1); maxiters = 4000)`,
test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:83-85); results are bit-identical to the launch-per-step loop.  On by default; `false` keeps
the loop.  `adam_path(e)` tells which one the last `adam!` ran (`:persistent | :loop | :none`).
"""
function set_persistent!(e::HIPEngine, on::Bool)
    check(ccall(sym(:pinn_set_option), Cint, (Ptr{Cvoid}, Cstring, Cstring), e.h, "persistent", on ? "on" : "off"), "pinn_set_option")
    return nothing
end
function adam_path(e::HIPEngine)
    buf = zeros(UInt8, 64)
    check(ccall(sym(:pinn_get_option), Cint, (Ptr{Cvoid}, Cstring, Ptr{UInt8}, Int64), e.h, "adam_path", buf, 64), "pinn_get_option")
    return Symbol(unsafe_string(pointer(buf)))
end

"""
    set_precision!(e, :f64 | :f32)

FLOAT64 evaluation of a live engine (`pinn_set_option(h, "precision", …)`, DESIGN.md section 4.5): `loss_grad` and `lbfgs!` then run the
double kernels — the reference's default eltype (src/discretize.jl:432-449) — for a `BFGS()` finisher below the fp32 noise floor or a
digit-by-digit comparison with a Float64 CPU run.  Point sets already installed are converted; `set_points!` keeps feeding both.
Throws (and leaves the fp32 plan untouched) for problems the mode does not cover (DGM nets; periodic input embeddings are covered since round 6).  Since round 5 the resident
Adam loop, the device samplers, per-point DATA channels and the device-pointer entries evaluate in double as well.
"""
function set_precision!(e::HIPEngine, mode::Symbol)
    mode in (:f64, :f32) || throw(ArgumentError("precision must be :f64 or :f32"))
    check(ccall(sym(:pinn_set_option), Cint, (Ptr{Cvoid}, Cstring, Cstring), e.h, "precision", String(mode)), "pinn_set_option")
    e.precision = mode
    return nothing
end

"""
    set_derivative!(e, :exact | :stencil)

`:stencil` (float64 mode only; a VALIDATION mode, `pinn_set_option(h, "derivative", "stencil")`): derivative slots evaluated as the reference's
central differences (`numeric_derivative`, src/pinn_types.jl:445-482, with `get_ε` steps, src/symbolic_utilities.jl:98-103) instead of exact
Taylor jets, so that `residual`, `loss_grad` and `term_grads` return the reference's own finite-difference numbers — what `selftest` compares
digit by digit with the generated loss functions.  `:exact` (default) restores the Taylor-jet kernels.
"""
function set_derivative!(e::HIPEngine, mode::Symbol)
    mode in (:exact, :stencil) || throw(ArgumentError("derivative must be :exact or :stencil"))
    check(ccall(sym(:pinn_set_option), Cint, (Ptr{Cvoid}, Cstring, Cstring), e.h, "derivative", String(mode)), "pinn_set_option")
    return nothing
end

"""
    resolve_precision(precision::Symbol, θ) -> :f32 | :f64

The glue's PRECISION POLICY (r06) = the reference's contract, compute dtype = eltype(θ) (src/eltype_matching.jl:8-10; `init_params` are
Float64 unless the user passes Float32 ones, src/discretize.jl:432-449):
`:auto` (default of `HIPStrategy` / `hip_discretize`) selects the float64 kernels for `eltype(θ) == Float64` and the fp32 kernels for
`Float32`; `:f32` is the explicit fast opt-in (fp32 kernels whatever eltype(θ), 7-8x faster on the matrix pipe, results converted at
the boundary); `:f64` forces the float64 kernels.  A problem the float64 kernels do not cover (DGM networks) fails
at `discretize` time under `:auto` with the library's message — never a silent narrowing; pass `precision = :f32` for it.
"""
function resolve_precision(precision::Symbol, θ)
    precision in (:auto, :f32, :f64) || throw(ArgumentError("precision must be :auto, :f32 or :f64"))
    precision === :auto || return precision
    return eltype(ComponentArrays.getdata(θ)) === Float32 ? :f32 : :f64
end

"""
    set_point_data_f64!(e, k, data)

Observations of a data-misfit term (descriptor op `DATA j`; `ndata × N`, channel-major) in double (`pinn_set_point_data_f64`): the float64
evaluation mode reads them as given, the fp32 kernels their float conversion.
"""
function set_point_data_f64!(e::HIPEngine, k::Integer, data::AbstractMatrix)
    d64 = Matrix{Float64}(permutedims(data))          # Julia column-major N × ndata == C row-major ndata × N
    GC.@preserve d64 check(ccall(sym(:pinn_set_point_data_f64), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint, Int64),
                                 e.h, k - 1, d64, size(data, 1), size(data, 2)), "pinn_set_point_data_f64")
    return nothing
end

"""
    loss_grad_device_f64!(e, dθ::Ptr{Float64}, dout::Ptr{Float64}, w; stream = C_NULL)

The float64 evaluation on DOUBLE device buffers (`pinn_loss_grad_device_f64`; e.g. the pointers of two `ROCArray{Float64}`): `dθ` holds P
parameters, `dout` receives `[gradient (P) | raw per-term sums of squares (K)]`; asynchronous on `stream`.  The engine must be in float64 mode.
"""
function loss_grad_device_f64!(e::HIPEngine, dθ::Ptr{Float64}, dout::Ptr{Float64}, w::AbstractVector{<:Real}; stream::Ptr{Cvoid} = C_NULL)
    w32 = Vector{Float32}(w)
    GC.@preserve w32 check(ccall(sym(:pinn_loss_grad_device_f64), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float32}, Ptr{Float64}, Ptr{Cvoid}),
                                 e.h, dθ, w32, dout, stream), "pinn_loss_grad_device_f64")
    return nothing
end

# The per-point / per-term closures cross the boundary in DOUBLE (r06: pinn_*_f64).  On an engine in float64 mode the double kernels evaluate
# them — the reference computes all of them in eltype(θ) (src/pinn_types.jl:88-90, 435-439, 445-482) — and nothing is narrowed; on an fp32 engine
# the library narrows / widens at the boundary, so the same glue serves both.
"Per-term gradients `P × K` (column k = ∂ term_losses[k] / ∂θ): what GradientScaleAdaptiveLoss and the per-term rrules consume."
function term_grads(e::HIPEngine, θ::AbstractVector{<:Real})
    θ64 = Vector{Float64}(θ)
    losses = zeros(Float64, e.K)
    tg = zeros(Float64, e.P, e.K)                    # C row-major K × P == Julia column-major P × K
    GC.@preserve θ64 losses tg check(ccall(sym(:pinn_term_grads_f64), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}),
                                           e.h, θ64, e.P, losses, tg), "pinn_term_grads_f64")
    return losses, tg
end

"`residual_k(set_k, θ)`: the datafree loss function of src/discretize.jl:174 on the installed set (1 × N like the reference)."
function residual(e::HIPEngine, k::Integer, θ::AbstractVector{<:Real}, n::Integer)
    θ64 = Vector{Float64}(θ); r = zeros(Float64, n)
    GC.@preserve θ64 r check(ccall(sym(:pinn_residual_f64), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Ptr{Float64}), e.h, k - 1, θ64, e.P, r),
                             "pinn_residual_f64")
    return reshape(r, 1, :)
end

"`phi(x, θ)` of network `net` (1-based) through the engine (src/pinn_types.jl:88-90)."
function phi(e::HIPEngine, net::Integer, θ::AbstractVector{<:Real}, x::AbstractMatrix)
    θ64 = Vector{Float64}(θ); x64 = Matrix{Float64}(x); out = zeros(Float64, size(x64, 2))
    GC.@preserve θ64 x64 out check(ccall(sym(:pinn_phi_f64), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}),
                                         e.h, net - 1, θ64, e.P, x64, size(x64, 2), out), "pinn_phi_f64")
    return reshape(out, 1, :)
end

"""
    derivative(e, net, θ, x, axes) -> 1 × N

`numeric_derivative(phi, u, x, εs, order, θ)` of the reference (src/pinn_types.jl:445-482) through the engine: the EXACT derivative of the trial
function of network `net` along `axes` (1-based input axes, `length(axes)` = order) — the value the reference's central differences
approximate to atol 1e-8 (order 1) / 4e-5 (order 2) in Float64 (test/Forward/forward__derivatives.jl:29-44), met at those tolerances by an
engine in float64 mode (tests/test_f64_mode.py).
"""
function derivative(e::HIPEngine, net::Integer, θ::AbstractVector{<:Real}, x::AbstractMatrix, axes::AbstractVector{<:Integer})
    θ64 = Vector{Float64}(θ); x64 = Matrix{Float64}(x); out = zeros(Float64, size(x64, 2))
    ax = Cint[a - 1 for a in axes]
    isempty(ax) && return phi(e, net, θ, x)
    GC.@preserve θ64 x64 ax out check(ccall(sym(:pinn_derivative_f64), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Cint, Ptr{Cint}, Ptr{Float64}),
        e.h, net - 1, θ64, e.P, x64, size(x64, 2), length(ax), ax, out), "pinn_derivative_f64")
    return reshape(out, 1, :)
end

"""
    loglik_grad(e, θ, stds) -> (loglik, ∇θ, ∂/∂stds)

BPINN physics (+ data) log-likelihood `Σ_k logpdf(MvNormal(r_k, σ_k² I), 0)` and its gradients in ONE device evaluation
(`pinn_loglik_grad_f64`; src/training_strategies.jl:113-127, ext/bpinn/PDE_BPINN.jl:16-26) — what an HMC leapfrog step needs instead of the
reference's ForwardDiff sweep over all P parameters (ext/bpinn/PDE_BPINN.jl:519).
"""
function loglik_grad(e::HIPEngine, θ::AbstractVector{<:Real}, stds::AbstractVector{<:Real})
    θ64 = Vector{Float64}(θ); sd = Vector{Float64}(stds)
    length(sd) == e.K || throw(DimensionMismatch("need one std per loss term ($(e.K))"))
    ll = Ref{Float64}(0.0); g = zeros(Float64, e.P); gs = zeros(Float64, e.K)
    GC.@preserve θ64 sd g gs check(ccall(sym(:pinn_loglik_grad_f64), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}),
        e.h, θ64, e.P, sd, ll, g, gs), "pinn_loglik_grad_f64")
    return ll[], g, gs
end

# ------------------------------------------------------------------------------------------------
# 1. s-expression printer of the Julia Expr trees `toexpr` returns
# ------------------------------------------------------------------------------------------------
"""
    sexpr(ex) -> String

Prefix form of a `toexpr` tree: `Expr(:call, f, args...)` -> `(name args...)` with `name` the function's name (`+ - * / ^ sin …`), a
dependent-variable Symbol (`(u x y)`, `(u 0 y)`), or `D` for a `Differential` head: `(D x 2 (u x y))` (variable, `order` field,
operand; nested Differentials nest).  Numbers print as literals (`π` as `pi`, rationals as `a//b`), Symbols by name.

Shapes of the five BASELINE configurations (up to Symbolics' own term order, which the library does not depend on) — the same
strings are committed under tests/golden/descriptors/ and exercised by tests/test_sexpr_frontend.py:

    cfg2 (2-D Poisson)  lhs (+ (D x 2 (u x y)) (D y 2 (u x y)))            rhs (* -1 (sin (* pi x)) (sin (* pi y)))
    cfg3 (Burgers)      lhs (+ (D t 1 (u t x)) (* (u t x) (D x 1 (u t x))) (* -0.01 (/ 1 pi) (D x 2 (u t x))))   rhs 0
    cfg5 (heat, κ est.) lhs (D t 1 (u t x y z))     rhs (* kappa (+ (D x 2 (u t x y z)) (D y 2 (u t x y z)) (D z 2 (u t x y z))))
    boundary terms      lhs (u 0 y)   rhs 0          (call arguments are dropped by the lowering, symbolic_utilities.jl:145-160)
"""
sexpr(x::Irrational{:π}) = "pi"
sexpr(x::Irrational{:ℯ}) = string(Float64(x))
sexpr(x::Rational) = string(numerator(x), "//", denominator(x))
sexpr(x::Integer) = string(x)
sexpr(x::AbstractFloat) = repr(Float64(x))
sexpr(x::Real) = repr(Float64(x))
sexpr(x::Symbol) = string(x)
function sexpr(ex::Expr)
    ex.head === :call || throw(HIPEngineError("cannot print expression head $(ex.head) (only calls occur in toexpr output): $ex"))
    f = ex.args[1]
    args = ex.args[2:end]
    if f isa Differential
        length(args) == 1 || throw(HIPEngineError("Differential with $(length(args)) operands"))
        order = hasproperty(f, :order) ? Int(f.order) : 1
        return string("(D ", sexpr(toexpr(f.x)), " ", order, " ", sexpr(args[1]), ")")
    end
    name = f isa Symbol ? string(f) : string(nameof(f))
    name == "σ" && (name = "sigmoid")
    return string("(", name, isempty(args) ? "" : " ", join((sexpr(a) for a in args), " "), ")")
end
sexpr(x) = sexpr(toexpr(x))          # Symbolics objects (Num, BasicSymbolic)

"lhs / rhs of an equation as the reference's `parse_equation` sees them (src/symbolic_utilities.jl:360-364)."
function equation_sexprs(eq)
    l = SymbolicUtils._iszero(expand_derivatives(eq.lhs)) ? eq.lhs : expand_derivatives(eq.lhs)
    r = SymbolicUtils._iszero(expand_derivatives(eq.rhs)) ? eq.rhs : expand_derivatives(eq.rhs)
    return sexpr(toexpr(l)), sexpr(toexpr(r))
end

# ------------------------------------------------------------------------------------------------
# 2. descriptor ("pinnir 2")
# ------------------------------------------------------------------------------------------------
const ACT_NAMES = Dict{Any, String}(tanh => "tanh", Lux.tanh_fast => "tanh", Lux.sigmoid => "sigmoid", Lux.sigmoid_fast => "sigmoid",
                                    sin => "sin", identity => "identity")

chains_of(pinnrep::PINNRepresentation) = pinnrep.phi isa AbstractVector ? [p.smodel.model for p in pinnrep.phi] : [pinnrep.phi.smodel.model]

act_name(f) = get(ACT_NAMES, f, nothing)

# the reference's DGM(in, 1, modes, L, activation1, activation2, identity) (src/dgm.jl:97-115): Chain(SkipConnection(Dense, DGMLSTMBlock), Dense)
function dgm_lines(i::Int, dgm, θoff::Int, depvar::Symbol, inputs)
    outer = collect(values(dgm.model.layers))
    first_dense, block, last_dense = outer[1].layers, outer[1].connection, outer[2]
    gated = collect(values(block.layers))
    lstm = gated[1] isa Lux.SkipConnection ? gated[1].layers : gated[1]
    a1, a2 = act_name(lstm.activation1), act_name(lstm.activation2)
    (a1 === nothing || a2 === nothing || a1 == "identity" || a2 == "identity") &&
        throw(HIPEngineError("unsupported DGM activations ($(lstm.activation1), $(lstm.activation2)); supported: tanh, sigmoid, sin"))
    act_name(first_dense.activation) == a1 || throw(HIPEngineError("DGM: the first Dense layer must use activation1"))
    act_name(last_dense.activation) == "identity" || throw(HIPEngineError("the HIP engine runs DGM networks with the identity output activation"))
    last_dense.out_dims == 1 || throw(HIPEngineError("each network must have a single output (one per dependent variable)"))
    d, M, L = first_dense.in_dims, first_dense.out_dims, length(gated)
    nparams = M * d + M + L * (4 * M * d + 4 * M * M + 4 * M) + M + 1
    nparams == Lux.LuxCore.parameterlength(dgm) || throw(HIPEngineError("DGM parameter count differs from the engine's layout"))
    return ["net $(i - 1) dgm,$a1,$a2,$L $θoff 3 $d $M 1", "netvar $(i - 1) $depvar $(length(inputs)) " * join(inputs, " ")], nparams
end

# [3P] Boltz.Layers.PeriodicEmbedding(idxs, periods) as the first layer (test/CUDA/nnpde_cuda__1d_pde_dirichlet_bc_cuda.jl:26,48), recognised
# by name and fields so that Boltz does not become a dependency of this file: inputs `idxs` leave the input list and come back at its end
# as their sines, then their cosines (period-scaled); the layer has no parameters, so the flat-θ layout is that of the Dense stack
is_periodic_embedding(l) = nameof(typeof(l)) === :PeriodicEmbedding && hasproperty(l, :idxs) && hasproperty(l, :periods)
function input_dim(model)
    model isa NeuralPDE.DGM && return first(values(model.model.layers)).layers.in_dims
    layers = collect(values(model.layers))
    is_periodic_embedding(layers[1]) && return layers[2].in_dims - length(layers[1].idxs)
    return layers[1].in_dims
end

function chain_lines(i::Int, chain, θoff::Int, depvar::Symbol, inputs)
    chain isa NeuralPDE.DGM && return dgm_lines(i, chain, θoff, depvar, inputs)
    layers = collect(values(chain.layers))
    embed = nothing
    if is_periodic_embedding(layers[1])
        embed, layers = layers[1], layers[2:end]
        length(embed.idxs) == length(embed.periods) || throw(HIPEngineError("PeriodicEmbedding: one period per embedded input"))
        Lux.LuxCore.parameterlength(embed) == 0 || throw(HIPEngineError("PeriodicEmbedding with parameters is not supported"))
    end
    all(l -> l isa Lux.Dense, layers) || throw(HIPEngineError("the HIP engine runs Chains of Dense layers, optionally behind a PeriodicEmbedding (got $(typeof.(layers)))"))
    length(layers) >= 2 || throw(HIPEngineError("the HIP engine needs at least one hidden layer"))
    acts = [act_name(l.activation) for l in layers]
    any(isnothing, acts) && throw(HIPEngineError("unsupported activation in chain $i: $([l.activation for l in layers]) (supported: tanh, sigmoid, sin)"))
    acts[end] == "identity" || throw(HIPEngineError("the last layer must have identity activation"))
    layers[end].out_dims == 1 || throw(HIPEngineError("each chain must have a single output (one chain per dependent variable, src/pinn_types.jl:106-108)"))
    all(l -> Lux.LuxCore.parameterlength(l) == l.in_dims * l.out_dims + l.out_dims, layers) ||
        throw(HIPEngineError("Dense layers without bias are not supported"))
    hidden = acts[1:(end - 1)]
    act = all(==(hidden[1]), hidden) ? hidden[1] : join(hidden, ",")
    sizes = vcat(layers[1].in_dims, [l.out_dims for l in layers])
    nparams = sum(l.in_dims * l.out_dims + l.out_dims for l in layers)
    lines = ["net $(i - 1) $act $θoff $(length(sizes)) " * join(sizes, " ")]
    if embed !== nothing              # embed <net> <n> <0-based input index> <period> ...   (csrc/descriptor.cpp: apply_embeddings)
        sizes[1] == length(inputs) + length(embed.idxs) ||
            throw(HIPEngineError("the first Dense layer must take $(length(inputs) + length(embed.idxs)) features (inputs + embedded inputs), got $(sizes[1])"))
        push!(lines, "embed $(i - 1) $(length(embed.idxs)) " * join(("$(Int(ix) - 1) $(repr(Float64(p)))" for (ix, p) in zip(embed.idxs, embed.periods)), " "))
    end
    push!(lines, "netvar $(i - 1) $depvar $(length(inputs)) " * join(inputs, " "))
    return lines, nparams
end

"""
    descriptor(pinnrep; hints = Int[]) -> String

The "pinnir 2" text handed to `pinn_create` (grammar: DESIGN.md §2).  θ layout = `pinnrep.flat_init_params` (src/discretize.jl:451-465):
`[depvar 1: W1 (out×in, column-major) | b1 | W2 | b2 … | depvar 2 … | p]` — the order ComponentArrays flattens Lux's
`(layer_1 = (weight, bias), …)` NamedTuples in; `verify_layout` checks it numerically.  Coordinate rows of term k =
`this_eq_indvars` of `build_symbolic_loss_function` (src/discretize.jl:41-43), computed with the reference's own `pair`.
"""
function descriptor(pinnrep::PINNRepresentation; hints::AbstractVector{<:Integer} = Int[])
    (; eqs, bcs, eq_params, default_p, param_estim, depvars, dict_depvars, dict_depvar_input, flat_init_params) = pinnrep
    chains = chains_of(pinnrep)
    length(chains) == length(depvars) || throw(HIPEngineError("$(length(depvars)) dependent variables need $(length(depvars)) single-output chains"))
    has_p = !(eq_params isa SciMLBase.NullParameters)
    np = has_p ? length(eq_params) : 0
    ne = (param_estim && has_p) ? np : 0
    lines = String[]
    netlines = String[]
    off = 0
    for (i, ch) in enumerate(chains)
        l, n = chain_lines(i, ch, off, depvars[i], dict_depvar_input[depvars[i]])
        append!(netlines, l)
        off += n
    end
    ntheta = length(flat_init_params)
    ntheta == off + ne || throw(HIPEngineError("flat_init_params has $ntheta entries, the chains (+ θ.p) need $(off + ne)"))
    push!(lines, "pinnir 2", "ntheta $ntheta", "params $np $ne $off",
          "defaults " * join((repr(Float64(v)) for v in (has_p ? default_p : Float64[])), " "),
          "pnames " * join((string(Symbol(toexpr(p))) for p in (has_p ? eq_params : [])), " "),
          "nets $(length(chains))")
    append!(lines, netlines)
    terms = vcat(collect(eqs isa AbstractArray ? eqs : [eqs]), collect(bcs))
    push!(lines, "terms $(length(terms))")
    for (k, eq) in enumerate(terms)
        this_eq_pair = NeuralPDE.pair(eq, depvars, dict_depvars, dict_depvar_input)
        this_eq_indvars = unique(vcat(values(this_eq_pair)...))
        isempty(this_eq_indvars) && throw(ArgumentError("equation $k does not contain a dependent variable"))
        l, r = equation_sexprs(eq)
        push!(lines, "sterm $(k - 1) $(length(this_eq_indvars)) " * join(this_eq_indvars, " "), "lhs " * l, "rhs " * r)
    end
    # optional tail: `hint <term> <points>` — the sizes of the sets about to be installed; lets the planner put a boundary condition of a
    # few points onto a launch its network already has instead of giving it a launch of its own (csrc/plan.cpp)
    for (k, n) in enumerate(hints)
        n > 0 && push!(lines, "hint $(k - 1) $n")
    end
    return join(lines, "\n") * "\n"
end

"θ-layout check: the engine's trial functions must equal the Lux chains on random points (throws otherwise)."
function verify_layout(e::HIPEngine, pinnrep::PINNRepresentation; n::Int = 16, rtol = 1.0e-4)
    θ = pinnrep.flat_init_params
    flat = collect(Float64, ComponentArrays.getdata(θ))
    phis = pinnrep.phi isa AbstractVector ? pinnrep.phi : [pinnrep.phi]
    for (i, ph) in enumerate(phis)
        d = input_dim(ph.smodel.model)
        x = rand(Float64, d, n)
        # src/discretize.jl:451-465: multioutput => θ.depvar.<name>; single chain => θ itself, or θ.depvar when param_estim adds θ.p
        θi = pinnrep.multioutput ? getproperty(θ.depvar, pinnrep.depvars[i]) : (pinnrep.param_estim ? θ.depvar : θ)
        ref = ph(x, θi)
        got = phi(e, i, flat, x)
        err = maximum(abs.(got .- ref)) / max(maximum(abs.(ref)), 1.0e-12)
        err <= rtol || throw(HIPEngineError("θ layout mismatch for dependent variable $(pinnrep.depvars[i]): engine and Lux chain differ by $err"))
    end
    return true
end

# ------------------------------------------------------------------------------------------------
# 3a. plug-in point (1): training strategy
# ------------------------------------------------------------------------------------------------
"""
    HIPStrategy(inner; precision = :auto)

`precision`: `:auto` (default) = compute dtype follows `eltype(θ)` as in the reference (src/eltype_matching.jl:8-10) — Float64 parameters,
the reference's default, run the float64 kernels, Float32 `init_params` the fp32 kernels; `:f32` = the fp32 kernels whatever eltype(θ)
(the explicit fast opt-in); `:f64` = the float64 kernels (`resolve_precision`).

`inner` is the reference strategy whose POINT SETS are used (`GridTraining`, `StochasticTraining`, `QuasiRandomTraining`); the residual,
`mean(abs2, ·)` and the gradient run on the engine.  Use it wherever a strategy goes:
`PhysicsInformedNN(chain, HIPStrategy(QuasiRandomTraining(65_536)))`.
"""
struct HIPStrategy{S} <: AbstractTrainingStrategy
    inner::S
    precision::Symbol        # :auto (compute dtype = eltype(θ), the reference's contract) | :f32 (explicit fast opt-in) | :f64
end
HIPStrategy(inner; precision::Symbol = :auto) = HIPStrategy(inner, precision)

# shared state of all closures of one discretisation: one fused evaluation per θ serves every term
mutable struct HIPState
    engine::HIPEngine
    pinnrep::PINNRepresentation
    n_pde::Int
    sets::Vector{Matrix{Float64}}                # current set of every term (fixed strategies), pde terms first
    resample::Union{Nothing, Function}           # () -> Vector{Matrix}: fresh sets (resampling strategies), called once per new θ
    key::UInt64
    losses::Vector{Float64}
    grad::Vector{Float64}                        # of Σ w_k L_k under `weights`; empty after a loss-only evaluation
    weights::Vector{Float64}
    tgrads::Union{Nothing, Matrix{Float64}}      # P × K per-term gradients (filled on the first per-term pullback at a θ)
    tgrads_key::UInt64                           # hash(θ) the per-term gradients were computed at (per-term gradients do not depend on the weights);
                                                 # pullbacks may run out of order (nested AD, several closures / threads), so it is compared, not st.key
    eager_grad::Bool                             # true (hip_discretize): a plain call of a term closure already runs the fused loss + gradient
                                                 # evaluation that the optimiser's `grad!` of the same iterate will ask for
    lock::ReentrantLock                          # the closures of one discretisation share this state; BPINN calls them from
                                                 # Threads.@threads (ext/bpinn/PDE_BPINN.jl:548)
end

function current_weights(st::HIPState)
    a = st.pinnrep.adaloss
    wp = Float64.(a.pde_loss_weights); wb = Float64.(a.bc_loss_weights)
    K = st.engine.K
    # before src/discretize.jl:553-559 has broadcast them, the weights may still be scalars / length-1 vectors
    w = vcat(length(wp) == st.n_pde ? wp : fill(first(wp), st.n_pde), length(wb) == K - st.n_pde ? wb : fill(first(wb), K - st.n_pde))
    return w
end

"""
One device evaluation per (θ, weights), shared by every closure of the discretisation.  `want_grad = false`: only the term losses are
needed (a callback, an adaptive-weight rule, a plain call of a term closure outside AD — the reference's closures are value-only unless
differentiated, src/training_strategies.jl:215-221) — the engine's loss-only evaluation; a later request for the gradient at the same θ
upgrades the memo with one fused evaluation.  Returns `(losses, grad, weights)` copied out under the lock.
"""
function evaluate!(st::HIPState, θ; want_grad::Bool = true)
    flat = collect(Float64, ComponentArrays.getdata(θ))
    lock(st.lock) do
        w = current_weights(st)
        key = hash(flat, hash(w))
        fresh = key != st.key
        if fresh && st.resample !== nothing                         # fresh sets on every new θ (src/training_strategies.jl:277-281, 375-381)
            st.sets = st.resample()
            for (k, s) in enumerate(st.sets)
                set_points!(st.engine, k, s)
            end
        end
        if fresh || (want_grad && isempty(st.grad))
            st.losses, st.grad = loss_grad(st.engine, flat, w; want_grad = want_grad || st.eager_grad)
            st.weights, st.key = w, key
            fresh && (st.tgrads = nothing)
        end
        return (copy(st.losses), st.grad, st.weights)
    end
end

"`θ -> mean(abs2, residual_k(set_k, θ))` (src/training_strategies.jl:220, 280, 380) served by the engine."
struct HIPTermLoss <: Function
    st::HIPState
    k::Int
end
(f::HIPTermLoss)(θ) = evaluate!(f.st, θ; want_grad = false)[1][f.k]

function ChainRulesCore.rrule(f::HIPTermLoss, θ)
    st = f.st
    y = evaluate!(st, θ; want_grad = false)[1][f.k]
    flat = collect(Float64, ComponentArrays.getdata(θ))
    function term_pullback(ȳ)
        tg = lock(st.lock) do
            if st.tgrads === nothing || st.tgrads_key != hash(flat)
                _, st.tgrads = term_grads(st.engine, flat)
                st.tgrads_key = hash(flat)
            end
            copy(view(st.tgrads, :, f.k))
        end
        g = tg .* ChainRulesCore.unthunk(ȳ)
        return NoTangent(), θ isa ComponentArray ? ComponentArray(g, ComponentArrays.getaxes(θ)) : g
    end
    return y, term_pullback
end

point_sets(pinnrep, s::GridTraining) = begin
    (; domains, eqs, bcs, dict_indvars, dict_depvars) = pinnrep
    pde, bc = NeuralPDE.generate_training_sets(domains, s.dx, eqs, bcs, Float64, dict_indvars, dict_depvars)
    (Matrix{Float64}[Matrix{Float64}(m) for m in vcat(pde, bc)], nothing)
end
function point_sets(pinnrep, s::StochasticTraining)
    (; domains, eqs, bcs, dict_indvars, dict_depvars) = pinnrep
    pde_b, bc_b = NeuralPDE.get_bounds(domains, eqs, bcs, Float64, dict_indvars, dict_depvars, s)
    draw() = vcat(Matrix{Float64}[NeuralPDE.generate_random_points(s.points, b, Float64) for b in pde_b],
                  Matrix{Float64}[NeuralPDE.generate_random_points(s.bcs_points, b, Float64) for b in bc_b])
    return draw(), draw
end
function point_sets(pinnrep, s::QuasiRandomTraining)
    (; domains, eqs, bcs, dict_indvars, dict_depvars) = pinnrep
    pde_b, bc_b = NeuralPDE.get_bounds(domains, eqs, bcs, Float64, dict_indvars, dict_depvars, s)
    design() = vcat(Matrix{Float64}[QuasiMonteCarlo.sample(s.points, b[1], b[2], s.sampling_alg) for b in pde_b],
                    Matrix{Float64}[QuasiMonteCarlo.sample(s.bcs_points, b[1], b[2], s.sampling_alg) for b in bc_b])
    s.resampling && return design(), design
    nb = max(s.minibatch, 1)
    batches = [design() for _ in 1:nb]
    return batches[1], (nb == 1 ? nothing : () -> batches[rand(1:nb)])      # src/training_strategies.jl:383-387
end
point_sets(pinnrep, s) = throw(HIPEngineError("HIPStrategy wraps GridTraining, StochasticTraining or QuasiRandomTraining (got $(typeof(s))); " *
                                              "QuadratureTraining's adaptive cubature is a host algorithm"))

function build_state(pinnrep::PINNRepresentation, inner; precision::Symbol = :auto)
    sets, resample = point_sets(pinnrep, inner)
    engine = HIPEngine(descriptor(pinnrep; hints = [size(s, 2) for s in sets]))
    # compute dtype = eltype(θ) unless the caller opted into :f32 / :f64 (resolve_precision); the float64 mode refuses what it does not cover HERE
    resolve_precision(precision, pinnrep.flat_init_params) === :f64 && set_precision!(engine, :f64)
    verify_layout(engine, pinnrep)
    for (k, s) in enumerate(sets)
        set_points!(engine, k, s)
    end
    n_pde = pinnrep.eqs isa AbstractArray ? length(pinnrep.eqs) : 1
    return HIPState(engine, pinnrep, n_pde, sets, resample, UInt64(0), Float64[], Float64[], Float64[], nothing, UInt64(0), false, ReentrantLock())
end

# The state (and with it the engine and its HBM) lives exactly as long as the closures that `merge_strategy_with_loss_function` returns:
# they end up in `pinnrep.loss_functions.pde_loss_functions` / `.bc_loss_functions` (src/discretize.jl:760-764), from where
# `state_of` reads it back — no global table that would keep every engine of the session alive.
function state_of(pinnrep::PINNRepresentation)
    fs = vcat(collect(pinnrep.loss_functions.pde_loss_functions), collect(pinnrep.loss_functions.bc_loss_functions))
    i = findfirst(f -> f isa HIPTermLoss, fs)
    i === nothing && throw(HIPEngineError("this PINNRepresentation was not built with a HIPStrategy"))
    return fs[i].st
end

function NeuralPDE.merge_strategy_with_loss_function(pinnrep::PINNRepresentation, strategy::HIPStrategy,
                                                    datafree_pde_loss_function, datafree_bc_loss_function)
    st = build_state(pinnrep, strategy.inner; precision = strategy.precision)
    n_pde, n_bc = length(datafree_pde_loss_function), length(datafree_bc_loss_function)
    n_pde + n_bc == st.engine.K || throw(HIPEngineError("engine has $(st.engine.K) terms, the discretisation $(n_pde + n_bc)"))
    return [HIPTermLoss(st, k) for k in 1:n_pde], [HIPTermLoss(st, n_pde + j) for j in 1:n_bc]
end

# ------------------------------------------------------------------------------------------------
# 3b. plug-in point (2): the fast path — explicit gradient, one fused device call per optimiser iteration
# ------------------------------------------------------------------------------------------------
"""
    prob = hip_discretize(pde_system, discretization::PhysicsInformedNN; precision = :auto)

Same `OptimizationProblem` as `discretize` (src/discretize.jl:776-780: objective `full_loss_function`, `u0 = flat_init_params`), but the
`OptimizationFunction` carries an explicit `grad`: the engine's fused `∇θ Σ_k w_k L_k` (+ the Zygote gradient of `additional_loss`, which
stays a Julia function).  The objective itself is the reference's own `full_loss_function`, so the iteration counter, adaptive
reweighting and logging behave as always; value and gradient of one iterate share ONE device evaluation (memoised on θ and weights).
"""
function hip_discretize(pde_system, discretization::PhysicsInformedNN; precision::Symbol = :auto)
    strat = discretization.strategy isa HIPStrategy ? discretization.strategy : HIPStrategy(discretization.strategy; precision = precision)
    disc = discretization.strategy isa HIPStrategy ? discretization : rebuild(discretization, strat)
    pinnrep = SciMLBase.symbolic_discretize(pde_system, disc)
    st = state_of(pinnrep)
    st.eager_grad = true                 # value and gradient of one iterate share ONE fused evaluation, whichever the optimiser asks for first
    full = pinnrep.loss_functions.full_loss_function
    function grad!(G, θ, p)
        _, g, _ = evaluate!(st, θ; want_grad = true)      # memoised on (θ, current weights): the reweighting of this iterate has already run
        G .= g
        if pinnrep.additional_loss !== nothing
            wa = pinnrep.adaloss.additional_loss_weights[1]
            ga = first(NeuralPDE.Zygote.gradient(θ) do t      # (NeuralPDE imports Zygote, src/NeuralPDE.jl:36; Optimization.jl does not export it)
                (t_, p_) = pinnrep.param_estim ? (t.depvar, t.p) : (t, nothing)
                pinnrep.additional_loss(pinnrep.phi, t_, p_)
            end)
            ga === nothing || (G .+= wa .* ComponentArrays.getdata(ga))
        end
        return G
    end
    f = Optimization.OptimizationFunction(full; grad = grad!)
    return Optimization.OptimizationProblem(f, pinnrep.flat_init_params)
end

# PhysicsInformedNN is an immutable struct of the reference (src/pinn_types.jl:147-163): the same discretisation with another strategy,
# through its positional constructor so that phi, the iteration counter and every other field are shared, not rebuilt
function rebuild(d::PhysicsInformedNN, strategy)
    return PhysicsInformedNN(d.chain, strategy, d.init_params, d.init_states, d.phi, d.derivative, d.param_estim, d.additional_loss,
                             d.adaptive_loss, d.logger, d.log_options, d.iteration, d.self_increment, d.multioutput, d.kwargs)
end

# ------------------------------------------------------------------------------------------------
# 4. multi-GPU (single Julia process, G devices): point shards + the engine's RCCL all-reduce
# ------------------------------------------------------------------------------------------------
"""
    ShardedEngine(desc, sets; devices = 0:G-1)

One engine per device (`pinn_create_on`), every term's set split into contiguous column blocks (`n_norm` = global N), one
communicator (`pinn_comm_init_all`); `loss_grad(se, θ, w)` = `pinn_loss_grad_sharded`: the global losses and gradient (SURVEY.md §8e).
"""
struct ShardedEngine
    engines::Vector{HIPEngine}
end
function ShardedEngine(desc::AbstractString, sets::Vector{<:AbstractMatrix}; devices = 0:0, precision::Symbol = :auto)
    engines = [HIPEngine(desc; device = d) for d in devices]
    G = length(engines)
    # precision as everywhere else: the sets' eltype decides (Float64 sets -> the float64 evaluation mode on every device, the communicator
    # then carries [P + K] doubles: pinn_loss_grad_sharded_f64 / the double resident loop, r06)
    prec = precision === :auto ? (eltype(first(sets)) === Float64 ? :f64 : :f32) : precision
    prec === :f64 && foreach(e -> set_precision!(e, :f64), engines)
    for (g, e) in enumerate(engines), (k, s) in enumerate(sets)
        n = size(s, 2)
        lo, hi = (n * (g - 1)) ÷ G + 1, (n * g) ÷ G
        set_points!(e, k, s[:, lo:hi]; n_norm = n)
    end
    hs = [e.h for e in engines]
    GC.@preserve hs check(ccall(sym(:pinn_comm_init_all), Cint, (Ptr{Ptr{Cvoid}}, Cint), hs, G), "pinn_comm_init_all")
    return ShardedEngine(engines)
end
function loss_grad(se::ShardedEngine, θ::AbstractVector{<:Real}, w::AbstractVector{<:Real})
    e = se.engines[1]
    if e.precision === :f64
        θ64 = Vector{Float64}(θ); w64 = Vector{Float64}(w)
        losses = zeros(Float64, e.K); grad = zeros(Float64, e.P)
        hs = [x.h for x in se.engines]
        GC.@preserve hs θ64 w64 losses grad check(ccall(sym(:pinn_loss_grad_sharded_f64), Cint,
            (Ptr{Ptr{Cvoid}}, Cint, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            hs, length(hs), θ64, e.P, w64, losses, grad), "pinn_loss_grad_sharded_f64")
        return losses, grad
    end
    θ32 = Vector{Float32}(θ); w32 = Vector{Float32}(w)
    losses = zeros(Float64, e.K); grad = zeros(Float32, e.P)
    hs = [x.h for x in se.engines]
    GC.@preserve hs θ32 w32 losses grad check(ccall(sym(:pinn_loss_grad_sharded), Cint,
        (Ptr{Ptr{Cvoid}}, Cint, Ptr{Float32}, Int64, Ptr{Float32}, Ptr{Float64}, Ptr{Float32}),
        hs, length(hs), θ32, e.P, w32, losses, grad), "pinn_loss_grad_sharded")
    return losses, Float64.(grad)
end

"""
    adam!(se::ShardedEngine, θ0, nsteps, η; w = ones(K)) -> (θ, loss_history)

`solve(prob, Adam(η); maxiters = nsteps)` resident on the devices of the communicator (`pinn_adam_steps_sharded`): per iteration every
device evaluates its shards, ONE all-reduce sums `[gradient | per-term sums]`, every device applies the same fused Adam update — θ, the
moments and the point sets never leave HBM and the host is not synchronised inside the loop.
"""
function adam!(se::ShardedEngine, θ0::AbstractVector{<:Real}, nsteps::Integer, η::Real; w::AbstractVector{<:Real} = ones(se.engines[1].K),
               β1::Real = 0.9, β2::Real = 0.999, ϵ::Real = 1.0e-8)
    e = se.engines[1]
    θ32 = Vector{Float32}(θ0); w32 = Vector{Float32}(w)
    f64 = e.precision === :f64
    θ64 = Vector{Float64}(θ0)
    for x in se.engines
        if f64
            GC.@preserve θ64 check(ccall(sym(:pinn_adam_init_f64), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), x.h, θ64, e.P), "pinn_adam_init_f64")
        else
            GC.@preserve θ32 check(ccall(sym(:pinn_adam_init), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), x.h, θ32, e.P), "pinn_adam_init")
        end
    end
    hist = zeros(Float64, nsteps)
    hs = [x.h for x in se.engines]
    GC.@preserve hs w32 hist check(ccall(sym(:pinn_adam_steps_sharded), Cint,
        (Ptr{Ptr{Cvoid}}, Cint, Cint, Cfloat, Cfloat, Cfloat, Cfloat, Ptr{Float32}, Ptr{Float64}),
        hs, length(hs), nsteps, η, β1, β2, ϵ, w32, hist), "pinn_adam_steps_sharded")
    if f64
        out64 = zeros(Float64, e.P)
        GC.@preserve out64 check(ccall(sym(:pinn_adam_get_f64), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), e.h, out64, e.P), "pinn_adam_get_f64")
        return out64, hist
    end
    out = zeros(Float32, e.P)
    GC.@preserve out check(ccall(sym(:pinn_adam_get), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64), e.h, out, e.P), "pinn_adam_get")
    return Float64.(out), hist
end

# ------------------------------------------------------------------------------------------------
# 5. self test (run once where Julia, NeuralPDE and the library are all present)
# ------------------------------------------------------------------------------------------------
"""
    selftest(pde_system, discretization; rtol = 1e-6)

Builds the problem twice — with the reference's own strategy (generated Julia loss functions, finite-difference `numeric_derivative`)
and with the engine — and compares, on the SAME point sets and θ, every datafree residual (`pinn_residual_f64` against
`pinnrep.loss_functions.datafree_*`, the reference's generated functions with their finite-difference `numeric_derivative`).
This is the check that pins `descriptor` / `sexpr` / the θ layout against the real reference objects; returns the worst relative
difference.  With Float64 θ the engine runs its float64 kernels in the reference-semantics mode (`set_derivative!(e, :stencil)`, r06): the
SAME central differences with the SAME `get_ε` steps, so the two sides differ only by the rounding noise of the difference formulas
(~1e-16 |u| / ε² per point: 1e-8 … 1e-7 for second derivatives) — `rtol = 1e-6` by default; with Float32 θ (fp32 kernels, exact derivatives
against Float32 stencils with ε = 0.02) pass `rtol = 1e-2`.
"""
function selftest(pde_system, discretization::PhysicsInformedNN; rtol = 1.0e-6)
    ref = SciMLBase.symbolic_discretize(pde_system, discretization)
    st = build_state(ref, GridTraining(0.1))                  # engine built from the REFERENCE's pinnrep, on its GridTraining(0.1) sets; precision :auto = eltype(θ)
    st.engine.precision === :f64 && set_derivative!(st.engine, :stencil)
    θ = ref.flat_init_params
    flat = collect(Float64, ComponentArrays.getdata(θ))
    dfs = vcat(ref.loss_functions.datafree_pde_loss_functions, ref.loss_functions.datafree_bc_loss_functions)
    worst = 0.0
    for (k, df) in enumerate(dfs)
        cord = st.sets[k]
        set_points!(st.engine, k, cord)
        r_ref = df(cord, θ)
        r_hip = residual(st.engine, k, flat, size(cord, 2))
        worst = max(worst, maximum(abs.(r_hip .- r_ref)) / max(maximum(abs.(r_ref)), 1.0))
    end
    worst <= rtol || throw(HIPEngineError("residuals differ from the reference's generated functions by $worst"))
    return worst
end

end # module
