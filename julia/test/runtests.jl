# julia/test/runtests.jl — the checks a maintainer runs ONCE on a machine that has Julia, NeuralPDE.jl (v6.2.x), Lux and a built
# libpinn_hip.so with a gfx950 device (the build container of this repository has no Julia, so NeuralPDEHIP.jl ships unexecuted):
#
#     PINN_HIP_LIB=/path/to/libpinn_hip.so julia --project=<env with NeuralPDE> julia/test/runtests.jl
#
# For each of the five BASELINE configurations (the same problem statements as neuralpde.jl_amd/workloads.py):
#   1. `descriptor(pinnrep)`: the structural lines (θ length, parameters, nets, θ offsets, coordinate order of every term) must be
#      STRING-EQUAL to tests/golden/descriptors/cfgN.pinnir2 (written by the Python mirror and exercised by the test-suite); the lhs / rhs
#      s-expressions may differ in Symbolics' term order, so they are compared by evaluation: an engine built from the Julia descriptor and
#      one built from the golden file must return the same residuals on the same θ and points;
#   2. `verify_layout` (runs inside `build_state`): the engine's trial functions equal the Lux chains — the flat-θ layout
#      (src/discretize.jl:451-465) is what SURVEY.md App. D states;
#   3. `selftest`: every datafree residual of the engine against the reference's own generated loss functions
#      (finite-difference `numeric_derivative`, src/pinn_types.jl:445-482) on the same sets;
#   4. `HIPStrategy` through `discretize` + one Zygote gradient of the full loss (the per-term rrules), and `hip_discretize` through a few
#      Adam iterations; a loss-only call (`HIPTermLoss(θ)` outside AD) against the fused evaluation.
using Test, Random
using NeuralPDE, Lux, ModelingToolkit, Optimization, OptimizationOptimisers, ComponentArrays, QuasiMonteCarlo
import ModelingToolkit: Interval
using Zygote
include(joinpath(@__DIR__, "..", "NeuralPDEHIP.jl"))
using .NeuralPDEHIP

const GOLDEN = joinpath(@__DIR__, "..", "..", "tests", "golden", "descriptors")
mlp(d, w, h; act = tanh) = Chain(Dense(d, w, act), [Dense(w, w, act) for _ in 1:(h - 1)]..., Dense(w, 1))

function cfg1()
    @parameters x
    @variables u(..)
    eq = Differential(x)(Differential(x)(u(x))) ~ -π^2 * sin(π * x)
    bcs = [u(0.0) ~ 0.0, u(1.0) ~ 0.0]
    @named sys = PDESystem(eq, bcs, [x ∈ Interval(0.0, 1.0)], [x], [u(x)])
    return sys, mlp(1, 32, 3), GridTraining(1 / 63), false
end
function cfg2()
    @parameters x y
    @variables u(..)
    Dxx, Dyy = Differential(x)^2, Differential(y)^2
    eq = Dxx(u(x, y)) + Dyy(u(x, y)) ~ -sin(π * x) * sin(π * y)
    bcs = [u(0, y) ~ 0.0, u(1, y) ~ 0.0, u(x, 0) ~ 0.0, u(x, 1) ~ 0.0]
    @named sys = PDESystem(eq, bcs, [x ∈ Interval(0.0, 1.0), y ∈ Interval(0.0, 1.0)], [x, y], [u(x, y)])
    return sys, mlp(2, 64, 4), QuasiRandomTraining(256; sampling_alg = SobolSample(), resampling = false, minibatch = 1), false
end
function cfg3()
    @parameters t x
    @variables u(..)
    Dt, Dx, Dxx = Differential(t), Differential(x), Differential(x)^2
    eq = Dt(u(t, x)) + u(t, x) * Dx(u(t, x)) - (0.01 / π) * Dxx(u(t, x)) ~ 0
    bcs = [u(0, x) ~ -sin(π * x), u(t, -1) ~ 0.0, u(t, 1) ~ 0.0]
    @named sys = PDESystem(eq, bcs, [t ∈ Interval(0.0, 1.0), x ∈ Interval(-1.0, 1.0)], [t, x], [u(t, x)])
    return sys, mlp(2, 64, 4), QuasiRandomTraining(256; sampling_alg = SobolSample(), resampling = false, minibatch = 1), false
end
function cfg4()
    @parameters x y
    @variables u(..) v(..) p(..)
    Dx, Dy = Differential(x), Differential(y)
    Dxx, Dyy = Dx^2, Dy^2
    ν = 0.01
    eqs = [u(x, y) * Dx(u(x, y)) + v(x, y) * Dy(u(x, y)) + Dx(p(x, y)) - ν * (Dxx(u(x, y)) + Dyy(u(x, y))) ~ 0,
           u(x, y) * Dx(v(x, y)) + v(x, y) * Dy(v(x, y)) + Dy(p(x, y)) - ν * (Dxx(v(x, y)) + Dyy(v(x, y))) ~ 0,
           Dx(u(x, y)) + Dy(v(x, y)) ~ 0]
    bcs = [u(0, y) ~ 0.0, u(1, y) ~ 0.0, u(x, 0) ~ 0.0, u(x, 1) ~ 1.0, v(0, y) ~ 0.0, v(1, y) ~ 0.0, v(x, 0) ~ 0.0, v(x, 1) ~ 0.0]
    @named sys = PDESystem(eqs, bcs, [x ∈ Interval(0.0, 1.0), y ∈ Interval(0.0, 1.0)], [x, y], [u(x, y), v(x, y), p(x, y)])
    return sys, [mlp(2, 128, 5) for _ in 1:3], QuasiRandomTraining(128; sampling_alg = SobolSample(), resampling = false, minibatch = 1), false
end
function cfg5()
    @parameters t x y z kappa
    @variables u(..)
    Dt = Differential(t)
    Dxx, Dyy, Dzz = Differential(x)^2, Differential(y)^2, Differential(z)^2
    U = u(t, x, y, z)
    eq = Dt(U) ~ kappa * (Dxx(U) + Dyy(U) + Dzz(U))
    bcs = [u(0, x, y, z) ~ sin(π * x) * sin(π * y) * sin(π * z), u(t, 0, y, z) ~ 0.0, u(t, 1, y, z) ~ 0.0, u(t, x, 0, z) ~ 0.0,
           u(t, x, 1, z) ~ 0.0, u(t, x, y, 0) ~ 0.0, u(t, x, y, 1) ~ 0.0]
    doms = [v ∈ Interval(0.0, 1.0) for v in (t, x, y, z)]
    @named sys = PDESystem(eq, bcs, doms, [t, x, y, z], [U], [kappa]; defaults = Dict(kappa => 1.0))
    return sys, mlp(4, 128, 6), StochasticTraining(256; bcs_points = 64), true
end

structural(desc) = [l for l in split(desc, '\n') if !(startswith(l, "lhs ") || startswith(l, "rhs ") || startswith(l, "hint ") || isempty(l))]

@testset "NeuralPDEHIP against the reference objects" begin
    for (name, make) in (("cfg1", cfg1), ("cfg2", cfg2), ("cfg3", cfg3), ("cfg4", cfg4), ("cfg5", cfg5))
        @testset "$name" begin
            Random.seed!(1000)
            sys, chain, strategy, param_estim = make()
            disc = PhysicsInformedNN(chain, strategy; param_estim = param_estim)
            ref = symbolic_discretize(sys, disc)
            desc = descriptor(ref)
            golden = read(joinpath(GOLDEN, name * ".pinnir2"), String)
            @test structural(desc) == structural(golden)                       # θ layout, nets, parameters, coordinate order: string-equal
            st = NeuralPDEHIP.build_state(ref, strategy)                       # creates the engine + verify_layout (θ layout against the Lux chains)
            eg = HIPEngine(golden)
            flat = collect(Float64, ComponentArrays.getdata(ref.flat_init_params))
            for (k, s) in enumerate(st.sets)
                NeuralPDEHIP.set_points!(eg, k, s)
                r1 = NeuralPDEHIP.residual(st.engine, k, flat, size(s, 2))
                r2 = NeuralPDEHIP.residual(eg, k, flat, size(s, 2))
                @test maximum(abs.(r1 .- r2)) <= 2e-6 * max(1.0, maximum(abs.(r2)))    # s-expressions: equal by evaluation
            end
            @test NeuralPDEHIP.selftest(sys, disc; rtol = 1e-4) <= 1e-4              # against the reference's generated loss functions
            # plug-in point 1: HIPStrategy through discretize, one Zygote gradient of the full loss (per-term rrules)
            hdisc = PhysicsInformedNN(chain, HIPStrategy(strategy); param_estim = param_estim, init_params = ref.flat_init_params)
            prob = discretize(sys, hdisc)
            θ0 = prob.u0
            l0 = prob.f(θ0, nothing)
            g0 = Zygote.gradient(t -> prob.f(t, nothing), θ0)[1]
            @test isfinite(l0) && all(isfinite, ComponentArrays.getdata(g0))
            hst = state_of(symbolic_discretize(sys, hdisc))
            lonly = NeuralPDEHIP.evaluate!(hst, θ0; want_grad = false)[1]
            lfull = NeuralPDEHIP.evaluate!(hst, θ0; want_grad = true)[1]
            strategy isa StochasticTraining || @test lonly ≈ lfull rtol = 1e-12   # loss-only evaluation = the fused evaluation's losses
            # plug-in point 2: the fast path, a few Adam iterations must lower the objective
            fprob = hip_discretize(sys, PhysicsInformedNN(chain, strategy; param_estim = param_estim, init_params = ref.flat_init_params))
            res = solve(fprob, OptimizationOptimisers.Adam(1e-3); maxiters = 20)
            @test res.objective < fprob.f(fprob.u0, nothing) || strategy isa StochasticTraining
        end
    end
end

# the reference's GPU test with a Boltz PeriodicEmbedding in front of the Dense stack
# (test/CUDA/nnpde_cuda__1d_pde_dirichlet_bc_cuda.jl:26-48); runs when Boltz is installed
if Base.find_package("Boltz") !== nothing
    @eval import Boltz.Layers: PeriodicEmbedding
    @testset "PeriodicEmbedding in front of the chain" begin
        Random.seed!(100)
        @parameters t x
        @variables u(..)
        eq = Differential(t)(u(t, x)) ~ (Differential(x)^2)(u(t, x))
        bcs = [u(0, x) ~ cos(x), u(t, 0) ~ exp(-t), u(t, 2π) ~ exp(-t)]
        @named sys = PDESystem(eq, bcs, [t ∈ Interval(0.0, 1.0), x ∈ Interval(0.0, 2π)], [t, x], [u(t, x)])
        inner = 30
        chain = Chain(PeriodicEmbedding([2], [2π]), Dense(3, inner, σ), [Dense(inner, inner, σ) for _ in 1:5]..., Dense(inner, 1))
        strategy = QuasiRandomTraining(256; sampling_alg = SobolSample(), resampling = false, minibatch = 1)
        disc = PhysicsInformedNN(chain, strategy)
        ref = symbolic_discretize(sys, disc)
        @test occursin("embed 0 1 1 ", descriptor(ref))
        NeuralPDEHIP.build_state(ref, strategy)                                   # verify_layout: engine trial function == Lux chain incl. the embedding
        @test NeuralPDEHIP.selftest(sys, disc; rtol = 1e-4) <= 1e-4               # residuals against the reference's generated loss functions
    end
end
