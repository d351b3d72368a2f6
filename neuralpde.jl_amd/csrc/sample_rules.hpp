// sample_rules.hpp — the on-device point samplers (uniform, Latin hypercube, Sobol') and the per-point evaluation of a term's
// coordinate-only source channels: ONE statement of the rules, shared by the stand-alone kernels (aux_kernels.hpp: k_sample*, k_src) and
// the persistent training kernel (pinn_train.hpp), which redraws the point sets inside the launch.  Free of host headers (the kernel
// translation units and the hiprtc back end include it).
#pragma once
#include "aux_limits.hpp"
#include "rprog.hpp"

#if defined(PINN_EMU)
#define AUX_DEV inline
#else
#define AUX_DEV __device__ __forceinline__
#endif

namespace aux {

// k_src: coordinate-only subexpressions of a residual (source terms f(x), boundary data g(x), variable coefficients ...),
// evaluated once per installed / redrawn point set, one thread per point, into channel arrays src[j][N] that the fused
// kernel's tape reads as input rows.  The reference re-evaluates them inside the generated loss function on every call
// (they are part of the broadcast expression, src/symbolic_utilities.jl:360-370); values are identical.
struct SrcArgs {
    const float* pts;                     // d x N point-major
    int N, d;
    const rp::Instr* prog;                // compact numbering: rows [0,d) coordinates, row d+q = op q
    int nops, nsrc;
    int root[SRC_MAX];                    // compact row of source j
    float* out;                           // [nsrc][N]
    const float* data;                    // [ndata][N] user-supplied per-point channels (OP_DATA), nullable
};
AUX_DEV void src_point(int p, const SrcArgs& a) {
    float v[EXPR_MAX_ROWS];
    for (int i = 0; i < a.d; ++i) v[i] = a.pts[(size_t)p * a.d + i];
    for (int q = 0; q < a.nops; ++q) {
        const rp::Instr ins = a.prog[q];
        v[a.d + q] = (ins.code == rp::OP_DATA) ? a.data[(size_t)(int)ins.imm * a.N + p] : rp::apply<float>(ins.code, v[ins.a], v[ins.b], ins.imm);
    }
    for (int j = 0; j < a.nsrc; ++j) a.out[(size_t)j * a.N + p] = v[a.root[j]];
}

// counter-based uniform sampler (StochasticTraining's rand(T, d, N) .* (ub .- lb) .+ lb, src/training_strategies.jl:242-245):
// value = hash(seed, draw counter, element index) -> [0, 1)
AUX_DEV unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
AUX_DEV void sample_body(int e, float* pts, int d, const float* lb, const float* ub, unsigned seed, unsigned draw) {
    const int i = e % d;
    unsigned h = mix32((unsigned)e * 0x9E3779B9U + seed);
    h = mix32(h ^ (draw * 0x85EBCA6BU + 0xC2B2AE35U));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    pts[e] = lb[i] + (ub[i] - lb[i]) * u;
}

// Latin-hypercube redraw (the reference's default QuasiRandomTraining sampler, LatinHypercubeSample, src/training_strategies.jl:321,
// 375-381): along every axis the n points occupy the n strata [k/n, (k+1)/n) exactly once, at a random position inside the
// stratum.  The per-axis random permutation of the strata is a keyed 4-round Feistel network on ceil(log2 n) bits with
// cycle walking (a bijection of [0, n) that needs no sort and no memory), keyed by (seed, draw counter, axis).
AUX_DEV unsigned lhs_perm(unsigned k, unsigned n, unsigned key) {
    unsigned bits = 1;
    while ((1u << bits) < n) ++bits;
    if (bits & 1) ++bits;                               // even split into two halves
    const unsigned half = bits >> 1, mask = (1u << half) - 1u;
    unsigned x = k;
    do {
        unsigned l = x >> half, r = x & mask;
        for (unsigned round = 0; round < 4; ++round) {
            const unsigned f = mix32(r ^ (key + round * 0x9E3779B9U)) & mask;
            const unsigned nl = r, nr = l ^ f;
            l = nl; r = nr;
        }
        x = (l << half) | r;
    } while (x >= n);                                   // cycle walking: stay inside [0, n)
    return x;
}
AUX_DEV void sample_lhs_body(int e, float* pts, int d, int n, const float* lb, const float* ub, unsigned seed, unsigned draw) {
    const int i = e % d, p = e / d;
    const unsigned key = mix32(seed ^ (draw * 0x85EBCA6BU + 0xC2B2AE35U) ^ ((unsigned)i * 0x27D4EB2FU));
    const unsigned stratum = lhs_perm((unsigned)p, (unsigned)n, key);
    unsigned h = mix32((unsigned)e * 0x9E3779B9U + seed);
    h = mix32(h ^ (draw * 0xC2B2AE3DU + 0x165667B1U));
    // stratum and in-stratum position are combined as integers (24 significant bits in total) so that the float sum can never
    // round up into the next stratum
    unsigned bits = 1;
    while ((1u << bits) < (unsigned)n) ++bits;
    const unsigned k = bits < 24u ? 24u - bits : 0u;
    const unsigned fixed = (stratum << k) | (k ? (h >> (32u - k)) : 0u);
    const float u = (float)fixed / ((float)n * (float)(1u << k));
    pts[e] = lb[i] + (ub[i] - lb[i]) * u;
}

// Sobol' redraw (QuasiRandomTraining(points; sampling_alg = SobolSample()), [3P] QuasiMonteCarlo.jl / Sobol.jl, used by the reference's
// deterministic tests, e.g. test/NNPDE1/nnpde__pde_vi_pde_with_mixed_derivative.jl:78-80).  Point p of the design is element p + 1
// of the Gray-code (Antonov-Saleev) Sobol' sequence — the all-zero first element is skipped as Sobol.jl does — with the
// Joe-Kuo direction numbers ("new-joe-kuo-6", the table Sobol.jl and scipy.stats.qmc share) of axes 1..8, regenerated from the
// primitive polynomial (degree s, coefficients a) and initial values m by the standard recurrence, so no table lives in memory.
// seed = 0: the plain sequence on every draw = what the reference's un-randomised SobolSample returns on every call; seed != 0:
// every draw applies a fresh per-axis digital shift keyed by (seed, draw counter) — a randomisation that keeps the net property.
AUX_DEV unsigned sobol_bits(unsigned index, int axis) {
    // degree | coefficient bits | up to five initial m values, 4 bits each (values 1..17 need 5 bits for the last one: kept apart)
    const int S[8] = {0, 1, 2, 3, 3, 4, 4, 5};
    const int A[8] = {0, 0, 1, 1, 2, 1, 4, 2};
    const int M[8][5] = {{0, 0, 0, 0, 0}, {1, 0, 0, 0, 0}, {1, 3, 0, 0, 0}, {1, 3, 1, 0, 0}, {1, 1, 1, 0, 0}, {1, 1, 3, 3, 0}, {1, 3, 5, 13, 0}, {1, 1, 5, 5, 17}};
    const unsigned gray = index ^ (index >> 1);
    const int s = S[axis], a = A[axis];
    unsigned m[5] = {0, 0, 0, 0, 0};                    // sliding window m[j-1] .. m[j-s] (newest first)
    unsigned x = 0;
    for (int j = 1; j <= 32 && (gray >> (j - 1)) != 0u; ++j) {
        unsigned mj;
        if (axis == 0) mj = 1u;
        else if (j <= s) mj = (unsigned)M[axis][j - 1];
        else {
            mj = m[s - 1] ^ (m[s - 1] << s);
            for (int k = 1; k < s; ++k)
                if ((a >> (s - 1 - k)) & 1) mj ^= m[k - 1] << k;
        }
        for (int k = 4; k > 0; --k) m[k] = m[k - 1];
        m[0] = mj;
        if ((gray >> (j - 1)) & 1u) x ^= mj << (32 - j);
    }
    return x;
}
AUX_DEV void sample_sobol_body(int e, float* pts, int d, const float* lb, const float* ub, unsigned seed, unsigned draw) {
    const int i = e % d, p = e / d;
    unsigned x = sobol_bits((unsigned)p + 1u, i);
    if (seed != 0u) x ^= mix32(seed ^ (draw * 0x85EBCA6BU + 0xC2B2AE35U) ^ ((unsigned)i * 0x27D4EB2FU));
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);        // exact for the plain sequence while n < 2^24
    pts[e] = lb[i] + (ub[i] - lb[i]) * u;
}


// A term whose point set is REDRAWN before every evaluation (StochasticTraining, QuasiRandomTraining(resampling = true):
// src/training_strategies.jl:242-245, 375-381), as one record of a device-side table: ONE launch (k_resample, or the update phase of the
// persistent training kernel) redraws every such term of a problem and re-evaluates its coordinate-only source channels, instead of a
// sampler launch + a source launch per term and step.
struct ResampleTerm {
    float* pts;                  // [n][d]
    int n, d, kind;              // 1 uniform, 2 Latin hypercube, 3 Sobol'
    const float* lb; const float* ub;
    unsigned seed, draw0;        // draw counter of step 0 of the table's lifetime
    int has_src;
    SrcArgs src;
};
// thread gid of nthreads: points gid, gid + nthreads, ... of every term, at draw counter draw0 + step
AUX_DEV void resample_point_sets(const ResampleTerm* samp, int nsamp, int gid, int nthreads, int step) {
    for (int t = 0; t < nsamp; ++t) {
        const ResampleTerm& S = samp[t];
        const unsigned draw = S.draw0 + (unsigned)step;
        for (int p = gid; p < S.n; p += nthreads) {
            for (int i = 0; i < S.d; ++i) {
                const int e = p * S.d + i;
                if (S.kind == 3) sample_sobol_body(e, S.pts, S.d, S.lb, S.ub, S.seed, draw);
                else if (S.kind == 2) sample_lhs_body(e, S.pts, S.d, S.n, S.lb, S.ub, S.seed, draw);
                else sample_body(e, S.pts, S.d, S.lb, S.ub, S.seed, draw);
            }
            if (S.has_src) src_point(p, S.src);               // (reads the coordinates this thread has just written)
        }
    }
}

}  // namespace aux
