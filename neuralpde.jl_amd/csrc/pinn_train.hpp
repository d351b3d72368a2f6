// pinn_train.hpp — K optimiser iterations of a SMALL problem in ONE launch (family 1: one wave per point tile).
//
// The reference's own regime — 12..32-wide networks on 100..1,000 collocation points, trained for thousands of Adam iterations
// (test/NNPDE1/nnpde__pde_ii_2d_poisson.jl:83-85: solve(prob, Adam(0.1); maxiters = 4000) over full_loss_function,
// src/discretize.jl:567-598) — is launch-bound on a GPU: the stand-alone loop (engine.cpp: adam_loop) pays three dependent launches per
// iteration (residual kernel -> fixed-order reduction -> Adam + weight-image scatter), each a few microseconds of work behind a kernel
// boundary, a grid ramp and first-touch cache misses.  Here the whole iteration is one pass of a persistent kernel whose workgroups are
// all resident (at most REDUCE_DIRECT_MAX = 32 workgroups on 256 CUs):
//
//     (once)  every thread takes ONE element of [theta | K sums] (TrainArgs::own_r: placed in slab-entry order) and keeps its reduction map,
//             its image positions and its (theta, m, v) in registers across the steps
//     for step = 1 .. K:
//         wave_main<MODE_FUSED>         forward jets + residual + reverse sweep of this wave's tiles -> gradient slabs, loss partials
//         grid barrier A                (write-through exchange, no fences: vec.hpp grid_barrier_wt; fenced when point sets are redrawn)
//         update                        the fixed-order sum of the element's slab entries over the workgroups (the association of
//                                       aux::reduce_direct_body: every load in flight at once), the Adam rule (update_rules.hpp), the new
//                                       value scattered into the packed weight image; K threads: the per-term sums of squares; one idle
//                                       thread: the loss history of the PREVIOUS step; every thread: its share of the NEXT step's redrawn
//                                       point sets (sample_rules.hpp) where the strategy resamples
//         grid barrier B
//
// Same arithmetic, same association, same rounding as the three stand-alone kernels: K iterations in one launch equal K single steps
// BIT FOR BIT (tests/test_train_kernel.py; GPU mirror in tests/test_gpu_mirror.py).  With TrainArgs::eval_only the same kernel is ONE
// evaluation (residual kernel, barrier A, sums -> out) for the host entry points.  Not eligible (the engine takes the loop): more
// workgroups than the one-stage reduction covers, fewer threads than parameters, estimated PDE parameters (training), several networks or
// launch groups, communicators, sin / per-layer activations, the float64 mode, embedded or data-carrying redrawn sets.  DESIGN.md 4.6.
#pragma once
#include "pinn_kernels.hpp"
#include "update_rules.hpp"
#include "sample_rules.hpp"

namespace pk {

using TrainSampler = aux::ResampleTerm;      // a term whose point set is redrawn before every evaluation (sample_rules.hpp)

struct TrainArgs {
    // reduction inputs: the launch's gradient slabs and per-wave loss partials, the theta -> slab-entry map of the (single) launch group
    const float* slabs;          // [nblocks][slab_floats]
    const double* losspart;      // [nblocks * 4][K]
    const int* row_ptr;          // [P + 1]
    const int* row_ent;          // slab entry of every contribution
    int slab_floats, nblocks;
    float* out;                  // [P gradient | K raw per-term sums of squares] of the last step
    double* lossraw;             // [K] the same sums in double
    // Adam state, resident
    float* theta; float* m; float* v;
    float lr, b1, b2, eps;
    const float* c12;            // [2 * nsteps]: 1 / (1 - b1^t), 1 / (1 - b2^t) of every step of this launch
    const int* inv_ptr;          // theta element -> positions in the packed weight image
    const int* inv_pos;
    float* packed;               // the image the residual kernel reads (GroupArgs::packed)
    double* hist;                // [nsteps] total weighted loss of every step's evaluation
    const float* w_over_n;       // [K]
    float* sums2;                // [2][K]: the K sums of the even / odd steps (the history entry of a step is written one update later)
    int P, K, nsteps;
    const int* own_r;            // cached form: [nblocks * 256] the element of [0, P + K) thread gid owns, -1: none.  The engine places theta element r
                                 // at gid = its first slab entry, so that the lanes of a wave read CONSECUTIVE slab entries (the slabs are in MFMA
                                 // fragment order, not theta order: with gid = r a wave's 64 lanes touch 64 cache lines per load — measured 8.9 us per
                                 // update phase on MI355X, most of it the L1's line-by-line service of 128 such loads per lane)
    const TrainSampler* samp;    // [nsamp] redrawn terms (device memory), nullptr: fixed point sets
    int nsamp;
    int fenced;                  // 1: barriers with release / acquire fences (redrawn point sets are read through the L1 by the evaluation)
    int eval_only;               // 1: ONE evaluation — residual kernel, barrier, fixed-order sums into out[] — and nothing else (pinn_loss_grad on a small
                                 // problem: one launch instead of the residual kernel + the reduction kernel); theta / m / v / hist / c12 unused
    unsigned* hflag;             // host-mapped word: set to 1 by the launch when its barrier timed out (the host reads it after its next synchronisation)
    unsigned arrivals0;          // value of the barrier's arrival counter when this launch starts (the counter is never reset between launches)
    int hist_gid;                // the thread that writes the loss history (one without an element where the grid has one)
    int cached;                  // every thread of the grid owns at most ONE element of [0, P + K) with at most TRAIN_MAX_CONTRIB slab entries and
                                 // TRAIN_MAX_POS image positions: its maps and its (theta, m, v) stay in registers across the steps
    unsigned* bar;               // [0] arrival counter of the grid barrier (zeroed once, then running on: arrivals0), [1] time-out flag
};

constexpr int TRAIN_MAX_CONTRIB = 4, TRAIN_MAX_POS = 4;

// PINN_TRAIN_WT (default 1): what crosses workgroups inside the launch — gradient slabs, loss partials, the weight image, the K sums — is
// stored WRITE-THROUGH (sc1) and read with agent-scope loads, so the two grid barriers of an iteration need no release / acquire fence
// (vec.hpp: grid_barrier_wt; cdna_hip_programming.md Guideline 16, forms R1 / R2).  0: plain stores and loads between fenced barriers.
#ifndef PINN_TRAIN_WT
#define PINN_TRAIN_WT 1
#endif
constexpr bool TRAIN_WT = PINN_TRAIN_WT != 0;
template <class T> DEV T train_ld(const T* p) { return TRAIN_WT ? uload_wt(p) : *p; }
template <class T> DEV void train_st(T* p, T x) { if (TRAIN_WT) ustore_wt(p, x); else *p = x; }
DEV void train_barrier(unsigned* bar, unsigned target, int fenced) { if (TRAIN_WT && !fenced) grid_barrier_wt(bar, target); else grid_barrier(bar, target); }

// profiling build only (-DPINN_STAMP variant, tools/time_adam_loop.py --lib <it>): the history thread of the launch accumulates the s_memtime ticks of the four
// phases of an iteration — evaluation, barrier, update, barrier — into bar[4 .. 11] (64-bit sums); never defined for the product
#if defined(PINN_STAMP) && !defined(PINN_EMU)
#define TRAIN_STAMP_DECL unsigned long long tst_acc[4] = {0, 0, 0, 0}, tst_last = __builtin_amdgcn_s_memtime();
#define TRAIN_STAMP(i) { const unsigned long long tst_now = __builtin_amdgcn_s_memtime(); tst_acc[i] += tst_now - tst_last; tst_last = tst_now; }
#define TRAIN_STAMP_STORE if (first) { for (int i_ = 0; i_ < 4; ++i_) reinterpret_cast<unsigned long long*>(ta.bar + 4)[i_] = tst_acc[i_]; }
#else
#define TRAIN_STAMP_DECL
#define TRAIN_STAMP(i)
#define TRAIN_STAMP_STORE
#endif

// the K raw sums of squares of a step: column k of the per-wave loss partials in wave order, one double accumulator (aux::reduce_direct_body)
DEV void train_sum_elem(int k, const TrainArgs& a, int step) {
    const double* p = a.losspart + k;
    const int nw = a.nblocks * 4;
    double s = 0.0;
#ifdef PINN_EMU
    for (int wv_ = 0; wv_ < nw; ++wv_) s += train_ld(p + (size_t)wv_ * a.K);
#else
    // Every load of a chunk in flight before the first add, through a buffer view: ONE per-lane offset register (the column), the row in
    // the scalar offset.  Every load is UNCONDITIONAL (rows past the end re-read row 0) and only the add is selected: a load under a
    // lane-dependent condition becomes a branch with the wait for its result at the join, i.e. one full L2 round trip per load; and with
    // plain pointers the compiler precomputes one 64-bit address per load in front of the step loop and spills them (measured on MI355X:
    // 23-35 us per update phase either way, 836 B of scratch per lane)
    const ubuf LPB = ub_make(reinterpret_cast<const float*>(a.losspart), (size_t)nw * a.K * 2);
    int kcol = k, K_ = a.K, nw_ = nw;
    opaque_v(kcol); opaque_s(K_); opaque_s(nw_);
    constexpr int CH = 72;                                    // (one chunk = one round trip for launches of up to 18 workgroups)
    for (int w0 = 0; w0 < nw_; w0 += CH) {
        double q[CH];
        PINN_UNROLL for (int b = 0; b < CH; ++b) q[b] = ub_loadd<TRAIN_WT ? 16 : 0>(LPB, ((w0 + b < nw_) ? w0 + b : 0) * K_, kcol);
        PINN_UNROLL for (int b = 0; b < CH; ++b) s += (w0 + b < nw_) ? q[b] : 0.0;      // (+0.0 is exact here: see train_own_step)
    }
#endif
    a.out[a.P + k] = (float)s;
    train_st(a.sums2 + (step & 1) * a.K + k, (float)s);
    if (a.lossraw) a.lossraw[k] = s;
}
// total weighted loss of step `step` (aux::total_loss_body) from the sums its update left in sums2
DEV void train_hist(const TrainArgs& a, int step) {
    double s = 0.0;
    for (int k = 0; k < a.K; ++k) s += (double)train_ld(a.sums2 + (step & 1) * a.K + k) * (double)a.w_over_n[k];
    a.hist[step] = s;
}

// element r of [0, P + K) after a step's evaluation (general form: maps and state read from memory every step)
DEV void train_update_elem(int r, const TrainArgs& a, int step) {
    if (r < a.P) {
        // the association of aux::reduce_direct_body: contributions in map order, workgroups in launch order, one double accumulator
        double s = 0.0;
        const int i0 = a.row_ptr[r], i1 = a.row_ptr[r + 1];
        for (int i = i0; i < i1; ++i) {
            const float* p = a.slabs + a.row_ent[i];
            for (int b = 0; b < a.nblocks; ++b) s += (double)train_ld(p + (size_t)b * a.slab_floats);
        }
        const float g = (float)s;
        a.out[r] = g;
        if (a.eval_only) return;
        float mi = a.m[r], vi = a.v[r];
        const float t = ur::adam_update(a.theta[r], mi, vi, g, a.lr, a.b1, a.b2, a.eps, a.c12[2 * step], a.c12[2 * step + 1]);
        a.m[r] = mi;
        a.v[r] = vi;
        a.theta[r] = t;
        for (int q = a.inv_ptr[r]; q < a.inv_ptr[r + 1]; ++q) train_st(a.packed + (a.inv_pos[q] & 0xFFFFFF), t);
    } else if (r < a.P + a.K) train_sum_elem(r - a.P, a, step);
}

// cached form (TrainArgs::cached): what a thread keeps across the steps for the one element it owns.  Per step only the slab entries are
// read — every load of the step in flight at once, ONE L2 round trip — instead of three dependent index / state round trips first
struct TrainOwn {
    int r, n, npos;
    int ent[TRAIN_MAX_CONTRIB], pos[TRAIN_MAX_POS];
    float th, m, v;
};
DEV void train_own_init(TrainOwn& o, int r, const TrainArgs& a) {
    o.r = r < 0 ? a.P + a.K : r;                              // (no element: past the end)
    r = o.r; o.n = 0; o.npos = 0; o.th = o.m = o.v = 0.f;
    PINN_UNROLL for (int j = 0; j < TRAIN_MAX_CONTRIB; ++j) o.ent[j] = 0;
    PINN_UNROLL for (int j = 0; j < TRAIN_MAX_POS; ++j) o.pos[j] = 0;
    if (r < a.P) {
        const int i0 = a.row_ptr[r];
        o.n = a.row_ptr[r + 1] - i0;
        PINN_UNROLL for (int j = 0; j < TRAIN_MAX_CONTRIB; ++j) if (j < o.n) o.ent[j] = a.row_ent[i0 + j];
        if (!a.eval_only) {                                   // (an evaluation-only launch has no optimiser state and no weight image to update)
            const int q0 = a.inv_ptr[r];
            o.npos = a.inv_ptr[r + 1] - q0;
            PINN_UNROLL for (int j = 0; j < TRAIN_MAX_POS; ++j) if (j < o.npos) o.pos[j] = a.inv_pos[q0 + j] & 0xFFFFFF;
            o.th = a.theta[r]; o.m = a.m[r]; o.v = a.v[r];
        }
    }
}
DEV void train_own_step(TrainOwn& o, const TrainArgs& a, int step) {
    const int r = o.r;
    if (r < a.P) {
        double s = 0.0;
#ifdef PINN_EMU
        for (int j = 0; j < o.n; ++j)
            for (int b = 0; b < a.nblocks; ++b) s += (double)train_ld(a.slabs + (size_t)o.ent[j] + (size_t)b * a.slab_floats);
#else
        // (see train_sum_elem) buffer view over the slabs: the slab entry in the per-lane offset, the workgroup in the scalar offset; the
        // launch has at most 32 workgroups
        const ubuf SLB = ub_make(a.slabs, (size_t)a.nblocks * a.slab_floats);
        int ent[TRAIN_MAX_CONTRIB], n_ = o.n, nb_ = a.nblocks, stride_ = a.slab_floats;
        PINN_UNROLL for (int j = 0; j < TRAIN_MAX_CONTRIB; ++j) { ent[j] = o.ent[j]; opaque_v(ent[j]); }
        opaque_v(n_); opaque_s(nb_); opaque_s(stride_);
        float q[TRAIN_MAX_CONTRIB][32];
        PINN_UNROLL for (int j = 0; j < TRAIN_MAX_CONTRIB; ++j)
            PINN_UNROLL for (int b = 0; b < 32; ++b) q[j][b] = ub_loadf<TRAIN_WT ? 16 : 0>(SLB, ((b < nb_) ? b : 0) * stride_, ent[j]);
        // unused slots add +0.0, which is exact: s starts at +0.0 and a sum that started there is never -0.0 — so the accumulator chain is
        // one v_add_f64 per slot, no select on it, and the sum equals the stand-alone kernel's bit for bit
        // (no early exit from these loops: with one, they are not flattened and q[][] lives in scratch memory — measured)
        PINN_UNROLL for (int j = 0; j < TRAIN_MAX_CONTRIB; ++j)
            PINN_UNROLL for (int b = 0; b < 32; ++b) s += (double)((j < n_ && b < nb_) ? q[j][b] : 0.f);
#endif
        const float g = (float)s;
        a.out[r] = g;
        if (a.eval_only) return;
        const float t = ur::adam_update(o.th, o.m, o.v, g, a.lr, a.b1, a.b2, a.eps, a.c12[2 * step], a.c12[2 * step + 1]);
        o.th = t;
        a.m[r] = o.m;
        a.v[r] = o.v;
        a.theta[r] = t;
        PINN_UNROLL for (int j = 0; j < TRAIN_MAX_POS; ++j) if (j < o.npos) train_st(a.packed + o.pos[j], t);
    } else if (r < a.P + a.K) train_sum_elem(r - a.P, a, step);
}

// thread gid's share of the NEXT step's point sets: points gid, gid + nthreads, ... of every redrawn term (drawn during the update phase of
// the step before: same counter-based rules as the stand-alone kernels, one thread per POINT — its coordinates, then its source channels)
DEV void train_resample(const TrainArgs& a, int gid, int nthreads, int next_step) { aux::resample_point_sets(a.samp, a.nsamp, gid, nthreads, next_step); }

// one thread reports a barrier time-out of this launch to the host
DEV void train_report(const TrainArgs& ta, int blk, int w) {
#ifndef PINN_EMU
    if (blk == 0 && threadIdx.x == 0 && __hip_atomic_load(ta.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) *ta.hflag = 1u;
#endif
}

// the wave program of the training kernel: workgroup `blk` of `nblocks`, wave `w` of the workgroup
template <class S, int ACTK>
DEV void wave_train(const GroupArgs& ga, const TrainArgs& ta, int blk, int nblocks, int w, float* lds) {
    unsigned arrivals = ta.arrivals0;
#ifdef PINN_EMU
    const int gid0 = (blk * 4 + w) * 64;                      // first of this wave's 64 threads
    TrainOwn own[64];
    if (ta.cached) for (int l = 0; l < 64; ++l) train_own_init(own[l], ta.own_r[gid0 + l], ta);
    const bool first = (ta.hist_gid >= gid0 && ta.hist_gid < gid0 + 64);
#else
    const int gid0 = blk * 256 + (int)threadIdx.x;
    TrainOwn own;
    if (ta.cached) train_own_init(own, ta.own_r[gid0], ta);
    const bool first = (gid0 == ta.hist_gid);
#endif
    TRAIN_STAMP_DECL
    for (int step = 0; step < ta.nsteps; ++step) {
        wave_main<S, MODE_FUSED, ACTK, TRAIN_WT>(ga, blk, nblocks, w, lds);
        TRAIN_STAMP(0)
        arrivals += (unsigned)nblocks;
        train_barrier(ta.bar, arrivals, ta.fenced);           // every workgroup's slabs and loss partials are visible
        TRAIN_STAMP(1)
        if (first && step > 0) train_hist(ta, step - 1);      // the previous step's loss: its K sums were written one barrier ago
#ifdef PINN_EMU
        for (int l = 0; l < 64; ++l) {
            if (ta.cached) train_own_step(own[l], ta, step);
            else for (int r = gid0 + l; r < ta.P + ta.K; r += nblocks * 256) train_update_elem(r, ta, step);
            if (ta.nsamp > 0 && step + 1 < ta.nsteps) train_resample(ta, gid0 + l, nblocks * 256, step + 1);
        }
#else
        if (ta.cached) train_own_step(own, ta, step);
        else for (int r = gid0; r < ta.P + ta.K; r += nblocks * 256) train_update_elem(r, ta, step);
        if (ta.nsamp > 0 && step + 1 < ta.nsteps) train_resample(ta, gid0, nblocks * 256, step + 1);
#endif
        TRAIN_STAMP(2)
        if (ta.eval_only) { train_report(ta, blk, w); return; }
        arrivals += (unsigned)nblocks;
        train_barrier(ta.bar, arrivals, ta.fenced);           // the new parameters (theta, weight image) and the K sums are visible
        TRAIN_STAMP(3)
    }
    if (first && ta.nsteps > 0) train_hist(ta, ta.nsteps - 1);
    train_report(ta, blk, w);
    TRAIN_STAMP_STORE
}

}  // namespace pk
