// aux_kernels.hpp — small HBM-side kernels around the fused residual kernel.
//   k_pack    : theta (ComponentArrays order, src/discretize.jl:451-465) -> padded MFMA-fragment order
//   k_params  : theta.p / default_p -> parameter rows of the residual tape (src/discretize.jl:83-109)
//   k_reduce1 : stage 1 of the fixed-order reduction: every gradient-slab offset (dense, coalesced) and loss column of
//               every launch group summed over one contiguous chunk of workgroups
//   k_reduce2 : stage 2: theta element p = sum over the groups / slab entries / chunks that feed it, written straight
//               into the output vector [P floats grad | K floats raw sums of squares] (+ K doubles for the host path)
// Every sum has a fixed order => bit-identical results run to run.
#pragma once
#include <cstdlib>
#include "aux_limits.hpp"
#include "plat.hpp"
#include "rprog.hpp"
#include "update_rules.hpp"
#include "sample_rules.hpp"

namespace aux {

#ifdef PINN_EMU
#define AUX_DEV inline
#else
#define AUX_DEV __device__ __forceinline__
#endif

struct alignas(16) F4 { float x, y, z, w; };

struct Reduce1Args {
    double* tmp[MAX_GROUPS];            // [nsplit][nent + K]
    const float* slabs[MAX_GROUPS];     // [nblocks][slab]
    const double* losspart[MAX_GROUPS]; // [nblocks*4][K]
    int slab[MAX_GROUPS], nblocks[MAX_GROUPS], nsplit[MAX_GROUPS], nent[MAX_GROUPS], active[MAX_GROUPS];
    int nwpb[MAX_GROUPS];               // waves per workgroup (rows of losspart per block)
    int K;
};
struct Reduce2Args {
    float* out;                         // [P] gradient (not touched when skip_grad)
    float* sums;                        // [K] per-term sums of squares (normally out + P; a buffer of its own for loss-only callers)
    double* lossraw;                    // [K] (nullable)
    const int* row_ptr;                 // [P + 1]
    const int* row_grp;                 // group of every contribution
    const int* row_ent;                 // slab-entry index (within its group) of every contribution
    const double* tmp[MAX_GROUPS];
    int stride[MAX_GROUPS], nsplit[MAX_GROUPS], nent[MAX_GROUPS], active[MAX_GROUPS];
    int ent_active[MAX_GROUPS];         // 0: this group's gradient went into another group's slabs (chained launch groups)
    int ngroups, P, K;
    int skip_grad;                      // loss-only evaluation: only the K sums are produced, out[0, P) is left alone
    const int* perm;                    // one-stage form (k_reduce_direct): thread t sums element perm[t] of [theta | K sums] (nullable: t itself).
                                        // The engine orders the elements by their first slab entry, so that the lanes of a wave read CONSECUTIVE
                                        // slab entries — the slabs are in MFMA-fragment order, in theta order a wave touches 64 cache lines per load
};
// ONE-kernel reduction for the common case that a single slab set carries the whole gradient (one network whose launch groups are
// merged / chained, family 2 or 3: every theta element has exactly one slab entry): a block sums 32 consecutive slab entries over all
// workgroups — 32 chunks of consecutive slabs in parallel, one 16-byte load per slab and lane, then the chunk sums in order — and
// scatters them through the entry -> theta map; one more block per term sums that term's per-wave loss partials.  Fixed association =>
// bit-identical run to run.  (k_reduce1 + k_reduce2 read the same 26 MB with 300 blocks and pay a second dependent launch:
// 13 + 6 us on the bench workload; this kernel: see profiles/r03_*.)
struct ReduceOneArgs {
    const float* slabs;                 // [nblocks][slab]
    int slab, nblocks, nent;            // nent: floats per slab that take part (multiple of 4)
    const int* ent_theta;               // [nent]: theta element fed by slab entry e, -1: none
    float* out;                         // [P] gradient (nent == 0: not touched)
    float* sums;                        // [K] per-term sums of squares (normally out + P)
    double* lossraw;                    // [K] (nullable)
    int P, K;
    int nloss;                          // launches whose per-wave loss partials are summed, in this order
    const double* losspart[MAX_GROUPS];
    int nrows[MAX_GROUPS];              // rows (waves) of losspart[i]
};
constexpr int RONE_CHUNKS = 32, RONE_ENT = 32;

// k_expr: the residual tape of an equation that couples several networks, one thread per collocation point.
// Inputs: the jet channels every network's FWD launch wrote ([channel][N]); outputs: d(loss)/d(jet) per slot for the
// GRADIN launches, per-wave loss / dL/dp partial sums (fixed order), optionally the residual itself.
struct ExprArgs {
    const float* pts;                     // d x N point-major
    int N, d, nparams, nparams_estim;
    const float* params;
    int nslots;
    const float* jets[EXPR_MAX_SLOTS];    // slot s -> its channel array [N]
    float* ubar[EXPR_MAX_SLOTS];          // slot s -> d(loss)/d(jet) channel array [N] (distinct slots = distinct arrays)
    int nzero;
    float* zero[EXPR_MAX_SLOTS];          // channel arrays of the kernels' jet sets that this equation does not use
    const rp::Instr* prog;
    int nops, out_row;
    float scale;                          // 2 w / N_norm
    double* losspart;                     // [nblocks*4][K]
    float* pslab;                         // [nblocks][4 waves][4 params]
    int K, term_id;
    float* resid;                         // nullable: write r[N] and skip the adjoint
    int loss_only;                        // 1: the sums of squares only (no adjoint, ubar untouched)
    const float* data;                    // [ndata][N] user-supplied per-point channels (OP_DATA), nullable
    const float* pw;                      // per-point factors sqrt(N w_i) of a quadrature-weighted term, nullable
};
// one point; returns r (masked by `valid`), writes ubar, returns dL/dp contributions in pb[]
AUX_DEV float expr_point(int p, const ExprArgs& a, float (&pb)[4]) {
    float v[EXPR_MAX_ROWS], g[EXPR_MAX_ROWS];
    const int R0 = a.d + a.nparams + a.nslots;
    for (int i = 0; i < a.d; ++i) v[i] = a.pts[(size_t)p * a.d + i];
    for (int j = 0; j < a.nparams; ++j) v[a.d + j] = a.params[j];
    for (int s = 0; s < a.nslots; ++s) v[a.d + a.nparams + s] = a.jets[s][p];
    for (int q = 0; q < a.nops; ++q) {
        const rp::Instr ins = a.prog[q];
        const float va = rp::is_nullary(ins.code) ? 0.f : v[ins.a];
        const float vb = rp::is_binary(ins.code) ? v[ins.b] : 0.f;
        v[R0 + q] = (ins.code == rp::OP_DATA) ? a.data[(size_t)(int)ins.imm * a.N + p] : rp::apply<float>(ins.code, va, vb, ins.imm);
    }
    const float r = v[a.out_row];
    for (int j = 0; j < 4; ++j) pb[j] = 0.f;
    if (a.resid) { a.resid[p] = r; return r; }
    if (a.loss_only) return r * (a.pw ? a.pw[p] : 1.0f);
    for (int q = 0; q < R0 + a.nops; ++q) g[q] = 0.f;
    g[a.out_row] = 1.0f;
    for (int q = a.nops - 1; q >= 0; --q) {
        const rp::Instr ins = a.prog[q];
        if (rp::is_nullary(ins.code)) continue;
        const float vb = rp::is_binary(ins.code) ? v[ins.b] : 0.f;
        float da, db;
        rp::adjoint<float>(ins.code, v[ins.a], vb, v[R0 + q], ins.imm, g[R0 + q], da, db);
        g[ins.a] += da;
        if (rp::is_binary(ins.code)) g[ins.b] += db;
    }
    const float sw = a.pw ? a.pw[p] : 1.0f;
    const float rs = r * sw;
    const float rbar = rs * a.scale * sw;
    for (int s = 0; s < a.nslots; ++s) a.ubar[s][p] = rbar * g[a.d + a.nparams + s];
    for (int z = 0; z < a.nzero; ++z) a.zero[z][p] = 0.f;
    for (int j = 0; j < a.nparams_estim; ++j) pb[j] = rbar * g[a.d + j];
    return rs;
}

// ---- resident-theta training loop ----
// the Adam rule itself lives in update_rules.hpp (shared with the persistent training kernel, pinn_train.hpp)
using ur::adam_update;
AUX_DEV void adam_body(int i, float* theta, float* m, float* v, const float* grad, float lr, float b1, float b2, float eps, float c1, float c2) {
    float mi = m[i], vi = v[i];
    const float t = adam_update(theta[i], mi, vi, grad[i], lr, b1, b2, eps, c1, c2);
    m[i] = mi;
    v[i] = vi;
    theta[i] = t;
}
// the resident loop's update kernel: Adam step of element i, its new value scattered into the packed weight images (inverse pack
// map), and — one thread — the evaluation's total weighted loss into the history: one launch instead of total_loss + adam + pack
struct AdamFusedArgs {
    float* theta; float* m; float* v; const float* out;      // out = [P gradient | K raw per-term sums]
    int P, K;
    float lr, b1, b2, eps, c1, c2;
    const int* inv_ptr; const int* inv_pos;
    float* packed[MAX_PACK_NETS];
    double* hist; int step; const float* w_over_n;
};
AUX_DEV void adam_fused_body(int i, const AdamFusedArgs& a) {
    if (i == 0) {
        double s = 0.0;
        for (int k = 0; k < a.K; ++k) s += (double)a.out[a.P + k] * (double)a.w_over_n[k];
        a.hist[a.step] = s;
    }
    if (i >= a.P) return;
    float mi = a.m[i], vi = a.v[i];
    const float t = adam_update(a.theta[i], mi, vi, a.out[i], a.lr, a.b1, a.b2, a.eps, a.c1, a.c2);
    a.m[i] = mi;
    a.v[i] = vi;
    a.theta[i] = t;
    for (int q = a.inv_ptr[i]; q < a.inv_ptr[i + 1]; ++q) {
        const int pos = a.inv_pos[q];
        a.packed[pos >> 24][pos & 0xFFFFFF] = t;
    }
}
// total weighted loss of one evaluation from the raw per-term sums in out[P..P+K)
AUX_DEV void total_loss_body(double* hist, int step, const float* out, int P, int K, const float* w_over_n) {
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)out[P + k] * (double)w_over_n[k];
    hist[step] = s;
}
// sharded evaluations: the K per-term sums of squares cross the ranks as DOUBLES (their own all-reduce, grouped with the gradient's), so
// that N > 1 delivers the same losses as N = 1 to double rounding; this writes them back into the float out vector [P .. P + K)
AUX_DEV void sums_from_double_body(int k, float* out_sums, const double* raw) { out_sums[k] = (float)raw[k]; }
// split-operand weight images of the neuron-split kernels (Spec2::BFX, v_mfma_f32_16x16x32_bf16): per hidden->hidden layer, neuron tile
// `ta`, k-block `kb` of 32 and piece (hi / mid / lo) one 1 KB fragment [64 lanes][8 bf16]; lane (g, c), element j <-> the OTHER index
// 16 (2 kb + (j >> 2)) + 4 g + (j & 3).  forward image: W[out = 16 ta + c][in = other]; transposed image: W[out = other][in = 16 ta + c].
// One thread per 32-bit word (two bf16).  Widths below the padded width read zeros.
constexpr int PACK_BF_MAX_LAYERS = 16;
struct PackBfArgs {
    const float* theta;
    unsigned* out_fwd;                   // packed + OFF_WB  (as 32-bit words)
    unsigned* out_tr;                    // packed + OFF_WTB
    int nhh, hp;                         // hidden->hidden layers, padded width
    float* packed;                       // the same launch gathers the fp32 image first (threads [0, n_gather): packed[i] = theta[idx[i]]);
    const int* idx;                      // n_gather = 0 when the optimiser's update kernel has written it already
    int n_gather;
    int woff[PACK_BF_MAX_LAYERS], nout[PACK_BF_MAX_LAYERS], nin[PACK_BF_MAX_LAYERS];      // per hidden->hidden layer: theta offset of W
                                         // (column-major: W[out + in nout]) and its real sizes
};
AUX_DEV unsigned bf16_rne(float x) {
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
AUX_DEV float bf16_val(unsigned h) { const unsigned u = h << 16; float x; __builtin_memcpy(&x, &u, 4); return x; }
AUX_DEV unsigned bf16_piece(float v, int piece) {
    const unsigned h = bf16_rne(v);
    if (piece == 0) return h;
    const float r = v - bf16_val(h);
    const unsigned m = bf16_rne(r);
    if (piece == 1) return m;
    return bf16_rne(r - bf16_val(m));
}
AUX_DEV void pack_bf16_body(int e, const PackBfArgs& a) {
    const int mt = a.hp / 16, kbn = a.hp / 32;
    const int total = a.nhh * mt * kbn * 3 * 256;        // words per image
    if (e >= 2 * total) return;
    const bool tr = e >= total;
    int r = tr ? e - total : e;
    const int j2 = r & 3; r >>= 2;
    const int lane = r & 63; r >>= 6;
    const int piece = r % 3; r /= 3;
    const int kb = r % kbn; r /= kbn;
    const int ta = r % mt; r /= mt;
    const int hl = r;
    const int g = lane >> 4, c = lane & 15;
    unsigned word = 0;
    for (int half = 0; half < 2; ++half) {
        const int j = 2 * j2 + half;
        const int other = 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3), mine = 16 * ta + c;
        const int out = tr ? other : mine, in = tr ? mine : other;
        const float v = (out < a.nout[hl] && in < a.nin[hl]) ? a.theta[a.woff[hl] + out + in * a.nout[hl]] : 0.f;
        word |= bf16_piece(v, piece) << (16 * half);
    }
    (tr ? a.out_tr : a.out_fwd)[tr ? e - total : e] = word;
}
// periodic-embedding rows of a point set (descriptor.cpp: apply_embeddings): point p of the installed set [n][du] -> device row
// [x_0 .. x_du-1 | sin / cos (omega x_src) ...] of width dx; the phase is formed in double so that the rows agree with the oracle's to
// float rounding
struct EmbedArgs {
    const float* upts;
    float* pts;
    int n, du, dx;
    int src[4], is_cos[4];
    double omega[4];
};
AUX_DEV void embed_body(int p, const EmbedArgs& a) {
    if (p >= a.n) return;
    for (int i = 0; i < a.du; ++i) a.pts[(size_t)p * a.dx + i] = a.upts[(size_t)p * a.du + i];
    for (int k = 0; k < a.dx - a.du; ++k) {
        const double ph = a.omega[k] * (double)a.upts[(size_t)p * a.du + a.src[k]];
        a.pts[(size_t)p * a.dx + a.du + k] = (float)(a.is_cos[k] ? cos(ph) : sin(ph));
    }
}
AUX_DEV void pack_body(int i, float* packed, const int* idx, const float* theta) {
    const int j = idx[i];
    packed[i] = (j >= 0) ? theta[j] : 0.f;
}
AUX_DEV void params_body(int j, float* params, const float* theta, const float* defaults, int ne, int p_off) {
    params[j] = (j < ne) ? theta[p_off + j] : defaults[j];
}
// thread e4 handles slab offsets 4*e4 .. 4*e4+3 (one 16-byte load per slab: dense and coalesced over consecutive threads);
// threads past the slab handle the loss columns
// loads of an unrolled body are independent of the (ordered) adds: 8 in flight instead of 1, same summation order
#ifdef PINN_EMU
#define AUX_UNROLL8
#else
#define AUX_UNROLL8 _Pragma("unroll 8")
#endif
AUX_DEV void reduce1_body(int e4, int chunk, int g, const Reduce1Args& a) {
    if (!a.active[g] || chunk >= a.nsplit[g]) return;
    const int nb = a.nblocks[g], ns = a.nsplit[g];
    const int per = (nb + ns - 1) / ns;
    const int b0 = chunk * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    double* out = a.tmp[g] + chunk;                      // tmp layout [entry][chunk]: stage 2 reads each entry's chunks contiguously
    const int nq = a.nent[g] / 4;                       // slab sizes are multiples of 4 (16 for the expression pseudo-group)
    if (e4 < nq) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const float* p = a.slabs[g] + 4 * e4;
        AUX_UNROLL8
        for (int b = b0; b < b1; ++b) {
            const F4 q = *reinterpret_cast<const F4*>(p + (size_t)b * a.slab[g]);      // one 16-byte load
            s0 += (double)q.x; s1 += (double)q.y; s2 += (double)q.z; s3 += (double)q.w;
        }
        out[(size_t)(4 * e4) * ns] = s0; out[(size_t)(4 * e4 + 1) * ns] = s1;
        out[(size_t)(4 * e4 + 2) * ns] = s2; out[(size_t)(4 * e4 + 3) * ns] = s3;
    } else if (e4 < nq + a.K) {
        const int k = e4 - nq;
        const double* p = a.losspart[g] + k;
        double s = 0.0;
        AUX_UNROLL8
        for (int wv = b0 * a.nwpb[g]; wv < b1 * a.nwpb[g]; ++wv) s += p[(size_t)wv * a.K];
        out[(size_t)(a.nent[g] + k) * ns] = s;
    }
}
// chunk `ch` of the one-kernel reduction: slab entries 4*e4 .. 4*e4+3 summed over the chunk's consecutive workgroups
AUX_DEV void reduce_one_chunk(int e4, int ch, const ReduceOneArgs& a, double (&s)[4]) {
    const int per = (a.nblocks + RONE_CHUNKS - 1) / RONE_CHUNKS;
    const int b0 = ch * per, b1 = (b0 + per < a.nblocks) ? b0 + per : a.nblocks;
    s[0] = s[1] = s[2] = s[3] = 0.0;
    const float* p = a.slabs + 4 * e4;
    AUX_UNROLL8
    for (int b = b0; b < b1; ++b) {
        const F4 q = *reinterpret_cast<const F4*>(p + (size_t)b * a.slab);
        s[0] += (double)q.x; s[1] += (double)q.y; s[2] += (double)q.z; s[3] += (double)q.w;
    }
}
// loss column k: rows strided over `nthreads` partial sums (thread t takes rows t, t + nthreads, ... of every launch in turn)
AUX_DEV double reduce_one_loss_part(int k, int t, int nthreads, const ReduceOneArgs& a) {
    double s = 0.0;
    for (int i = 0; i < a.nloss; ++i)
        for (int r = t; r < a.nrows[i]; r += nthreads) s += a.losspart[i][(size_t)r * a.K + k];
    return s;
}
AUX_DEV void reduce2_body(int r, const Reduce2Args& a) {
    if (r < a.P) {
        if (a.skip_grad) return;
        // contributions in batches of 4: their index loads (map -> group -> chunk row) are issued together, then the values are added
        // in the fixed map order — one contribution at a time the three dependent loads of each were serialised (12 us for a 2,209-
        // parameter net with 8 contributions per element)
        double s = 0.0;
        const int i0 = a.row_ptr[r], i1 = a.row_ptr[r + 1];
        for (int i = i0; i < i1; i += 4) {
            const double* t[4];
            int ns[4];
            for (int j = 0; j < 4; ++j) {
                const int ii = (i + j < i1) ? i + j : i;
                const int g = a.row_grp[ii];
                ns[j] = (i + j < i1 && a.ent_active[g]) ? a.nsplit[g] : 0;
                t[j] = a.tmp[g] + (size_t)a.row_ent[ii] * a.nsplit[g];
            }
            for (int j = 0; j < 4; ++j) {
                AUX_UNROLL8
                for (int ch = 0; ch < ns[j]; ++ch) s += t[j][ch];
            }
        }
        a.out[r] = (float)s;
    } else if (r < a.P + a.K) {
        const int k = r - a.P;
        double s = 0.0;
        for (int g = 0; g < a.ngroups; ++g) {
            if (!a.active[g]) continue;
            const double* t = a.tmp[g] + (size_t)(a.nent[g] + k) * a.nsplit[g];
            AUX_UNROLL8
            for (int ch = 0; ch < a.nsplit[g]; ++ch) s += t[ch];
        }
        a.sums[k] = (float)s;
        if (a.lossraw) a.lossraw[k] = s;
    }
}

// one-stage variant for SMALL launches (every group at most REDUCE_DIRECT_MAX workgroups): theta element r sums its slab entries over the
// workgroups directly, in the fixed order (map entry, workgroup); two kernels of ~6 us each for a 17-workgroup problem were a third of
// its evaluation time
constexpr int REDUCE_DIRECT_MAX = 32;
AUX_DEV void reduce_direct_body(int t, const Reduce1Args& a1, const Reduce2Args& a) {
    if (t >= a.P + a.K) return;
    const int r = a.perm ? a.perm[t] : t;
    if (r < a.P) {
        if (a.skip_grad) return;
        double s = 0.0;
        for (int i = a.row_ptr[r]; i < a.row_ptr[r + 1]; ++i) {
            const int g = a.row_grp[i];
            if (!a.ent_active[g]) continue;
            const float* p = a1.slabs[g] + a.row_ent[i];
            const int nb = a1.nblocks[g], st = a1.slab[g];
            AUX_UNROLL8
            for (int b = 0; b < nb; ++b) s += (double)p[(size_t)b * st];
        }
        a.out[r] = (float)s;
    } else if (r < a.P + a.K) {
        const int k = r - a.P;
        double s = 0.0;
        for (int g = 0; g < a.ngroups; ++g) {
            if (!a.active[g]) continue;
            const double* p = a1.losspart[g] + k;
            const int nw = a1.nblocks[g] * a1.nwpb[g];
            AUX_UNROLL8
            for (int wv = 0; wv < nw; ++wv) s += p[(size_t)wv * a.K];
        }
        a.sums[k] = (float)s;
        if (a.lossraw) a.lossraw[k] = s;
    }
}

inline bool reduce_is_small(const Reduce1Args& a1, const Reduce2Args& a2) {
    const char* e = std::getenv("PINN_REDUCE_DIRECT_MAX");       // (read per call: tests switch between the two paths; 0 = always two stages)
    const int lim = e ? std::atoi(e) : REDUCE_DIRECT_MAX;
    for (int g = 0; g < a2.ngroups; ++g)
        if (a1.active[g] && a1.nblocks[g] > lim) return false;
    return true;
}
#ifdef PINN_EMU
inline void launch_pack(float* packed, const int* idx, const float* theta, int n, plat_stream) {
    for (int i = 0; i < n; ++i) pack_body(i, packed, idx, theta);
}
inline void launch_embed(const EmbedArgs& a, plat_stream) {
    for (int p = 0; p < a.n; ++p) embed_body(p, a);
}
inline void launch_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off, plat_stream) {
    for (int j = 0; j < np; ++j) params_body(j, params, theta, defaults, ne, p_off);
}
inline void launch_pack_bf16(const PackBfArgs& a, plat_stream) {
    const int total = a.nhh * (a.hp / 16) * (a.hp / 32) * 3 * 256;
    for (int i = 0; i < a.n_gather; ++i) pack_body(i, a.packed, a.idx, a.theta);
    for (int e = 0; e < 2 * total; ++e) pack_bf16_body(e, a);
}
inline void launch_sums_from_double(float* out_sums, const double* raw, int K, plat_stream) {
    for (int k = 0; k < K; ++k) sums_from_double_body(k, out_sums, raw);
}
inline void launch_adam(float* theta, float* m, float* v, const float* grad, int P, float lr, float b1, float b2, float eps, float c1, float c2, plat_stream) {
    for (int i = 0; i < P; ++i) adam_body(i, theta, m, v, grad, lr, b1, b2, eps, c1, c2);
}
inline void launch_total_loss(double* hist, int step, const float* out, int P, int K, const float* w_over_n, plat_stream) {
    total_loss_body(hist, step, out, P, K, w_over_n);
}
inline void launch_sample(int kind, float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, unsigned draw, plat_stream) {
    for (int e = 0; e < n_elems; ++e) {
        if (kind == 3) sample_sobol_body(e, pts, d, lb, ub, seed, draw);
        else if (kind == 2) sample_lhs_body(e, pts, d, n_elems / d, lb, ub, seed, draw);
        else sample_body(e, pts, d, lb, ub, seed, draw);
    }
}
inline void launch_adam_fused(const AdamFusedArgs& a, plat_stream) {
    for (int i = 0; i < (a.P > 1 ? a.P : 1); ++i) adam_fused_body(i, a);
}
inline void launch_resample(const ResampleTerm* samp, int nsamp, int max_n, int step, plat_stream) {
    for (int p = 0; p < max_n; ++p) resample_point_sets(samp, nsamp, p, max_n, step);
}
// device-counter variants for the resident optimiser loop: the step index and the samplers' draw counters live in device memory, so one
// step's launch sequence is the same every step and can be replayed as a graph (engine.cpp: pinn_adam_steps)
inline void launch_adam_dev(float* theta, float* m, float* v, const float* grad, int P, float lr, float b1, float b2, float eps, const float* c12, const int* step, plat_stream st) {
    launch_adam(theta, m, v, grad, P, lr, b1, b2, eps, c12[2 * step[0]], c12[2 * step[0] + 1], st);
}
inline void launch_total_loss_dev(double* hist, const int* step, const float* out, int P, int K, const float* w_over_n, plat_stream st) {
    launch_total_loss(hist, step[0], out, P, K, w_over_n, st);
}
inline void launch_sample_dev(int kind, float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, const unsigned* draw, plat_stream st) {
    launch_sample(kind, pts, n_elems, d, lb, ub, seed, draw[0], st);
}
inline void launch_advance(int* step, unsigned* draws, const int* sampled, int K, plat_stream) {
    for (int t = 0; t < K; ++t) if (sampled[t]) ++draws[t];
    ++step[0];
}
inline void launch_src(const SrcArgs& a, plat_stream) {
    for (int p = 0; p < a.N; ++p) src_point(p, a);
}
inline void launch_expr(const ExprArgs& a, int nblocks, plat_stream) {
    for (int b = 0; b < nblocks; ++b)
        for (int w = 0; w < 4; ++w) {
            double ls = 0.0;
            double ps[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) {
                const int p = (b * 4 + w) * 64 + l;
                if (p >= a.N) continue;
                float pb[4];
                const float r = expr_point(p, a, pb);
                ls += (double)r * (double)r;
                for (int j = 0; j < 4; ++j) ps[j] += (double)pb[j];
            }
            if (!a.resid) {
                a.losspart[(size_t)(b * 4 + w) * a.K + a.term_id] = ls;
                for (int j = 0; j < 4; ++j) a.pslab[(size_t)(b * 4 + w) * 4 + j] = (float)ps[j];
            }
        }
}
inline void launch_reduce_one(const ReduceOneArgs& a, plat_stream) {
    for (int e4 = 0; e4 < a.nent / 4; ++e4) {
        double tot[4] = {0, 0, 0, 0}, s[4];
        for (int ch = 0; ch < RONE_CHUNKS; ++ch) {
            reduce_one_chunk(e4, ch, a, s);
            for (int e = 0; e < 4; ++e) tot[e] += s[e];
        }
        for (int e = 0; e < 4; ++e)
            if (a.ent_theta[4 * e4 + e] >= 0) a.out[a.ent_theta[4 * e4 + e]] = (float)tot[e];
    }
    for (int k = 0; k < a.K; ++k) {
        double part[256], s = 0.0;
        for (int t = 0; t < 256; ++t) part[t] = reduce_one_loss_part(k, t, 256, a);
        for (int st = 128; st >= 1; st >>= 1)
            for (int t = 0; t < st; ++t) part[t] += part[t + st];
        s = part[0];
        a.sums[k] = (float)s;
        if (a.lossraw) a.lossraw[k] = s;
    }
}
inline void launch_reduce(const Reduce1Args& a1, const Reduce2Args& a2, int max_n1, int max_split, plat_stream) {
    if (reduce_is_small(a1, a2)) {
        for (int r = 0; r < a2.P + a2.K; ++r) reduce_direct_body(r, a1, a2);
        return;
    }
    for (int g = 0; g < a2.ngroups; ++g)
        for (int ch = 0; ch < max_split; ++ch)
            for (int e = 0; e < max_n1; ++e) reduce1_body(e, ch, g, a1);
    for (int r = 0; r < a2.P + a2.K; ++r) reduce2_body(r, a2);
}
#else
__global__ void k_pack(float* packed, const int* idx, const float* theta, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pack_body(i, packed, idx, theta);
}
__global__ void k_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off) {
    const int j = threadIdx.x;
    if (j < np) params_body(j, params, theta, defaults, ne, p_off);
}
__global__ void k_adam(float* theta, float* m, float* v, const float* grad, int P, float lr, float b1, float b2, float eps, float c1, float c2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) adam_body(i, theta, m, v, grad, lr, b1, b2, eps, c1, c2);
}
__global__ void k_total_loss(double* hist, int step, const float* out, int P, int K, const float* w_over_n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) total_loss_body(hist, step, out, P, K, w_over_n);
}
__global__ void k_sample(float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, unsigned draw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_elems) sample_body(e, pts, d, lb, ub, seed, draw);
}
__global__ void k_sample_lhs(float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, unsigned draw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_elems) sample_lhs_body(e, pts, d, n_elems / d, lb, ub, seed, draw);
}
__global__ void k_sample_sobol(float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, unsigned draw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_elems) sample_sobol_body(e, pts, d, lb, ub, seed, draw);
}
inline void launch_adam(float* theta, float* m, float* v, const float* grad, int P, float lr, float b1, float b2, float eps, float c1, float c2, plat_stream st) {
    hipLaunchKernelGGL(k_adam, dim3((P + 255) / 256), dim3(256), 0, st, theta, m, v, grad, P, lr, b1, b2, eps, c1, c2);
}
inline void launch_total_loss(double* hist, int step, const float* out, int P, int K, const float* w_over_n, plat_stream st) {
    hipLaunchKernelGGL(k_total_loss, dim3(1), dim3(64), 0, st, hist, step, out, P, K, w_over_n);
}
inline void launch_sample(int kind, float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, unsigned draw, plat_stream st) {
    if (kind == 3) hipLaunchKernelGGL(k_sample_sobol, dim3((n_elems + 255) / 256), dim3(256), 0, st, pts, n_elems, d, lb, ub, seed, draw);
    else if (kind == 2) hipLaunchKernelGGL(k_sample_lhs, dim3((n_elems + 255) / 256), dim3(256), 0, st, pts, n_elems, d, lb, ub, seed, draw);
    else hipLaunchKernelGGL(k_sample, dim3((n_elems + 255) / 256), dim3(256), 0, st, pts, n_elems, d, lb, ub, seed, draw);
}
// every redrawn term of a problem in ONE launch (sample_rules.hpp: ResampleTerm): one thread per point index over the largest set
__global__ void __launch_bounds__(256) k_resample(const ResampleTerm* samp, int nsamp, int step) {
    resample_point_sets(samp, nsamp, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x), step);
}
inline void launch_resample(const ResampleTerm* samp, int nsamp, int max_n, int step, plat_stream st) {
    hipLaunchKernelGGL(k_resample, dim3((max_n + 255) / 256), dim3(256), 0, st, samp, nsamp, step);
}
__global__ void k_adam_fused(const AdamFusedArgs a) { adam_fused_body((int)(blockIdx.x * blockDim.x + threadIdx.x), a); }
inline void launch_adam_fused(const AdamFusedArgs& a, plat_stream st) {
    hipLaunchKernelGGL(k_adam_fused, dim3((a.P + 255) / 256), dim3(256), 0, st, a);
}
// device-counter variants for the resident optimiser loop (see the emulation section above)
__global__ void k_adam_dev(float* theta, float* m, float* v, const float* grad, int P, float lr, float b1, float b2, float eps, const float* c12, const int* step) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = step[0];
    if (i < P) adam_body(i, theta, m, v, grad, lr, b1, b2, eps, c12[2 * s], c12[2 * s + 1]);
}
__global__ void k_total_loss_dev(double* hist, const int* step, const float* out, int P, int K, const float* w_over_n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) total_loss_body(hist, step[0], out, P, K, w_over_n);
}
__global__ void k_sample_dev(int kind, float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, const unsigned* draw) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elems) return;
    const unsigned dr = draw[0];
    if (kind == 3) sample_sobol_body(e, pts, d, lb, ub, seed, dr);
    else if (kind == 2) sample_lhs_body(e, pts, d, n_elems / d, lb, ub, seed, dr);
    else sample_body(e, pts, d, lb, ub, seed, dr);
}
__global__ void k_advance(int* step, unsigned* draws, const int* sampled, int K) {
    const int t = threadIdx.x;
    if (t < K && sampled[t]) ++draws[t];
    if (t == 0) ++step[0];
}
inline void launch_adam_dev(float* theta, float* m, float* v, const float* grad, int P, float lr, float b1, float b2, float eps, const float* c12, const int* step, plat_stream st) {
    hipLaunchKernelGGL(k_adam_dev, dim3((P + 255) / 256), dim3(256), 0, st, theta, m, v, grad, P, lr, b1, b2, eps, c12, step);
}
inline void launch_total_loss_dev(double* hist, const int* step, const float* out, int P, int K, const float* w_over_n, plat_stream st) {
    hipLaunchKernelGGL(k_total_loss_dev, dim3(1), dim3(64), 0, st, hist, step, out, P, K, w_over_n);
}
inline void launch_sample_dev(int kind, float* pts, int n_elems, int d, const float* lb, const float* ub, unsigned seed, const unsigned* draw, plat_stream st) {
    hipLaunchKernelGGL(k_sample_dev, dim3((n_elems + 255) / 256), dim3(256), 0, st, kind, pts, n_elems, d, lb, ub, seed, draw);
}
inline void launch_advance(int* step, unsigned* draws, const int* sampled, int K, plat_stream st) {
    hipLaunchKernelGGL(k_advance, dim3(1), dim3(256), 0, st, step, draws, sampled, K);
}
__global__ void __launch_bounds__(256) k_expr(const ExprArgs a) {
    __shared__ double sh[5][256];
    const int p = blockIdx.x * 256 + threadIdx.x;
    float pb[4] = {0.f, 0.f, 0.f, 0.f};
    float r = 0.f;
    if (p < a.N) r = expr_point(p, a, pb);
    if (a.resid) return;
    sh[0][threadIdx.x] = (double)r * (double)r;
    for (int j = 0; j < 4; ++j) sh[1 + j][threadIdx.x] = (double)pb[j];
    __syncthreads();
    if (threadIdx.x < 20) {            // 4 waves x (loss + 4 params): serial fixed-order sums of 64 values each
        const int w = threadIdx.x & 3, q = threadIdx.x >> 2;
        double s = 0.0;
        for (int l = 0; l < 64; ++l) s += sh[q][w * 64 + l];
        if (q == 0) a.losspart[(size_t)(blockIdx.x * 4 + w) * a.K + a.term_id] = s;
        else a.pslab[(size_t)(blockIdx.x * 4 + w) * 4 + (q - 1)] = (float)s;
    }
}
__global__ void k_reduce1(const Reduce1Args a) {
    reduce1_body((int)(blockIdx.x * blockDim.x + threadIdx.x), (int)blockIdx.y, (int)blockIdx.z, a);
}
__global__ void k_reduce2(const Reduce2Args a) {
    reduce2_body((int)(blockIdx.x * blockDim.x + threadIdx.x), a);
}
inline void launch_pack(float* packed, const int* idx, const float* theta, int n, plat_stream st) {
    hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, st, packed, idx, theta, n);
}
__global__ void k_embed(const EmbedArgs a) { embed_body((int)(blockIdx.x * blockDim.x + threadIdx.x), a); }
inline void launch_embed(const EmbedArgs& a, plat_stream st) {
    hipLaunchKernelGGL(k_embed, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
}
inline void launch_params(float* params, const float* theta, const float* defaults, int np, int ne, int p_off, plat_stream st) {
    if (np > 0) hipLaunchKernelGGL(k_params, dim3(1), dim3(64), 0, st, params, theta, defaults, np, ne, p_off);
}
__global__ void k_pack_bf16(const PackBfArgs a) {
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e < a.n_gather) pack_body(e, a.packed, a.idx, a.theta);
    else pack_bf16_body(e - a.n_gather, a);
}
inline void launch_pack_bf16(const PackBfArgs& a, plat_stream st) {
    const int total = a.nhh * (a.hp / 16) * (a.hp / 32) * 3 * 256;
    hipLaunchKernelGGL(k_pack_bf16, dim3((a.n_gather + 2 * total + 255) / 256), dim3(256), 0, st, a);
}
__global__ void k_sums_from_double(float* out_sums, const double* raw, int K) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < K) sums_from_double_body(k, out_sums, raw);
}
inline void launch_sums_from_double(float* out_sums, const double* raw, int K, plat_stream st) {
    hipLaunchKernelGGL(k_sums_from_double, dim3((K + 63) / 64), dim3(64), 0, st, out_sums, raw, K);
}
__global__ void __launch_bounds__(256) k_src(const SrcArgs a) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < a.N) src_point(p, a);
}
inline void launch_src(const SrcArgs& a, plat_stream st) {
    hipLaunchKernelGGL(k_src, dim3((a.N + 255) / 256), dim3(256), 0, st, a);
}
inline void launch_expr(const ExprArgs& a, int nblocks, plat_stream st) {
    hipLaunchKernelGGL(k_expr, dim3(nblocks), dim3(256), 0, st, a);
}
__global__ void __launch_bounds__(256) k_reduce_one(const ReduceOneArgs a) {
    __shared__ double sh[RONE_CHUNKS][RONE_ENT + 1];
    const int tid = (int)threadIdx.x;
    const int nentblocks = (a.nent + RONE_ENT - 1) / RONE_ENT;
    if ((int)blockIdx.x < nentblocks) {
        const int j = tid & 7, ch = tid >> 3;
        const int e4 = (int)blockIdx.x * (RONE_ENT / 4) + j;
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        if (4 * e4 < a.nent) reduce_one_chunk(e4, ch, a, s);
        for (int e = 0; e < 4; ++e) sh[ch][4 * j + e] = s[e];
        __syncthreads();
        if (tid < RONE_ENT) {
            const int ent = (int)blockIdx.x * RONE_ENT + tid;
            if (ent < a.nent) {
                double tot = 0.0;
                for (int c2 = 0; c2 < RONE_CHUNKS; ++c2) tot += sh[c2][tid];
                const int th = a.ent_theta[ent];
                if (th >= 0) a.out[th] = (float)tot;
            }
        }
    } else {
        const int k = (int)blockIdx.x - nentblocks;
        double* part = &sh[0][0];
        part[tid] = reduce_one_loss_part(k, tid, 256, a);
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if (tid < st) part[tid] += part[tid + st];
            __syncthreads();
        }
        if (tid == 0) {
            a.sums[k] = (float)part[0];
            if (a.lossraw) a.lossraw[k] = part[0];
        }
    }
}
inline void launch_reduce_one(const ReduceOneArgs& a, plat_stream st) {
    hipLaunchKernelGGL(k_reduce_one, dim3((a.nent + RONE_ENT - 1) / RONE_ENT + a.K), dim3(256), 0, st, a);
}
__global__ void k_reduce_direct(const Reduce1Args a1, const Reduce2Args a2) {
    reduce_direct_body((int)(blockIdx.x * blockDim.x + threadIdx.x), a1, a2);
}
inline void launch_reduce(const Reduce1Args& a1, const Reduce2Args& a2, int max_n1, int max_split, plat_stream st) {
    if (reduce_is_small(a1, a2)) {
        hipLaunchKernelGGL(k_reduce_direct, dim3((a2.P + a2.K + 63) / 64), dim3(64), 0, st, a1, a2);
        return;
    }
    hipLaunchKernelGGL(k_reduce1, dim3((max_n1 + 255) / 256, max_split, a2.ngroups), dim3(256), 0, st, a1);
    hipLaunchKernelGGL(k_reduce2, dim3((a2.P + a2.K + 255) / 256), dim3(256), 0, st, a2);
}
#endif

}  // namespace aux
