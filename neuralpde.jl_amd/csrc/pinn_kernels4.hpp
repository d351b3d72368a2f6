// pinn_kernels4.hpp — "family 4": the FLOAT64 evaluation of the PINN loss and its gradient, one lane per collocation point.
//
// The reference's default parameter eltype is Float64 (src/discretize.jl:432-449) and its quasi-Newton stages rely on it
// (test/NNPDE1/nnpde__pde_iii_3rd_order_ode.jl:89-93: objective < 1e-9).  fp32 arithmetic — whatever the GEMM — cannot follow a TRAINED
// network below ~1e-5 relative (the residual is a small difference of O(1) terms), so the engine carries this opt-in
// precise mode (pinn_set_option(h, "precision", "f64")): same mathematics as the fp32 kernels — exact Taylor jets through the layers
// (the activation / jet rules of pinn_kernels.hpp instantiated with V = double), the residual tape (rprog.hpp with double immediates),
// the hand-derived reverse sweep — but built for accuracy and generality, not for the matrix pipe:
//   * one LANE per point, every per-point vector (records, post-activation jets, dZ) in point-major scratch rows [row][point] in
//     HBM / L2 (coalesced; a lane only touches its own column: no LDS, no barriers), weights as wave-uniform loads straight from theta
//     (double, ComponentArrays order — no packed image), layer widths and depth are RUN-TIME values: one kernel per (jet set, activation);
//   * the weight gradients by two more kernels into per-block slabs, summed in a fixed order — deterministic, no atomics: the
//     hidden-to-hidden matrices in 8 x 4 register tiles with the block's points across the lanes (k_f64_dwt), the remaining entries (first /
//     last layer, biases, PDE parameters, the sum of squares) one thread per entry (k_f64_dw);
//   * cost: VALU fp64 FMAs with L2-resident operands, register-blocked over 4-8 neurons (the scratch-row loads, not the FMAs, bound these
//     kernels) — 20x (the reference's own regime: nets of 12-64 neurons, 10^2-10^4 points) to a few hundred times (10^5+ points, 128-wide
//     nets) slower than the fp32 kernels (tools/time_f64.py): for finishing stages and digit-by-digit comparisons.
#pragma once
#include "pinn_kernels.hpp"

namespace pk {

constexpr int F64_MAX_LAYERS = 16;      // Dense layers including the output layer
constexpr int F64_MAX_ROWS = 96;        // tape rows [coordinates | params | slots | ops]
constexpr int F64_MAX_SLOTS = 24;
constexpr int F64_BLOCK = 512;          // points per block of the weight-gradient kernels (= rows of the per-block slabs)

constexpr int F64_MAX_NETS = 6;         // dependent variables one equation may reference (systems: test/NNPDE1/nnpde__pde_iii_3rd_order_ode.jl:70-76)

struct F64Net {
    int d;                              // inputs of the network
    int imap[4];                        // network input i = coordinate imap[i] of the term
    int nl;                             // Dense layers (hidden layers + the output layer)
    int sizes[F64_MAX_LAYERS + 1];      // n_0 .. n_nl, n_nl == 1
    int woff[F64_MAX_LAYERS], boff[F64_MAX_LAYERS];      // theta offsets of W_l (n_out x n_in, column-major) and b_l
    int r_rec[F64_MAX_LAYERS], r_post[F64_MAX_LAYERS], r_dz[F64_MAX_LAYERS];      // scratch row bases of hidden layer l: record / post-activation
                                                                                  // jets / dZ, H_l * C rows each (row = base + neuron * C + channel)
    int r_ubar;                         // seeds d(loss)/d(jet channel) of this network [C]
    int act_layers;                     // act == ACT_MIXED: kind (tanh / sigmoid) of hidden layer l in bits 4l .. 4l+3 (<= 8 hidden layers), as GroupArgs::act_layers
    int act;                            // ACT_TANH / ACT_SIGMOID / ACT_SIN on every hidden layer, or ACT_MIXED
    int theta0, nparams, ent0;          // the network's slice of theta; its first slab entry
    int tp0;                            // matrix-pipe kernels: first column of this network in a tile's row of partial sums (F64Args::tpart):
                                        // [W_0: n_1 * d][b_0: n_1][W_L: n_L][b_L]
    int ceff;                           // channel PREFIX this term reads from the network (1 + its highest referenced jet channel; a channel only depends on
                                        // lower ones): the sliced kernels (pinn_kernels6.hpp) carry channels [0, ceff) of this network and leave the rest zero —
                                        // an equation that couples several networks needs the full set from one of them only (cfg4: u 5, v 1, p 2 of C = 5)
};
// What differs between the terms of a MERGED launch of the matrix-pipe tile kernel (r06; f64.cpp: f64_eval_device): small problems — the reference's own
// regime, a few hundred points per term — are bound by the latency of one tile per launch, so every term that shares the networks, the input
// binding and an instantiated jet set rides in ONE launch (its tiles back to back, the tile's term looked up from sub_tile0).  A plain launch
// reads entry 0.
constexpr int F64_MAX_SUB = 6;
struct F64Sub {
    const double* pts; const float* pw; const double* data;
    const rp::Instr* prog; const double* imm;
    const double* lin;                  // affine residual (tile kernel fast path): [nslots coefficients a_s | N values of the coordinate-only part]; nullptr: run the tape
    double scale;
    int N, nops, out_row, nslots;
    unsigned char slot_net[F64_MAX_SLOTS], slot_chan[F64_MAX_SLOTS];
};
struct F64Args {
    const double* theta;                // the whole parameter vector
    const double* pts;                  // [N][dt] point-major
    const float* pw;                    // per-point factors sqrt(N w_i) of a quadrature-weighted term, nullable
    const double* data;                 // [ndata][N] user-supplied per-point channels of the term (OP_DATA: observations of a data-misfit term), nullable
    const double* lin;                  // matrix-pipe tile kernel: the term's affine form (F64Sub::lin), nullable
    int N, p0, npts;                    // points of the term; first point and point count of this launch (one chunk)
    int dt;                             // coordinates per point of the term
    int nnets;                          // networks the equation references (all with the same number of inputs: one jet set serves them)
    F64Net net[F64_MAX_NETS];
    int np, ne, p_off;                  // tape rows of PDE parameters; the first ne live in theta at p_off, the rest are defaults
    double pdef[MAX_PARAMS];
    const rp::Instr* prog;              // descriptor numbering: rows [coordinates dt | params np | slots | ops]
    const double* imm;                  // the ops' immediates in double (Instr::imm is a float)
    int nops, out_row, nslots;
    int slot_net[F64_MAX_SLOTS];        // slot s reads network slot_net[s] (index into `net`) ...
    int slot_chan[F64_MAX_SLOTS];       // ... jet channel slot_chan[s]
    double scale;                       // 2 w_k / N_norm: the reverse sweep's seed factor
    double* scratch;                    // rows [nrows][npad] (family 4); the matrix-pipe kernels (pinn_kernels5.hpp) lay the same rows out point-block-major
    int npad;
    int nrows;                          // total scratch rows of the term (pinn_kernels5.hpp: f64m_six)
    int post_alias;                     // matrix-pipe kernels, value-only tanh / sigmoid terms: r_post == r_rec (one copy: the record is the activation)
    int r_pbar, r_sq;                   // PDE-parameter partials [ne], squared weighted residual [1]
    int mode;                           // 0: loss + gradient, 1: loss only, 2: residual values into `resid`
    double* resid;
    // SEEDED launches (the reference-semantics "stencil" validation mode, f64.cpp: f64_stencil_*): the term is one network's value at (shifted)
    // points and the reverse sweep starts from a per-point seed computed elsewhere (the stencil tape, k_f64_stape) instead of from this launch's
    // own tape: record [seed_stride] per point = { d loss / d u(point), squared weighted residual, PDE-parameter partials [ne] }; nullptr: off
    const double* seed;
    int seed_stride;
    int use_ceff;                       // 1: the scratch rows hold channels [0, F64Net::ceff) of every network only (written by a sliced tile kernel): the dW kernels stop there
    // weight-gradient kernel
    int C, first_ch[8];                 // channel of d/dx_i (-1: not carried)
    double* slab;                       // [nblocks][nent], nent = sum of the networks' parameters + ne + 1 (last entry: the block's sum of squares)
    int nent, ent_p;                    // ent_p: first PDE-parameter entry
    // matrix-pipe kernels (pinn_kernels5.hpp): the tile kernel leaves the sums of everything that is not a hidden-to-hidden weight / bias — first and
    // last layer, PDE parameters, the squared residuals — per TILE: [tiles][ntp], columns = the networks' blocks (F64Net::tp0), then [ne] PDE
    // parameters, then the sum of squares; f64m_tsum_entry adds a block's tiles in order into the slab
    double* tpart;
    int ntp, tp_p, tile_pts;            // columns per tile; first PDE-parameter column; points per tile (16 * PG)
    // the tile kernel's per-term view (F64Sub above): nsub == 0: a plain launch, sub[0] repeats this struct's own fields; nsub >= 1: a merged launch —
    // sub-term s owns tiles [sub_tile0[s], sub_tile0[s + 1]), its points are 0 .. sub[s].N - 1 of its own set, and the slab ends in nsub sums of squares
    int nsub;
    int block_pts;                      // matrix-pipe dW kernel: points per block (= slab row); 0: F64_BLOCK.  Small launches use short blocks so that the
                                        // few hundred points of a small problem spread over many workgroups instead of one (f64.cpp)
    int sub_tile0[F64_MAX_SUB + 1];
    F64Sub sub[F64_MAX_SUB];
};
static_assert(sizeof(F64Args) <= 4096, "F64Args is a by-value kernel argument (4 KB)");
HD int f64m_block(const F64Args& a) { return a.block_pts > 0 ? a.block_pts : F64_BLOCK; }
// number of sum-of-squares entries at the end of a block's slab row
HD int f64_nsq(const F64Args& a) { return a.nsub > 0 ? a.nsub : 1; }

// activation kind of hidden layer l: tanh and sigmoid are a RUN-TIME kind in the float64 kernels, so per-layer mixes (Lux chains like
// Dense(.., sigma) -> Dense(.., tanh), test/NNPDE2/additional_loss__lorenz_system.jl) cost nothing extra
HD int f64_act(const F64Net& n, int l) { return n.act == ACT_MIXED ? ((n.act_layers >> (4 * l)) & 15) : n.act; }

// ---- kernel A: forward jets of every network, residual tape, reverse sweep of ONE point ----
template <class J, int ACTK /* ACT_TANH: tanh / sigmoid by run-time kind; ACT_SIN: sin */>
DEV void f64_point(int lp, const F64Args& a) {
    constexpr int C = J::C;
    constexpr bool SIN = (ACTK == ACT_SIN);
    constexpr int MB = (C <= 3) ? 8 : 4;                         // neurons per register block (MB * C accumulators in double)
    const int p = a.p0 + lp;
    double* S = a.scratch + lp;                                  // element `row` of this point: S[row * npad]
    const size_t np_ = (size_t)a.npad;
    const double* sd = a.seed ? a.seed + (size_t)p * (size_t)a.seed_stride : nullptr;
    if (sd && a.mode == 1) { S[(size_t)a.r_sq * np_] = sd[1]; return; }      // seeded loss-only launch: the squared residual is the seed record's
    double U[F64_MAX_NETS][C];
    // =========================== forward ===========================
    for (int ni = 0; ni < a.nnets; ++ni) {
        const F64Net& n = a.net[ni];
        const int L = n.nl - 1;                                  // hidden layers
        double x[4] = {0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < n.d; ++i) x[i] = a.pts[(size_t)p * a.dt + n.imap[i]];
        for (int l = 0; l < L; ++l) {
            const int n_in = n.sizes[l], n_out = n.sizes[l + 1];
            const double* W = a.theta + n.woff[l];
            const double* B = a.theta + n.boff[l];
            // MB output neurons at a time: every input jet loaded from the scratch rows feeds MB accumulators (the rows live in L2 / HBM: the
            // loads, not the FMAs, bound this kernel).  Per output the sum runs over k in the same order whatever MB: blocking changes no bit.
            for (int m0 = 0; m0 < n_out; m0 += MB) {
                const int nb = (n_out - m0 < MB) ? n_out - m0 : MB;
                double z[MB][C];
                int mj[MB];
                PINN_UNROLL for (int j = 0; j < MB; ++j) {
                    mj[j] = (j < nb) ? m0 + j : n_out - 1;           // (tail lanes of the block repeat the last neuron; never stored)
                    z[j][0] = B[mj[j]];
                    PINN_UNROLL for (int c = 1; c < C; ++c) z[j][c] = 0.0;
                }
                if (l == 0) {
                    PINN_UNROLL for (int j = 0; j < MB; ++j) {
                        for (int i = 0; i < n.d; ++i) z[j][0] = vfma(W[mj[j] + (size_t)i * n_out], x[i], z[j][0]);
                        PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf) z[j][J::CH_FIRST + kf] = W[mj[j] + (size_t)J::first_axis(kf) * n_out];
                    }
                } else {
                    const size_t base = (size_t)n.r_post[l - 1];
                    for (int k = 0; k < n_in; ++k) {
                        double ak[C];
                        PINN_UNROLL for (int c = 0; c < C; ++c) ak[c] = S[(base + (size_t)k * C + c) * np_];
                        PINN_UNROLL for (int j = 0; j < MB; ++j) {
                            const double w = W[mj[j] + (size_t)k * n_out];
                            PINN_UNROLL for (int c = 0; c < C; ++c) z[j][c] = vfma(w, ak[c], z[j][c]);
                        }
                    }
                }
                PINN_UNROLL for (int j = 0; j < MB; ++j) {
                    if (j >= nb) break;
                    const int m = m0 + j;
                    const double a0 = act_value<SIN>(f64_act(n, l), z[j][0]);
                    z[j][0] = act_record<SIN>(z[j][0], a0);          // the record: a (tanh / sigmoid) or z (sin), then the pre-activation channels
                    PINN_UNROLL for (int c = 0; c < C; ++c) S[((size_t)n.r_rec[l] + (size_t)m * C + c) * np_] = z[j][c];
                    double dd[ND];
                    act_derivs_n<J::NORD - 1, SIN>(f64_act(n, l), z[j][0], dd);
                    jet_forward<J>(z[j], dd);
                    z[j][0] = a0;
                    PINN_UNROLL for (int c = 0; c < C; ++c) S[((size_t)n.r_post[l] + (size_t)m * C + c) * np_] = z[j][c];
                }
            }
        }
        const int n_in = n.sizes[L];
        const double* W = a.theta + n.woff[L];
        U[ni][0] = a.theta[n.boff[L]];
        PINN_UNROLL for (int c = 1; c < C; ++c) U[ni][c] = 0.0;
        if (L == 0) {                                            // a single Dense layer: u = W x + b
            for (int i = 0; i < n.d; ++i) U[ni][0] = vfma(W[i], x[i], U[ni][0]);
            PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf) U[ni][J::CH_FIRST + kf] = W[J::first_axis(kf)];
        } else {
            const size_t base = (size_t)n.r_post[L - 1];
            for (int k = 0; k < n_in; ++k) {
                const double w = W[k];
                PINN_UNROLL for (int c = 0; c < C; ++c) U[ni][c] = vfma(w, S[(base + (size_t)k * C + c) * np_], U[ni][c]);
            }
        }
    }
    // =========================== residual tape ===========================
    double v[F64_MAX_ROWS];
    double g[F64_MAX_ROWS];
    double rbar = 0.0;
    if (!sd) {
    const int R0 = a.dt + a.np + a.nslots;
    for (int i = 0; i < a.dt; ++i) v[i] = a.pts[(size_t)p * a.dt + i];
    for (int j = 0; j < a.np; ++j) v[a.dt + j] = j < a.ne ? a.theta[a.p_off + j] : a.pdef[j];
    for (int s = 0; s < a.nslots; ++s) {
        double u = 0.0;
        for (int ni = 0; ni < a.nnets; ++ni)
            PINN_UNROLL for (int c = 0; c < C; ++c) if (a.slot_net[s] == ni && a.slot_chan[s] == c) u = U[ni][c];
        v[a.dt + a.np + s] = u;
    }
    for (int q = 0; q < a.nops; ++q) {
        const rp::Instr ins = a.prog[q];
        const double va = rp::is_nullary(ins.code) ? 0.0 : v[ins.a];
        const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
        v[R0 + q] = (ins.code == rp::OP_DATA) ? a.data[(size_t)(int)a.imm[q] * (size_t)a.N + (size_t)p] : rp::apply<double, double>(ins.code, va, vb, a.imm[q]);
    }
    const double r = v[a.out_row];
    if (a.mode == 2) { a.resid[p] = r; return; }
    const double sw = a.pw ? (double)a.pw[p] : 1.0;
    const double rs = r * sw;
    S[(size_t)a.r_sq * np_] = rs * rs;
    if (a.mode == 1) return;
    for (int q = 0; q < R0 + a.nops; ++q) g[q] = 0.0;
    g[a.out_row] = 1.0;
    for (int q = a.nops - 1; q >= 0; --q) {
        const rp::Instr ins = a.prog[q];
        if (rp::is_nullary(ins.code)) continue;
        const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
        double da, db;
        rp::adjoint<double, double>(ins.code, v[ins.a], vb, v[R0 + q], a.imm[q], g[R0 + q], da, db);
        g[ins.a] += da;
        if (rp::is_binary(ins.code)) g[ins.b] += db;
    }
    rbar = rs * a.scale * sw;
    for (int j = 0; j < a.ne; ++j) S[((size_t)a.r_pbar + j) * np_] = rbar * g[a.dt + j];
    } else {
        // seeded: d loss / d (slot s) = the record's seed for slot 0 (one network, value channel), nothing for any other row
        if (a.mode == 2) { a.resid[p] = U[0][0]; return; }
        S[(size_t)a.r_sq * np_] = sd[1];
        for (int j = 0; j < a.ne; ++j) S[((size_t)a.r_pbar + j) * np_] = sd[2 + j];
        for (int s = 0; s < a.nslots; ++s) g[a.dt + a.np + s] = 0.0;
        g[a.dt + a.np] = 1.0;
        rbar = sd[0];
    }
    // =========================== reverse sweep, network by network ===========================
    for (int ni = 0; ni < a.nnets; ++ni) {
        const F64Net& n = a.net[ni];
        const int L = n.nl - 1;
        double ubar[C];
        PINN_UNROLL for (int c = 0; c < C; ++c) ubar[c] = 0.0;
        for (int s = 0; s < a.nslots; ++s) {
            if (a.slot_net[s] != ni) continue;
            const double gs = rbar * g[a.dt + a.np + s];
            PINN_UNROLL for (int c = 0; c < C; ++c) if (a.slot_chan[s] == c) ubar[c] += gs;
        }
        PINN_UNROLL for (int c = 0; c < C; ++c) S[((size_t)n.r_ubar + c) * np_] = ubar[c];
        for (int l = L - 1; l >= 0; --l) {
            const int H = n.sizes[l + 1];
            const int n_next = n.sizes[l + 2];                   // neurons of the layer above (1 for the output layer)
            const double* Wn = a.theta + n.woff[l + 1];          // W_{l+1}[m + k * n_next]
            for (int k0 = 0; k0 < H; k0 += MB) {
                const int nb = (H - k0 < MB) ? H - k0 : MB;
                double gq[MB][C];
                int kj[MB];
                PINN_UNROLL for (int j = 0; j < MB; ++j) kj[j] = (j < nb) ? k0 + j : H - 1;
                if (l == L - 1) {
                    PINN_UNROLL for (int j = 0; j < MB; ++j) {
                        const double w = Wn[kj[j]];
                        PINN_UNROLL for (int c = 0; c < C; ++c) gq[j][c] = w * ubar[c];
                    }
                } else {
                    PINN_UNROLL for (int j = 0; j < MB; ++j)
                        PINN_UNROLL for (int c = 0; c < C; ++c) gq[j][c] = 0.0;
                    const size_t base = (size_t)n.r_dz[l + 1];
                    for (int m = 0; m < n_next; ++m) {
                        double dzm[C];
                        PINN_UNROLL for (int c = 0; c < C; ++c) dzm[c] = S[(base + (size_t)m * C + c) * np_];
                        PINN_UNROLL for (int j = 0; j < MB; ++j) {
                            const double w = Wn[m + (size_t)kj[j] * n_next];
                            PINN_UNROLL for (int c = 0; c < C; ++c) gq[j][c] = vfma(w, dzm[c], gq[j][c]);
                        }
                    }
                }
                PINN_UNROLL for (int j = 0; j < MB; ++j) {
                    if (j >= nb) break;
                    const int k = k0 + j;
                    double s[C], dd[ND];
                    PINN_UNROLL for (int c = 0; c < C; ++c) s[c] = S[((size_t)n.r_rec[l] + (size_t)k * C + c) * np_];
                    act_derivs_n<J::NORD, SIN>(f64_act(n, l), s[0], dd);
                    jet_adjoint<J>(gq[j], s, dd);
                    PINN_UNROLL for (int c = 0; c < C; ++c) S[((size_t)n.r_dz[l] + (size_t)k * C + c) * np_] = gq[j][c];
                }
            }
        }
    }
}

// ---- kernel B: entry e of the slab of point block b: one theta element of one of the networks (theta order), one PDE parameter, or the
// block's sum of squared residuals; fixed order over the block's points ----
DEV void f64_dw_entry(int e, int b, const F64Args& a) {
    const int lo = b * F64_BLOCK, hi = (lo + F64_BLOCK < a.npts) ? lo + F64_BLOCK : a.npts;
    const size_t np_ = (size_t)a.npad;
    const double* S = a.scratch;
    const int C = a.C;
    double s = 0.0;
    if (e == a.nent - 1) {
        for (int p = lo; p < hi; ++p) s += S[(size_t)a.r_sq * np_ + p];
    } else if (e >= a.ent_p) {
        if (a.mode == 0) {
            const int j = e - a.ent_p;
            for (int p = lo; p < hi; ++p) s += S[((size_t)a.r_pbar + j) * np_ + p];
        }
    } else if (a.mode == 0) {
        int ni = 0;
        while (ni + 1 < a.nnets && e >= a.net[ni + 1].ent0) ++ni;
        const F64Net& n = a.net[ni];
        const int L = n.nl - 1;
        // which layer does theta element t belong to?
        const int t = n.theta0 + (e - n.ent0);
        int l = 0;
        while (l + 1 < n.nl && t >= n.woff[l + 1]) ++l;
        const int n_out = n.sizes[l + 1];
        const bool bias = t >= n.boff[l];
        if (!bias && l >= 1 && l < L) return;                  // hidden-to-hidden weight matrices: the tiled kernel below writes these entries
        const int m = bias ? t - n.boff[l] : (t - n.woff[l]) % n_out;
        const int k = bias ? 0 : (t - n.woff[l]) / n_out;
        // dZ of this layer's outputs: hidden layer l's dZ rows, or (output layer) the seeds ubar
        const size_t dz = (l == L) ? (size_t)n.r_ubar : (size_t)n.r_dz[l] + (size_t)m * C;
        if (bias) {
            for (int p = lo; p < hi; ++p) s += S[dz * np_ + p];
        } else if (l == 0) {
            // inputs of the first layer: value channel x_k, first-derivative channel of axis k = 1, everything else 0
            const int ck = a.first_ch[k];
            for (int p = lo; p < hi; ++p) {
                const int gp = a.p0 + p;
                double t2 = S[dz * np_ + p] * a.pts[(size_t)gp * a.dt + n.imap[k]];
                if (ck >= 0) t2 += S[(dz + ck) * np_ + p];
                s += t2;
            }
        } else {
            const size_t in = (size_t)n.r_post[l - 1] + (size_t)k * C;
            for (int p = lo; p < hi; ++p) {
                double t2 = 0.0;
                for (int c = 0; c < C; ++c) t2 = vfma(S[(dz + c) * np_ + p], S[(in + c) * np_ + p], t2);
                s += t2;
            }
        }
    }
    a.slab[(size_t)b * a.nent + e] = s;
}

// ---- kernel B2: the hidden-to-hidden weight matrices (all but a sliver of theta) in TILES of F64_TM outputs x F64_TK inputs: the points
// of a block across the lanes (coalesced row reads), every loaded dZ / input jet feeds a whole row / column of the tile's accumulators
// (12 loads per 32 FMAs; one thread per entry needs 2 per FMA and reads 64 different rows per load), lane partials summed in lane order:
// fixed order, no atomics ----
constexpr int F64_TM = 8, F64_TK = 4;
HD int f64_layer_tiles(const F64Net& n, int l) { return ((n.sizes[l + 1] + F64_TM - 1) / F64_TM) * ((n.sizes[l] + F64_TK - 1) / F64_TK); }
HD int f64_num_tiles(const F64Args& a) {
    int t = 0;
    for (int ni = 0; ni < a.nnets; ++ni)
        for (int l = 1; l < a.net[ni].nl - 1; ++l) t += f64_layer_tiles(a.net[ni], l);
    return t;
}
struct F64Tile { int ni, l, m0, k0; };
HD F64Tile f64_tile_locate(const F64Args& a, int tile) {
    F64Tile T = {0, 1, 0, 0};
    for (int ni = 0; ni < a.nnets; ++ni)
        for (int l = 1; l < a.net[ni].nl - 1; ++l) {
            const int nt = f64_layer_tiles(a.net[ni], l);
            if (tile < nt) {
                const int kt = (a.net[ni].sizes[l] + F64_TK - 1) / F64_TK;
                T.ni = ni; T.l = l; T.m0 = (tile / kt) * F64_TM; T.k0 = (tile % kt) * F64_TK;
                return T;
            }
            tile -= nt;
        }
    return T;
}
// one lane's partial sums over its points p = lo + lane, lo + lane + 64, ... of block b
DEV void f64_dwt_lane(const F64Tile& T, int b, int lane, const F64Args& a, double (&acc)[F64_TM * F64_TK]) {
    const F64Net& n = a.net[T.ni];
    const int lo = b * F64_BLOCK, hi = (lo + F64_BLOCK < a.npts) ? lo + F64_BLOCK : a.npts;
    const size_t np_ = (size_t)a.npad;
    const double* S = a.scratch;
    const int C = a.C, n_out = n.sizes[T.l + 1], n_in = n.sizes[T.l];
    size_t rz[F64_TM], ri[F64_TK];
    PINN_UNROLL for (int j = 0; j < F64_TM; ++j) rz[j] = (size_t)n.r_dz[T.l] + (size_t)((T.m0 + j < n_out) ? T.m0 + j : n_out - 1) * C;
    PINN_UNROLL for (int i = 0; i < F64_TK; ++i) ri[i] = (size_t)n.r_post[T.l - 1] + (size_t)((T.k0 + i < n_in) ? T.k0 + i : n_in - 1) * C;
    PINN_UNROLL for (int e = 0; e < F64_TM * F64_TK; ++e) acc[e] = 0.0;
    for (int p = lo + lane; p < hi; p += 64)
        for (int c = 0; c < C; ++c) {
            double dz[F64_TM], in[F64_TK];
            PINN_UNROLL for (int j = 0; j < F64_TM; ++j) dz[j] = S[(rz[j] + c) * np_ + p];
            PINN_UNROLL for (int i = 0; i < F64_TK; ++i) in[i] = S[(ri[i] + c) * np_ + p];
            PINN_UNROLL for (int j = 0; j < F64_TM; ++j)
                PINN_UNROLL for (int i = 0; i < F64_TK; ++i) acc[j * F64_TK + i] = vfma(dz[j], in[i], acc[j * F64_TK + i]);
        }
}
DEV void f64_dwt_store(const F64Tile& T, int b, int e, double s, const F64Args& a) {
    const F64Net& n = a.net[T.ni];
    const int m = T.m0 + e / F64_TK, k = T.k0 + e % F64_TK, n_out = n.sizes[T.l + 1];
    if (m >= n_out || k >= n.sizes[T.l]) return;
    a.slab[(size_t)b * a.nent + n.ent0 + (n.woff[T.l] - n.theta0) + m + (size_t)k * n_out] = s;
}

// ---- kernel C: sum the per-block slabs in block order onto the gradient / the term's sum of squares ----
struct F64ReduceArgs {
    const double* slab; int nblocks, nent, ent_p;
    double* grad;                        // [P] (accumulated: += )
    int nnets, ent0[F64_MAX_NETS], theta0[F64_MAX_NETS];
    int p_off;
    double* sumsq;                       // += the term's sum of squares
    int nsq, sq_off[F64_MAX_SUB];        // merged launches: the slab row ends in nsq sums; entry k goes to sumsq[sq_off[k]] (plain: nsq = 1, sq_off[0] = 0)
    int with_grad;
    int init_grad, init_sumsq;           // 1: this launch WRITES its sums (the evaluation's first reduction covering all of theta / the term's first chunk)
                                         // instead of adding to them: no memset launches in front of an evaluation
};
DEV void f64_reduce_write(int e, double s, const F64ReduceArgs& a) {
    if (e >= a.nent - a.nsq) { double* q = a.sumsq + a.sq_off[e - (a.nent - a.nsq)]; *q = a.init_sumsq ? s : *q + s; return; }
    int i;
    if (e >= a.ent_p) i = a.p_off + (e - a.ent_p);
    else {
        int ni = 0;
        while (ni + 1 < a.nnets && e >= a.ent0[ni + 1]) ++ni;
        i = a.theta0[ni] + (e - a.ent0[ni]);
    }
    a.grad[i] = a.init_grad ? s : a.grad[i] + s;
}
DEV void f64_reduce_entry(int e, const F64ReduceArgs& a) {
    if (e < a.nent - a.nsq && !a.with_grad) return;
    double s = 0.0;
    for (int b = 0; b < a.nblocks; ++b) s += a.slab[(size_t)b * a.nent + e];
    f64_reduce_write(e, s, a);
}

// ---- the reference-semantics ("stencil") validation mode: derivative slots as the reference's CENTRAL DIFFERENCES (numeric_derivative,
// src/pinn_types.jl:445-482, steps from get_eps, src/symbolic_utilities.jl:98-103) instead of exact Taylor jets.  Not a performance path: it
// exists so that the engine can be compared digit by digit with the reference's generated loss functions (NeuralPDEHIP.selftest) and with the
// stencil oracle at TRAINED parameters, where exact and finite-difference derivatives differ by 1e-3 in the gradient.  Every derivative is a
// linear combination of VALUES of the trial function at shifted points, so an evaluation is (f64.cpp: f64_stencil_term)
//   1. one value-only launch (k_f64_point, mode 2) per distinct (network, shift sequence) — a "virtual network" — on its shifted copy of the set,
//   2. k_f64_stape: per point the STENCIL TAPE — the reference's difference formulas as tape ops in the reference's order of operations over the
//      virtual networks' values, then the term's own tape — forward and reverse: residual, squared residual, and d loss / d (value of virtual
//      network v at this point) = the seed of
//   3. one SEEDED loss + gradient launch per virtual network (F64Args::seed): forward again, reverse sweep from the seed, the usual
//      weight-gradient kernels and slab reduction, accumulated into the gradient. ----
struct F64ShiftArgs {
    const double* pts; int dt;          // the term's set [n][dt]
    double* out; int d;                 // the virtual network's set [n][d]: out[p][i] = pts[p][imap[i]], then the shifts in order (one rounding each,
    int imap[4];                        // like the reference's x .+ eps, then .+ the inner level's eps, ...)
    int nshift, axis[8];
    double delta[8];
    int n;
};
DEV void f64_shift_point(int p, const F64ShiftArgs& a) {
    double x[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < a.d; ++i) x[i] = a.pts[(size_t)p * a.dt + a.imap[i]];
    for (int s = 0; s < a.nshift; ++s)
        for (int i = 0; i < a.d; ++i) if (i == a.axis[s]) x[i] = x[i] + a.delta[s];
    for (int i = 0; i < a.d; ++i) a.out[(size_t)p * a.d + i] = x[i];
}
struct F64StapeArgs {
    const double* pts; int dt, n;       // the term's set
    const double* theta;                // PDE parameters theta.p
    int np, ne, p_off;
    double pdef[MAX_PARAMS];
    const float* pw;                    // quadrature factors sqrt(N w_i), nullable
    const double* data;                 // [ndata][n], nullable
    const double* uv; int nv;           // values of the virtual networks [nv][n]: tape rows dt + np + v
    const rp::Instr* prog; const double* imm; int nops, out_row;      // stencil tape: difference formulas, then the term's own ops
    double scale;                       // 2 w_k / N_norm
    int mode;                           // 0: seeds for a loss + gradient evaluation, 1: squared residuals only, 2: residuals into `resid`
    double* resid;
    double* seeds; int seed_stride;     // [nv][n][seed_stride]: { d loss / d uv[v][p], (v == 0: squared weighted residual, parameter partials; else 0) }
};
DEV void f64_stape_point(int p, const F64StapeArgs& a) {
    double v[F64_MAX_ROWS];
    const int R0 = a.dt + a.np + a.nv;
    for (int i = 0; i < a.dt; ++i) v[i] = a.pts[(size_t)p * a.dt + i];
    for (int j = 0; j < a.np; ++j) v[a.dt + j] = j < a.ne ? a.theta[a.p_off + j] : a.pdef[j];
    for (int s = 0; s < a.nv; ++s) v[a.dt + a.np + s] = a.uv[(size_t)s * a.n + p];
    for (int q = 0; q < a.nops; ++q) {
        const rp::Instr ins = a.prog[q];
        const double va = rp::is_nullary(ins.code) ? 0.0 : v[ins.a];
        const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
        v[R0 + q] = (ins.code == rp::OP_DATA) ? a.data[(size_t)(int)a.imm[q] * (size_t)a.n + (size_t)p] : rp::apply<double, double>(ins.code, va, vb, a.imm[q]);
    }
    const double r = v[a.out_row];
    if (a.mode == 2) { a.resid[p] = r; return; }
    const double sw = a.pw ? (double)a.pw[p] : 1.0;
    const double rs = r * sw;
    double* sd0 = a.seeds + (size_t)p * a.seed_stride;
    sd0[1] = rs * rs;
    if (a.mode == 1) return;
    double g[F64_MAX_ROWS];
    for (int q = 0; q < R0 + a.nops; ++q) g[q] = 0.0;
    g[a.out_row] = 1.0;
    for (int q = a.nops - 1; q >= 0; --q) {
        const rp::Instr ins = a.prog[q];
        if (rp::is_nullary(ins.code)) continue;
        const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
        double da, db;
        rp::adjoint<double, double>(ins.code, v[ins.a], vb, v[R0 + q], a.imm[q], g[R0 + q], da, db);
        g[ins.a] += da;
        if (rp::is_binary(ins.code)) g[ins.b] += db;
    }
    const double rbar = rs * a.scale * sw;
    for (int j = 0; j < a.ne; ++j) sd0[2 + j] = rbar * g[a.dt + j];
    for (int s = 0; s < a.nv; ++s) {
        double* sd = a.seeds + ((size_t)s * a.n + p) * a.seed_stride;
        sd[0] = rbar * g[a.dt + a.np + s];
        if (s > 0) { sd[1] = 0.0; for (int j = 0; j < a.ne; ++j) sd[2 + j] = 0.0; }
    }
}
#ifdef PINN_EMU
inline void launch_f64_shift(const F64ShiftArgs& a, plat_stream) { for (int p = 0; p < a.n; ++p) f64_shift_point(p, a); }
inline void launch_f64_stape(const F64StapeArgs& a, plat_stream) { for (int p = 0; p < a.n; ++p) f64_stape_point(p, a); }
#else
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_shift(const F64ShiftArgs a) {
    const int p = (int)(blockIdx.x * 256 + threadIdx.x);
    if (p < a.n) f64_shift_point(p, a);
}
template <int UNUSED> __global__ void __launch_bounds__(64) k_f64_stape(const F64StapeArgs a) {
    const int p = (int)(blockIdx.x * 64 + threadIdx.x);
    if (p < a.n) f64_stape_point(p, a);
}
inline void launch_f64_shift(const F64ShiftArgs& a, plat_stream st) { hipLaunchKernelGGL((k_f64_shift<0>), dim3((a.n + 255) / 256), dim3(256), 0, st, a); }
inline void launch_f64_stape(const F64StapeArgs& a, plat_stream st) { hipLaunchKernelGGL((k_f64_stape<0>), dim3((a.n + 63) / 64), dim3(64), 0, st, a); }
#endif

// ---- the kernel table: one entry per (inputs, jet set); activations tanh / sigmoid / sin inside ----
struct F64Kernel {
    int D, C, NFIRST, NPAIR;
    unsigned D1MASK, HI;
    unsigned long long PAIRS;
    int first_ch[8];
    void (*launch_point)(const F64Args&, bool sin_act, plat_stream);
};
std::deque<F64Kernel>& f64_registry();

#ifdef PINN_EMU
template <class J, int ACT> void run_f64_point(const F64Args& a) { for (int lp = 0; lp < a.npts; ++lp) f64_point<J, ACT>(lp, a); }
inline void launch_f64_dw(const F64Args& a, plat_stream) {
    const int nb = (a.npts + F64_BLOCK - 1) / F64_BLOCK;
    for (int b = 0; b < nb; ++b) for (int e = 0; e < a.nent; ++e) f64_dw_entry(e, b, a);
}
inline void launch_f64_dwt(const F64Args& a, plat_stream) {
    if (a.mode != 0) return;
    const int nb = (a.npts + F64_BLOCK - 1) / F64_BLOCK, nt = f64_num_tiles(a);
    for (int b = 0; b < nb; ++b)
        for (int t = 0; t < nt; ++t) {
            const F64Tile T = f64_tile_locate(a, t);
            double part[F64_TM * F64_TK][64];
            for (int lane = 0; lane < 64; ++lane) {
                double acc[F64_TM * F64_TK];
                f64_dwt_lane(T, b, lane, a, acc);
                for (int e = 0; e < F64_TM * F64_TK; ++e) part[e][lane] = acc[e];
            }
            for (int e = 0; e < F64_TM * F64_TK; ++e) {
                double s = 0.0;
                for (int lane = 0; lane < 64; ++lane) s += part[e][lane];
                f64_dwt_store(T, b, e, s, a);
            }
        }
}
inline void launch_f64_reduce(const F64ReduceArgs& a, plat_stream) { for (int e = 0; e < a.nent; ++e) f64_reduce_entry(e, a); }
#define PINN_LAUNCH_F64(J, ACT, a, st) run_f64_point<J, ACT>(a)
#else
template <class J, int ACT> __global__ void __launch_bounds__(64) k_f64_point(const F64Args a) {
    const int lp = (int)(blockIdx.x * 64 + threadIdx.x);
    if (lp < a.npts) f64_point<J, ACT>(lp, a);
}
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_dw(const F64Args a) {
    const int e = (int)(blockIdx.x * 256 + threadIdx.x);
    if (e < a.nent) f64_dw_entry(e, (int)blockIdx.y, a);
}
template <int UNUSED> __global__ void __launch_bounds__(64) k_f64_dwt(const F64Args a) {
    __shared__ double part[F64_TM * F64_TK][65];                // (65: the column sums below walk a row each, conflict-free)
    const int lane = (int)threadIdx.x;
    const F64Tile T = f64_tile_locate(a, (int)blockIdx.x);
    double acc[F64_TM * F64_TK];
    f64_dwt_lane(T, (int)blockIdx.y, lane, a, acc);
    PINN_UNROLL for (int e = 0; e < F64_TM * F64_TK; ++e) part[e][lane] = acc[e];
    __syncthreads();
    if (lane < F64_TM * F64_TK) {
        double s = 0.0;
        for (int t = 0; t < 64; ++t) s += part[lane][t];
        f64_dwt_store(T, (int)blockIdx.y, lane, s, a);
    }
}
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_reduce(const F64ReduceArgs a) {
    const int e = (int)(blockIdx.x * 256 + threadIdx.x);
    if (e < a.nent) f64_reduce_entry(e, a);
}
inline void launch_f64_dw(const F64Args& a, plat_stream st) {
    const int nb = (a.npts + F64_BLOCK - 1) / F64_BLOCK;
    hipLaunchKernelGGL((k_f64_dw<0>), dim3((a.nent + 255) / 256, nb), dim3(256), 0, st, a);
}
inline void launch_f64_dwt(const F64Args& a, plat_stream st) {
    const int nb = (a.npts + F64_BLOCK - 1) / F64_BLOCK, nt = f64_num_tiles(a);
    if (a.mode != 0 || nt == 0) return;
    hipLaunchKernelGGL((k_f64_dwt<0>), dim3(nt, nb), dim3(64), 0, st, a);
}
inline void launch_f64_reduce(const F64ReduceArgs& a, plat_stream st) {
    hipLaunchKernelGGL((k_f64_reduce<0>), dim3((a.nent + 255) / 256), dim3(256), 0, st, a);
}
#define PINN_LAUNCH_F64(J, ACT, a, st) hipLaunchKernelGGL((k_f64_point<J, ACT>), dim3((a.npts + 63) / 64), dim3(64), 0, st, a)
#endif

// tanh and sigmoid are a RUN-TIME kind per network here (the activation rules branch on it: no matrix pipe to keep fed); sin needs the other
// record convention and is its own instantiation (every network of the equation then uses sin)
template <class J> void launch_f64_point(const F64Args& a, bool sin_act, plat_stream st) {
    (void)st;
    if (sin_act) PINN_LAUNCH_F64(J, ACT_SIN, a, st);
    else PINN_LAUNCH_F64(J, ACT_TANH, a, st);
}
template <int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI> F64Kernel make_f64_kernel() {
    using J = JetSet<D1MASK, PAIRS, NPAIR, HI>;
    static_assert(J::NLAP == 0, "the float64 kernels carry plain derivative channels (no forward-Laplacian channel)");
    F64Kernel k;
    k.D = D; k.C = J::C; k.NFIRST = J::NFIRST; k.NPAIR = J::NPAIR; k.D1MASK = D1MASK; k.HI = HI; k.PAIRS = PAIRS;
    for (int i = 0; i < 8; ++i) {
        k.first_ch[i] = -1;
        for (int kf = 0; kf < J::NFIRST; ++kf) if (J::first_axis(kf) == i) k.first_ch[i] = J::CH_FIRST + kf;
    }
    k.launch_point = &launch_f64_point<J>;
    return k;
}
struct F64Registrar { explicit F64Registrar(const F64Kernel& k) { f64_registry().push_back(k); } };
#define PINN_INSTANTIATE_F64(NAME, D, D1MASK, PAIRS, NPAIR, HI) \
    namespace { pk::F64Registrar NAME##_regf64(pk::make_f64_kernel<D, D1MASK, PAIRS, NPAIR, HI>()); }

}  // namespace pk
