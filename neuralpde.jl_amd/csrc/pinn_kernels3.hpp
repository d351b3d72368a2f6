// pinn_kernels3.hpp — "family 3": the Deep Galerkin (DGM) architecture of the reference (src/dgm.jl:40-48, 97-115), the one non-MLP
// network that routes through PhysicsInformedNN (DeepGalerkin, src/dgm.jl:143-152):
//     S^1     = s1(W^1 x + b^1)
//     Z, G, R = s1(U^{z,g,r} x + W^{z,g,r} S + b^{z,g,r})
//     H       = s2(U^h x + W^h (S . R) + b^h)
//     S'      = (1 - G) . H + Z . S            (L gated layers)
//     f       = W S^{L+1} + b                  (identity output activation)
// with the same exact Taylor-jet derivatives and hand-derived reverse sweep as the MLP kernels: jets travel through the affine maps
// channel by channel, through the activations by jet_forward / jet_adjoint, and through the element-wise PRODUCTS of the gates by the
// Leibniz rule  (a b)_alpha = sum_{beta <= alpha} C(alpha, beta) a_beta b_{alpha - beta}  over the multi-indices of the channel set.
//
// Mapping: DGM nets are small (the reference's examples use 30-50 modes, 3 layers, a few thousand points), so this family is built for
// generality, not for the matrix pipe: one LANE per collocation point, 64 points per wave, every per-point vector (S, gate records, S.R,
// adjoints) in a point-major scratch slab in HBM/L2 ([row][point]: coalesced, a lane only ever touches its own column — no LDS, no
// barriers), weights as wave-uniform scalar loads straight from theta (no packing).  The weight gradients are contractions over points,
//     dW[m][k] = sum_{p, c} dP[m][c][p] In[k][c][p],
// done by a second kernel (k_dgm_dw) per block of points into the block's gradient slab, which the usual fixed-order reduction sums
// (deterministic, no atomics).  Specialised at run time per (modes, layers, inputs, jet set, activations) by csrc/jit.cpp.
#pragma once
#include "pinn_kernels.hpp"

namespace pk {

// ---------------------------------------------------------------------------------------------------------------------
// multi-index arithmetic on the encoded form (nibble 0 = order, nibbles 1.. = sorted axes) and the Leibniz coefficients
// ---------------------------------------------------------------------------------------------------------------------
HD constexpr int mi_order(unsigned m) { return (int)(m & 0xFu); }
HD constexpr int mi_axis(unsigned m, int i) { return (int)((m >> (4 * (i + 1))) & 0xFu); }
HD constexpr int mi_count(unsigned m, int axis) {
    int n = 0;
    for (int i = 0; i < mi_order(m); ++i) n += mi_axis(m, i) == axis;
    return n;
}
HD constexpr bool mi_leq(unsigned b, unsigned a) {          // b is a sub-multi-index of a
    for (int ax = 0; ax < 8; ++ax)
        if (mi_count(b, ax) > mi_count(a, ax)) return false;
    return true;
}
HD constexpr unsigned mi_minus(unsigned a, unsigned b) {    // a - b (b <= a), sorted
    unsigned out = 0;
    int n = 0;
    for (int ax = 0; ax < 8; ++ax)
        for (int r = 0; r < mi_count(a, ax) - mi_count(b, ax); ++r) { out |= (unsigned)ax << (4 * (n + 1)); ++n; }
    return out | (unsigned)n;
}
HD constexpr int binom_i(int n, int k) { int r = 1; for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i; return r; }
HD constexpr int mi_binom(unsigned a, unsigned b) {         // number of ways the positions of a split into b and a - b
    int r = 1;
    for (int ax = 0; ax < 8; ++ax) r *= binom_i(mi_count(a, ax), mi_count(b, ax));
    return r;
}
template <class J> HD constexpr int chan_find(unsigned mi) {
    for (int c = 0; c < J::C; ++c)
        if (J::channel_mi(c) == mi) return c;
    return -1;
}
// y = a . b on jets (all channels); the channel set is closed under sub-multi-indices by construction
template <class J> DEV void jet_mul(const vfloat (&a)[J::C], const vfloat (&b)[J::C], vfloat (&y)[J::C]) {
    PINN_UNROLL for (int al = 0; al < J::C; ++al) {
        vfloat s = vfloat(0.f);
        PINN_UNROLL for (int be = 0; be < J::C; ++be) {
            if (!mi_leq(J::channel_mi(be), J::channel_mi(al))) continue;
            const int ga = chan_find<J>(mi_minus(J::channel_mi(al), J::channel_mi(be)));
            if (ga < 0) continue;                               // cannot happen for a closed set
            s = vfma(vfloat((float)mi_binom(J::channel_mi(al), J::channel_mi(be))) * a[be], b[ga], s);
        }
        y[al] = s;
    }
}
// da += adjoint of y = a . b with respect to a, given dy and b:  da_beta += sum_{alpha >= beta} C(alpha, beta) dy_alpha b_{alpha - beta}
template <class J> DEV void jet_mul_adj(const vfloat (&dy)[J::C], const vfloat (&b)[J::C], vfloat (&da)[J::C], float sign = 1.0f) {
    PINN_UNROLL for (int be = 0; be < J::C; ++be) {
        vfloat s = vfloat(0.f);
        PINN_UNROLL for (int al = 0; al < J::C; ++al) {
            if (!mi_leq(J::channel_mi(be), J::channel_mi(al))) continue;
            const int ga = chan_find<J>(mi_minus(J::channel_mi(al), J::channel_mi(be)));
            if (ga < 0) continue;
            s = vfma(vfloat((float)mi_binom(J::channel_mi(al), J::channel_mi(be))) * dy[al], b[ga], s);
        }
        da[be] = vfma(vfloat(sign), s, da[be]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int MP_, int L_, int D_, unsigned D1MASK_, unsigned long long PAIRS_, int NPAIR_, unsigned HI_, int ACT1_, int ACT2_>
struct Spec3 {
    using J = JetSet<D1MASK_, PAIRS_, NPAIR_, HI_>;
    static_assert(J::NLAP == 0, "the DGM kernels carry multi-index channels only (no forward-Laplacian channel)");
    static constexpr int FAMILY = 3;
    static constexpr int MP = MP_, L = L_, D = D_, C = J::C, ACT1 = ACT1_, ACT2 = ACT2_;
    static constexpr unsigned D1MASK = D1MASK_, HI = HI_;
    static constexpr unsigned long long PAIRS = PAIRS_;
    static constexpr int NPAIR = NPAIR_, NFIRST = J::NFIRST, PG = 4, TP = 64, NG = C;
    static constexpr int MB = MP_ < 16 ? MP_ : 16;                     // neurons per register block of the affine maps
    static constexpr int MC = MP_ * C;                                 // rows of one per-point vector
    // scratch rows (each row = one float per point of the launch group)
    static constexpr int R_S = 0;                                      // (L+1) x MC : S^1 .. S^{L+1}
    static constexpr int R_REC1 = R_S + (L_ + 1) * MC;                 // record of the first dense layer
    static constexpr int R_REC = R_REC1 + MC;                          // [l][q = z,g,r,h] records of the gates
    static constexpr int R_SR = R_REC + L_ * 4 * MC;                   // [l] S . R
    static constexpr int R_DP1 = R_SR + L_ * MC;                       // adjoint of the first layer's pre-activation jets
    static constexpr int R_DP = R_DP1 + MC;                            // [l][q] adjoints of the gates' pre-activation jets
    static constexpr int R_DPO = R_DP + L_ * 4 * MC;                   // C rows: seeds d loss / d u-jets (output layer)
    static constexpr int R_DS = R_DPO + C;                             // 2 x MC ping-pong: adjoint of S
    static constexpr int ROWS = R_DS + 2 * MC;
};

// theta layout of the reference's DGM chain ([3P] Lux NamedTuple order, ComponentArrays flattening; checked numerically by the Julia
// glue's verify_layout): Dense(d -> M): W1 (M x d, column-major), b1 | per gated layer: Uz Ug Ur Uh (M x d each), Wz Wg Wr Wh (M x M each),
// bz bg br bh | Dense(M -> 1): WL (1 x M), bL.   M = the real number of modes (<= MP).
struct DgmLayout {
    int M, d;
    HD int w1() const { return 0; }
    HD int b1() const { return M * d; }
    HD int layer(int l) const { return M * d + M + l * (4 * M * d + 4 * M * M + 4 * M); }
    HD int U(int l, int q) const { return layer(l) + q * M * d; }
    HD int W(int l, int q) const { return layer(l) + 4 * M * d + q * M * M; }
    HD int b(int l, int q) const { return layer(l) + 4 * M * d + 4 * M * M + q * M; }
    HD int wL(int L) const { return layer(L); }
    HD int bL(int L) const { return layer(L) + M; }
    HD int total(int L) const { return layer(L) + M + 1; }
};

// post-activation jets of one neuron from its record (rec[0] = activation value, or z for sin; rec[c > 0] = pre-activation channels)
template <class J, int ACTK>
DEV void jets_from_record(const vfloat (&rec)[J::C], vfloat (&out)[J::C]) {
    constexpr bool SINACT = (ACTK == ACT_SIN);
    vfloat dd[ND];
    PINN_UNROLL for (int c = 0; c < J::C; ++c) out[c] = rec[c];
    act_derivs_n<J::NORD - 1, SINACT>(ACTK, rec[0], dd);
    jet_forward<J>(out, dd);
    out[0] = act_from_record<SINACT>(rec[0]);
}

template <class S, int MODE>
DEV void wave_dgm(const GroupArgs& ga, int blk, int nblocks) {
    using J = typename S::J;
    constexpr int MP = S::MP, L = S::L, D = S::D, C = S::C, MB = S::MB, MC = S::MC, NFIRST = S::NFIRST;
    constexpr bool SIN1 = (S::ACT1 == ACT_SIN), SIN2 = (S::ACT2 == ACT_SIN);
    constexpr bool BWD = (MODE == MODE_FUSED);
    const vint lane = lane_id();
    const float* th = ga.packed;                  // this network's parameters inside theta (unpacked)
    const int M = ga.dgm_modes;                   // real number of modes (<= MP)
    const DgmLayout LO{M, D};
    const int NPAD = ga.dgm_npad;                 // points per scratch row
    float* SC = ga.scratch;
    const vbool l0 = veq(lane, 0);
    // weight (out m, in k) of an M x K column-major block, zero outside the real sizes (wave-uniform scalar load)
    auto wat = [&](int off, int m, int k, int K) -> float { return (m < M && k < K) ? th[off + m + k * M] : 0.f; };
    auto bat = [&](int off, int m) -> float { return m < M ? th[off + m] : 0.f; };

    vfloat pbar[MAX_PARAMS];
    PINN_UNROLL for (int j = 0; j < MAX_PARAMS; ++j) pbar[j] = vfloat(0.f);
    vdacc lsum = vdacc_zero();
    int cur_term = -1;
    constexpr bool SUMS = (MODE == MODE_FUSED || MODE == MODE_LOSS);      // modes that deliver the per-term sums of squares
    if (SUMS)
        for (int j = 0; j < ga.nterms; ++j) ga.losspart[(size_t)blk * ga.nterms_total + ga.terms[j].term_id] = 0.0;

    const int niter = (ga.ntiles + nblocks - 1) / nblocks;
    for (int it = 0; it < niter; ++it) {
        const int tix = it * nblocks + blk;
        if (tix >= ga.ntiles) break;
        int kt = 0;
        for (int j = 1; j < ga.nterms; ++j)
            if (tix >= ga.terms[j].tile0) kt = j;
        if (kt != cur_term) {
            if (cur_term >= 0 && SUMS)
                ga.losspart[(size_t)blk * ga.nterms_total + ga.terms[cur_term].term_id] = wave_sum_dd(lsum, vlt(lane, 64));
            lsum = vdacc_zero();
            cur_term = kt;
        }
        const TermDev& T = ga.terms[kt];
        const vint p = vint((tix - T.tile0) * 64) + lane;           // point inside the term
        const vbool valid = vlt(p, T.N);
        const vint col = vint(tix * 64) + lane;                      // column of the scratch rows
        auto LD = [&](int row) -> vfloat { return gload(SC, vint(row) * NPAD + col); };
        auto ST = [&](int row, vfloat v) { gstore(SC, vint(row) * NPAD + col, v); };

        vfloat x[D];
        PINN_UNROLL for (int i = 0; i < D; ++i) x[i] = gload_masked(T.pts, p * T.dt + vint(T.imap[i]), valid);
        // jets of input i: value x_i, first derivative along its own axis 1, everything else 0
        auto xjet = [&](int i, int c) -> vfloat {
            if (c == 0) return x[i];
            return (c >= J::CH_FIRST && c < J::CH_FIRST + NFIRST && J::first_axis(c - J::CH_FIRST) == i) ? vfloat(1.0f) : vfloat(0.f);
        };
        // acc[m][c] (m in a block of MB neurons starting at m0) = bias + U x-jets + W In-jets, the affine map of one gate
        auto affine = [&](vfloat (&acc)[MB][C], int m0, int offW, int offU, int offb, int rowIn, int Kin) {
            PINN_UNROLL for (int mm = 0; mm < MB; ++mm) {
                PINN_UNROLL for (int c = 0; c < C; ++c) acc[mm][c] = vfloat(0.f);
                acc[mm][0] = vfloat(bat(offb, m0 + mm));
                PINN_UNROLL for (int i = 0; i < D; ++i) {
                    const float u = wat(offU, m0 + mm, i, D);
                    PINN_UNROLL for (int c = 0; c < C; ++c) acc[mm][c] = vfma(vfloat(u), xjet(i, c), acc[mm][c]);
                }
            }
            if (rowIn >= 0)
                for (int k = 0; k < Kin; ++k) {
                    vfloat s[C];
                    PINN_UNROLL for (int c = 0; c < C; ++c) s[c] = LD(rowIn + k * C + c);
                    PINN_UNROLL for (int mm = 0; mm < MB; ++mm) {
                        const float w = wat(offW, m0 + mm, k, M);
                        PINN_UNROLL for (int c = 0; c < C; ++c) acc[mm][c] = vfma(vfloat(w), s[c], acc[mm][c]);
                    }
                }
        };
        // activation of one neuron's pre-activation jets: writes its record, returns the post-activation jets
        auto activate = [&](const vfloat (&pre)[C], int act, bool sinact, int rowRec, int m, vfloat (&post)[C]) {
            vfloat rec[C];
            PINN_UNROLL for (int c = 0; c < C; ++c) rec[c] = pre[c];
            const vfloat a = sinact ? act_value<true>(act, pre[0]) : act_value<false>(act, pre[0]);
            rec[0] = sinact ? pre[0] : a;
            PINN_UNROLL for (int c = 0; c < C; ++c) ST(rowRec + m * C + c, rec[c]);
            vfloat dd[ND];
            PINN_UNROLL for (int c = 0; c < C; ++c) post[c] = rec[c];
            if (sinact) act_derivs_n<J::NORD - 1, true>(act, rec[0], dd); else act_derivs_n<J::NORD - 1, false>(act, rec[0], dd);
            jet_forward<J>(post, dd);
            post[0] = a;
        };

        // =========================== forward ===========================
        for (int m0 = 0; m0 < MP; m0 += MB) {                         // S^1 = s1(W1 x + b1)
            vfloat acc[MB][C];
            affine(acc, m0, 0, LO.w1(), LO.b1(), -1, 0);
            PINN_UNROLL for (int mm = 0; mm < MB; ++mm) {
                vfloat post[C];
                activate(acc[mm], S::ACT1, SIN1, S::R_REC1, m0 + mm, post);
                PINN_UNROLL for (int c = 0; c < C; ++c) ST(S::R_S + (m0 + mm) * C + c, (m0 + mm) < M ? post[c] : vfloat(0.f));
            }
        }
        for (int l = 0; l < L; ++l) {
            const int rS = S::R_S + l * MC, rS2 = S::R_S + (l + 1) * MC, rSR = S::R_SR + l * MC;
            auto rREC = [&](int q) { return S::R_REC + (l * 4 + q) * MC; };
            for (int q = 0; q < 3; ++q)                               // Z, G, R = s1(U x + W S + b)
                for (int m0 = 0; m0 < MP; m0 += MB) {
                    vfloat acc[MB][C];
                    affine(acc, m0, LO.W(l, q), LO.U(l, q), LO.b(l, q), rS, M);
                    PINN_UNROLL for (int mm = 0; mm < MB; ++mm) {
                        vfloat post[C];
                        activate(acc[mm], S::ACT1, SIN1, rREC(q), m0 + mm, post);
                        if (q == 2) {                                 // S . R
                            vfloat s[C], sr[C];
                            PINN_UNROLL for (int c = 0; c < C; ++c) s[c] = LD(rS + (m0 + mm) * C + c);
                            jet_mul<J>(s, post, sr);
                            PINN_UNROLL for (int c = 0; c < C; ++c) ST(rSR + (m0 + mm) * C + c, sr[c]);
                        }
                    }
                }
            for (int m0 = 0; m0 < MP; m0 += MB) {                     // H = s2(U x + W (S . R) + b);  S' = H - G . H + Z . S
                vfloat acc[MB][C];
                affine(acc, m0, LO.W(l, 3), LO.U(l, 3), LO.b(l, 3), rSR, M);
                PINN_UNROLL for (int mm = 0; mm < MB; ++mm) {
                    const int m = m0 + mm;
                    vfloat h[C], z[C], g[C], s[C], rz[C], rg[C], t1[C], t2[C];
                    activate(acc[mm], S::ACT2, SIN2, rREC(3), m, h);
                    PINN_UNROLL for (int c = 0; c < C; ++c) { rz[c] = LD(rREC(0) + m * C + c); rg[c] = LD(rREC(1) + m * C + c); s[c] = LD(rS + m * C + c); }
                    jets_from_record<J, S::ACT1>(rz, z);
                    jets_from_record<J, S::ACT1>(rg, g);
                    jet_mul<J>(g, h, t1);
                    jet_mul<J>(z, s, t2);
                    PINN_UNROLL for (int c = 0; c < C; ++c) ST(rS2 + m * C + c, m < M ? h[c] - t1[c] + t2[c] : vfloat(0.f));
                }
            }
        }
        // output layer (identity): u-jets
        vfloat U[C];
        PINN_UNROLL for (int c = 0; c < C; ++c) U[c] = vfloat(0.f);
        U[0] = vfloat(th[LO.bL(L)]);
        for (int m = 0; m < M; ++m) {
            const float w = th[LO.wL(L) + m];
            PINN_UNROLL for (int c = 0; c < C; ++c) U[c] = vfma(vfloat(w), LD(S::R_S + L * MC + m * C + c), U[c]);
        }
        if (MODE == MODE_FWD) {
            PINN_UNROLL for (int c = 0; c < C; ++c) gstore_masked(T.out, vint(c * T.N) + p, U[c], valid);
            continue;
        }

        // =========================== residual tape (one lane per point) ===========================
        const int NP = ga.nparams, DT = T.dt, R0 = DT + NP + C + T.nsrc;
        const rp::Instr* prog = ga.prog + T.prog_off;
        vtape tv;
        tape_zero(tv);
        if (!T.hetero) { PINN_UNROLL for (int i = 0; i < D; ++i) tape_set(tv, i, x[i]); }
        else for (int j = 0; j < DT; ++j) tape_set(tv, j, gload_masked(T.pts, p * DT + vint(j), valid));
        for (int j = 0; j < NP; ++j) tape_set(tv, DT + j, vfloat(ga.params[j]));
        PINN_UNROLL for (int c = 0; c < C; ++c) tape_set(tv, DT + NP + c, U[c]);
        for (int j = 0; j < T.nsrc; ++j) tape_set(tv, DT + NP + C + j, gload_masked(T.src, vint(j * T.N) + p, valid));
        for (int q = 0; q < T.nops; ++q) {
            const rp::Instr ins = rp::fetch_uniform(prog, q);
            const vfloat va = tape_get(tv, ins.a), vb = tape_get(tv, ins.b);
            tape_set(tv, R0 + q, rp::is_bilinear(ins.code) ? rp::apply_bilinear<vfloat>(ins, va, vb) : rp::apply<vfloat>(ins.code, va, vb, ins.imm));
        }
        const vfloat r = tape_get(tv, T.out_row);
        if (MODE == MODE_RESID) { gstore_masked(T.out, p, r, valid); continue; }
        vfloat sw = vfloat(1.0f);
        if (T.pw) sw = gload_masked(T.pw, p, valid);
        const vfloat rm = vselect(valid, r * sw, vfloat(0.f));
        lsum = vdacc_fma(rm, rm, lsum);
        if (MODE == MODE_LOSS) continue;
        const vfloat rbar = rm * vfloat(T.scale) * sw;
        vtape ta;
        tape_zero(ta);
        tape_set(ta, T.out_row, vfloat(1.0f));
        for (int q = T.nops - 1; q >= 0; --q) {
            const rp::Instr ins = rp::fetch_uniform(prog, q);
            const vfloat va = tape_get(tv, ins.a), vb = tape_get(tv, ins.b), gq = tape_get(ta, R0 + q);
            vfloat da, db;
            if (rp::is_bilinear(ins.code)) rp::adjoint_bilinear<vfloat>(ins, va, vb, gq, da, db);
            else rp::adjoint<vfloat>(ins.code, va, vb, tape_get(tv, R0 + q), ins.imm, gq, da, db);
            tape_set(ta, ins.a, tape_get(ta, ins.a) + da);
            tape_set(ta, ins.b, tape_get(ta, ins.b) + db);
        }
        vfloat ubar[C];
        PINN_UNROLL for (int c = 0; c < C; ++c) {
            ubar[c] = vselect(valid, rbar * tape_get(ta, DT + NP + c), vfloat(0.f));
            ST(S::R_DPO + c, ubar[c]);
        }
        for (int j = 0; j < ga.nparams_estim; ++j) {
            const vfloat pj = vselect(valid, rbar * tape_get(ta, DT + j), vfloat(0.f));
            PINN_UNROLL for (int jj = 0; jj < MAX_PARAMS; ++jj) if (jj == j) pbar[jj] += pj;
        }

        // =========================== reverse sweep ===========================
        int cur = 0;
        for (int m = 0; m < MP; ++m) {
            const float w = m < M ? th[LO.wL(L) + m] : 0.f;
            PINN_UNROLL for (int c = 0; c < C; ++c) ST(S::R_DS + cur * MC + m * C + c, vfloat(w) * ubar[c]);
        }
        for (int l = L - 1; l >= 0; --l) {
            const int rS = S::R_S + l * MC, rDin = S::R_DS + cur * MC, rDout = S::R_DS + (1 - cur) * MC;
            auto rREC = [&](int q) { return S::R_REC + (l * 4 + q) * MC; };
            auto rDP = [&](int q) { return S::R_DP + (l * 4 + q) * MC; };
            // element-wise part: S' = H - G . H + Z . S
            for (int m = 0; m < MP; ++m) {
                vfloat ds[C], s[C], rz[C], rg[C], rh[C], z[C], g[C], h[C];
                PINN_UNROLL for (int c = 0; c < C; ++c) {
                    ds[c] = LD(rDin + m * C + c); s[c] = LD(rS + m * C + c);
                    rz[c] = LD(rREC(0) + m * C + c); rg[c] = LD(rREC(1) + m * C + c); rh[c] = LD(rREC(3) + m * C + c);
                }
                jets_from_record<J, S::ACT1>(rz, z);
                jets_from_record<J, S::ACT1>(rg, g);
                jets_from_record<J, S::ACT2>(rh, h);
                vfloat dh[C], dg[C], dz[C], dsd[C];
                PINN_UNROLL for (int c = 0; c < C; ++c) { dh[c] = ds[c]; dg[c] = vfloat(0.f); dz[c] = vfloat(0.f); dsd[c] = vfloat(0.f); }
                jet_mul_adj<J>(ds, g, dh, -1.0f);                     // d/dH of -G.H
                jet_mul_adj<J>(ds, h, dg, -1.0f);                     // d/dG of -G.H
                jet_mul_adj<J>(ds, s, dz);                            // d/dZ of Z.S
                jet_mul_adj<J>(ds, z, dsd);                           // d/dS of Z.S
                vfloat dd[ND];
                act_derivs_n<J::NORD, SIN2>(S::ACT2, rh[0], dd); jet_adjoint<J>(dh, rh, dd);
                act_derivs_n<J::NORD, SIN1>(S::ACT1, rg[0], dd); jet_adjoint<J>(dg, rg, dd);
                act_derivs_n<J::NORD, SIN1>(S::ACT1, rz[0], dd); jet_adjoint<J>(dz, rz, dd);
                PINN_UNROLL for (int c = 0; c < C; ++c) {
                    const bool on = m < M;
                    ST(rDP(3) + m * C + c, on ? dh[c] : vfloat(0.f)); ST(rDP(1) + m * C + c, on ? dg[c] : vfloat(0.f));
                    ST(rDP(0) + m * C + c, on ? dz[c] : vfloat(0.f)); ST(rDout + m * C + c, on ? dsd[c] : vfloat(0.f));
                }
            }
            // d(S.R) = Wh^T dP_h;  dS += d(S.R) . R;  dR = d(S.R) . S  ->  dP_r
            for (int k0 = 0; k0 < MP; k0 += MB) {
                vfloat acc[MB][C];
                PINN_UNROLL for (int kk = 0; kk < MB; ++kk) PINN_UNROLL for (int c = 0; c < C; ++c) acc[kk][c] = vfloat(0.f);
                for (int m = 0; m < M; ++m) {
                    vfloat dp[C];
                    PINN_UNROLL for (int c = 0; c < C; ++c) dp[c] = LD(rDP(3) + m * C + c);
                    PINN_UNROLL for (int kk = 0; kk < MB; ++kk) {
                        const float w = wat(LO.W(l, 3), m, k0 + kk, M);
                        PINN_UNROLL for (int c = 0; c < C; ++c) acc[kk][c] = vfma(vfloat(w), dp[c], acc[kk][c]);
                    }
                }
                PINN_UNROLL for (int kk = 0; kk < MB; ++kk) {
                    const int k = k0 + kk;
                    vfloat s[C], rr[C], rj[C], dso[C], dr[C];
                    PINN_UNROLL for (int c = 0; c < C; ++c) { s[c] = LD(rS + k * C + c); rr[c] = LD(rREC(2) + k * C + c); dso[c] = LD(rDout + k * C + c); dr[c] = vfloat(0.f); }
                    jets_from_record<J, S::ACT1>(rr, rj);
                    jet_mul_adj<J>(acc[kk], rj, dso);
                    jet_mul_adj<J>(acc[kk], s, dr);
                    vfloat dd[ND];
                    act_derivs_n<J::NORD, SIN1>(S::ACT1, rr[0], dd); jet_adjoint<J>(dr, rr, dd);
                    PINN_UNROLL for (int c = 0; c < C; ++c) { ST(rDout + k * C + c, k < M ? dso[c] : vfloat(0.f)); ST(rDP(2) + k * C + c, k < M ? dr[c] : vfloat(0.f)); }
                }
            }
            // dS += W^T dP for the gates that read S directly
            for (int q = 0; q < 3; ++q)
                for (int k0 = 0; k0 < MP; k0 += MB) {
                    vfloat acc[MB][C];
                    PINN_UNROLL for (int kk = 0; kk < MB; ++kk) PINN_UNROLL for (int c = 0; c < C; ++c) acc[kk][c] = vfloat(0.f);
                    for (int m = 0; m < M; ++m) {
                        vfloat dp[C];
                        PINN_UNROLL for (int c = 0; c < C; ++c) dp[c] = LD(rDP(q) + m * C + c);
                        PINN_UNROLL for (int kk = 0; kk < MB; ++kk) {
                            const float w = wat(LO.W(l, q), m, k0 + kk, M);
                            PINN_UNROLL for (int c = 0; c < C; ++c) acc[kk][c] = vfma(vfloat(w), dp[c], acc[kk][c]);
                        }
                    }
                    PINN_UNROLL for (int kk = 0; kk < MB; ++kk)
                        PINN_UNROLL for (int c = 0; c < C; ++c) ST(rDout + (k0 + kk) * C + c, LD(rDout + (k0 + kk) * C + c) + acc[kk][c]);
                }
            cur = 1 - cur;
        }
        for (int m = 0; m < MP; ++m) {                               // first layer: dP_1 = activation adjoint of dS^1
            vfloat ds[C], rec[C], dd[ND];
            PINN_UNROLL for (int c = 0; c < C; ++c) { ds[c] = LD(S::R_DS + cur * MC + m * C + c); rec[c] = LD(S::R_REC1 + m * C + c); }
            act_derivs_n<J::NORD, SIN1>(S::ACT1, rec[0], dd);
            jet_adjoint<J>(ds, rec, dd);
            PINN_UNROLL for (int c = 0; c < C; ++c) ST(S::R_DP1 + m * C + c, m < M ? ds[c] : vfloat(0.f));
        }
    }
    if (!SUMS) return;
    if (cur_term >= 0) ga.losspart[(size_t)blk * ga.nterms_total + ga.terms[cur_term].term_id] = wave_sum_dd(lsum, vlt(lane, 64));
    if (MODE != MODE_FUSED) return;
    // PDE-parameter gradients of this block (the weight gradients come from k_dgm_dw)
    float* slab = ga.slabs + (size_t)blk * ga.dgm_slab;
    PINN_UNROLL for (int j = 0; j < MAX_PARAMS; ++j)
        gstore_masked(slab + ga.dgm_nparams, vint(j) + (lane & vint(0)), vfloat((float)wave_sum_d(pbar[j], vlt(lane, 64))), l0);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradients: block b's slab entry e = sum over the points of block b's tiles (fixed order).  One thread per theta entry.
// ---------------------------------------------------------------------------------------------------------------------
struct DgmDwArgs {
    const float* scratch; float* slabs;
    int npad, slab, nblocks, ntiles;
    int M, MPad, d, L, C, nparams;
    int r_s, r_rec1, r_sr, r_dp1, r_dp, r_dpo;          // scratch row bases (Spec3::R_*)
    int first_ch[8];                                    // channel of d/dx_i (or -1)
    int nterms;
    TermDev terms[MAX_GROUP_TERMS];
};
HD void dgm_dw_entry(int e, int b, const DgmDwArgs& a) {
    const DgmLayout LO{a.M, a.d};
    const int MC = a.MPad * a.C;
    // decode the theta entry: which affine map, output neuron m, and input (k of In / coordinate i / bias)
    int rowDP, rowIn = -1, m, k = -1, xi = -1, is_bias = 0, Cdp = a.C;
    auto decode = [&](int off_U, int off_W, int off_b, int rdp, int rin, int e_) -> bool {
        const int M = a.M, d = a.d;
        if (off_U >= 0 && e_ >= off_U && e_ < off_U + M * d) { m = (e_ - off_U) % M; xi = (e_ - off_U) / M; rowDP = rdp; return true; }
        if (off_W >= 0 && e_ >= off_W && e_ < off_W + M * M) { m = (e_ - off_W) % M; k = (e_ - off_W) / M; rowDP = rdp; rowIn = rin; return true; }
        if (e_ >= off_b && e_ < off_b + M) { m = e_ - off_b; is_bias = 1; rowDP = rdp; return true; }
        return false;
    };
    bool found = decode(LO.w1(), -1, LO.b1(), a.r_dp1, -1, e);
    for (int l = 0; l < a.L && !found; ++l)
        for (int q = 0; q < 4 && !found; ++q)
            found = decode(LO.U(l, q), LO.W(l, q), LO.b(l, q), a.r_dp + (l * 4 + q) * MC, q == 3 ? a.r_sr + l * MC : a.r_s + l * MC, e);
    int out_layer = 0;
    if (!found) {                                        // output layer: WL (1 x M) then bL; dP = the seeds (C rows, no neuron index)
        out_layer = 1;
        if (e >= LO.wL(a.L) && e < LO.wL(a.L) + a.M) { k = e - LO.wL(a.L); rowIn = a.r_s + a.L * MC; }
        else is_bias = 1;
        rowDP = a.r_dpo; m = 0;
    }
    (void)Cdp;
    double s = 0.0;
    for (int tix = b; tix < a.ntiles; tix += a.nblocks) {
        int kt = 0;
        for (int j = 1; j < a.nterms; ++j) if (tix >= a.terms[j].tile0) kt = j;
        const TermDev& T = a.terms[kt];
        for (int ln = 0; ln < 64; ++ln) {
            const int p = (tix - T.tile0) * 64 + ln;
            if (p >= T.N) break;
            const size_t col = (size_t)tix * 64 + ln;
            auto DP = [&](int c) -> float { return a.scratch[(size_t)(rowDP + (out_layer ? 0 : m * a.C) + c) * a.npad + col]; };
            if (is_bias) s += (double)DP(0);
            else if (xi >= 0) {
                float v = DP(0) * T.pts[(size_t)p * T.dt + T.imap[xi]];
                if (a.first_ch[xi] >= 0) v += DP(a.first_ch[xi]);
                s += (double)v;
            } else {
                float v = 0.f;
                for (int c = 0; c < a.C; ++c) v += DP(c) * a.scratch[(size_t)(rowIn + k * a.C + c) * a.npad + col];
                s += (double)v;
            }
        }
    }
    a.slabs[(size_t)b * a.slab + e] = (float)s;
}

}  // namespace pk
