// pinn_kernels6.hpp — "family 4s": the FLOAT64 evaluation on the matrix pipe for WIDE layers and LARGE jet sets (r06).
//
// Family 4m (pinn_kernels5.hpp) keeps a tile's activations and pre-activations of ALL jet channels in the wave's registers from the first
// layer to the last: 2 x 4 HT x C doubles per lane, which ends at HT x C <= 24 (64-wide nets with C <= 6).  BASELINE configs 4 and 5
// (128-wide nets, the 4-D jet set {u, u_t, u_x, u_y, u_z, u_xx, u_yy, u_zz}: HT x C = 40 ... 72) and the 3-D Hessian sets therefore stayed on
// the lane-per-point kernels — 261 x the fp32 path on cfg5 (profiles/r05_f64_kernel_stats.txt section 3).  This family puts them on
// v_mfma_f64_16x16x4_f64 by SLICING the channel dimension:
//   * one wave per tile of 16 points; a layer's GEMM runs in passes over channel groups of CG = 24 / HT channels (accumulators 4 HT x CG = 96
//     doubles per lane); the B operand (the layer below's post-activation jets / the layer above's dZ) comes from the scratch rows, the result
//     (pre-activation jets / G = W^T dZ) goes to the scratch rows of the layer — the rows family 4 and 4m use, in 4m's point-block-major layout
//     (f64m_six), so the hidden-to-hidden dW kernel and the slab reduction are shared;
//   * the f64 MFMA layouts are SELF-FEEDING in memory as they are in registers (pinn_kernels5.hpp): lane (q, j) owns neuron rows q + 4 r of
//     point j in the C/D layout AND reads exactly those elements as the B operand (k = lane >> 4) of the next GEMM — every lane reads back only
//     what it wrote itself: no barriers, no cross-lane memory dependencies, wave-private tiles;
//   * between the GEMM passes an element-wise pass per layer applies the activation / jet rules (templates of pinn_kernels.hpp with V = double)
//     to one element at a time (C loads, C stores): register use is independent of C and of the width (<= 256: two waves per SIMD cover each
//     other's memory latency);
//   * the networks' outputs and seeds (U) live in LDS (nnets x C x 16 doubles per wave), the residual tape is family 4's;
//   * first / last layer, PDE-parameter and squared-residual sums per tile into F64Args::tpart as in family 4m.
// Weight fragments (A operands) stream from theta through L2: one fragment per (output tile, k-block) and channel group.
// The weight-gradient kernel for HT = 8 splits the OUTPUT tiles across the four waves of a workgroup (HT^2 x 4 accumulators do not fit one wave):
// every wave walks all points of the 512-point block, no LDS combine.
#pragma once
#include "pinn_kernels5.hpp"
#ifndef PINN_F64S_ZMAX
#define PINN_F64S_ZMAX 24               // channel group size CG = ZMAX / HT: the GEMM accumulators are 4 HT x CG = 4 ZMAX doubles per lane (A/B)
#endif
#ifndef PINN_F64S_DWT_SYNC
#define PINN_F64S_DWT_SYNC 0            // 1: workgroup barrier per step of the output-split dW kernel (keeps the four waves on the same input rows) — measured SLOWER (cfg5 26.0 -> 27.0 ms, cfg4 17.4 -> 18.2 ms): off
#endif
#ifndef PINN_F64S_WAVES
#define PINN_F64S_WAVES 1               // waves per SIMD the sliced tile kernel is compiled for (register cap 512 / waves): at 1 the GEMM loop is operand loads and MFMAs only; at 2 (256 registers, 192 of them accumulators) it spills inside the loop (A/B: profiles/r06_f64_sliced.txt)
#endif
namespace pk {
// what the sliced kernels may read past a layer's weight matrix (theta) / past a layer's scratch rows when the layer is narrower than 16 HT: in doubles
constexpr size_t F64S_PAD_THETA = 128 * 128;
constexpr size_t F64S_PAD_SCRATCH = (size_t)128 * 24 * 16;
}

namespace pk {

// ---- kernel A'': one tile of 16 points through every network of the term, sliced over channel groups ----
template <class J, int HT, int CG, int ACTK>
DEV void f64s_tile(int tile, const F64Args& a, double* ulds /* [F64_MAX_NETS][C][16] */) {
    constexpr int C = J::C, NR = HT * 4;
    constexpr int TB = (C <= 2) ? 8 : ((C <= 6) ? 4 : 2);       // tile rows per batch of the element-wise passes (loads in flight: TB x C, reverse: 2 TB x C)
    constexpr bool SIN = (ACTK == ACT_SIN);
    const int pbase = tile * 16;
    double* S = a.scratch;
    double* TP = a.tpart + (size_t)tile * (size_t)a.ntp;
    const bool rev = a.mode == 0;
    // one GEMM pass: Z[rows of the layer `n_out` wide][group g0 .. g0 + CG) = A (n_out x n_in, element (m, k) at Aw[m * sm + k * sk]) x B rows
    // `brow` (n_in neurons x C channels), (+ bias on channel 0), stored to rows `zrow`
    auto gemm = [&](const double* Aw, int sm, int sk, int n_out, int n_in, int brow, int zrow, const double* bias, int ce) {
        for (int g0 = 0; g0 < ce; g0 += CG) {
            LVd<NR * CG> Z;
            PINN_LANES(l) { PINN_UNROLL for (int e = 0; e < NR * CG; ++e) Z(l, e) = 0.0; }
            // operand ring: the fragments of k-block kb + PD are requested before the MFMAs of k-block kb (HT x CG MFMAs of 64 cycles each: one
            // k-block ahead covers an L2 round trip at CG = 3; value-only terms — 8 MFMAs per k-block — look three ahead)
            constexpr int PD = (CG >= 3) ? 1 : ((CG == 2) ? 2 : 3);
            LVd<HT> Af[PD + 1];
            LVd<CG> Bf[PD + 1];
            // UNCONDITIONAL, AFFINE loads: per-lane base + (compile-time multiples of) uniform strides.  Nothing is clamped or predicated — a clamp per
            // element makes every address its own value (the compiler then precomputes all 256 of a GEMM, spills them, and waits for memory in front
            // of each load: measured) — so a layer narrower than the kernel's 16 HT reads past its matrix / its scratch rows: theta and the scratch are
            // allocated with that margin (f64.cpp: F64S_PAD), what is read there meets B rows that a select has zeroed (k >= n_in) or lands in rows of
            // Z that are never stored (m >= n_out)
            auto load = [&](int kb, LVd<HT>& A_, LVd<CG>& B_) {
                PINN_LANES(l) {
                    const double* pa = Aw + ((l & 15) * sm + (l >> 4) * sk);
                    PINN_UNROLL for (int t = 0; t < HT; ++t) A_(l, t) = pa[16 * t * sm + 4 * kb * sk];
                    const bool kin = 4 * kb + (l >> 4) < n_in;
                    const double* pb = S + f64m_six(a, (size_t)brow + (size_t)((l >> 4) * C + g0), pbase + (l & 15));
                    PINN_UNROLL for (int g = 0; g < CG; ++g) {
                        const double v = pb[(4 * kb * C + g) * 16];
                        B_(l, g) = kin ? v : 0.0;            // (a column group past the channel prefix multiplies whatever lies there: its Z columns are never stored)
                    }
                }
            };
            // (no early exits on the layer's true width: operands beyond it are loaded as zeros — with run-time `break`s the compiler keeps the
            // k-block loop rolled and the accumulators / operand rings, indexed by a run-time counter then, go to private memory: measured)
            PINN_UNROLL for (int kb = 0; kb < PD; ++kb) load(kb, Af[kb], Bf[kb]);
            PINN_UNROLL for (int kb = 0; kb < NR; ++kb) {
                if (kb + PD < NR) load(kb + PD, Af[(kb + PD) % (PD + 1)], Bf[(kb + PD) % (PD + 1)]);
                PINN_UNROLL for (int t = 0; t < HT; ++t)
                    PINN_UNROLL for (int g = 0; g < CG; ++g) mfma_f64(Z, (4 * t) * CG + g, CG, Af[kb % (PD + 1)], t, Bf[kb % (PD + 1)], g);
                sched_fence();                                   // (keeps the operand requests where they are: unfenced, the scheduler hoists the loads of all 32 k-blocks to the front and spills)
            }
            PINN_LANES(l) {
                PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                    const int m = 16 * (tr >> 2) + 4 * (tr & 3) + (l >> 4);
                    if (m < n_out) {
                        PINN_UNROLL for (int g = 0; g < CG; ++g) {
                            const int c = g0 + g;
                            if (c < ce) S[f64m_six(a, (size_t)zrow + (size_t)m * C + c, pbase + (l & 15))] = Z(l, tr * CG + g) + ((bias && c == 0) ? bias[m] : 0.0);
                        }
                    }
                }
            }
        }
    };
    // =========================== forward ===========================
    for (int ni = 0; ni < a.nnets; ++ni) {
        const F64Net& n = a.net[ni];
        const int L = n.nl - 1;
        const int ce = n.ceff;                                   // channels [0, ce) of this network are carried (F64Net::ceff), the rest stay zero
        LVd<C> u;                                                // this lane's partial sums of the output layer over its neurons
        PINN_LANES(l) { PINN_UNROLL for (int c = 0; c < C; ++c) u(l, c) = 0.0; }
        const double* WL = a.theta + n.woff[L];
        for (int lyr = 0; lyr < L; ++lyr) {
            const int n_in = n.sizes[lyr], n_out = n.sizes[lyr + 1];
            const double* W = a.theta + n.woff[lyr];
            const double* B = a.theta + n.boff[lyr];
            if (lyr > 0) gemm(W, 1, n_out, n_out, n_in, n.r_post[lyr - 1], n.r_rec[lyr], B, ce);
            // element-wise pass: pre-activation jets (layer 0: formed here) -> record, post-activation jets (last hidden layer: straight into the output sums).
            // TB tile rows at a time with every load of the batch in flight before the first use: one element per iteration would expose a full
            // memory round trip per element (2 waves per SIMD cover nothing) — the first version of this kernel spent 14 x its MFMA time here
            PINN_LANES(l) {
                const int q = l >> 4, j = l & 15;
                const int p = pbase + j, pc = p < a.npts ? p : a.npts - 1;
                const int ce_v = opaque_lane(ce);                    // (selects, not uniform branches, around the loads below)
                double x[4] = {0.0, 0.0, 0.0, 0.0};
                if (lyr == 0) { for (int i = 0; i < n.d; ++i) x[i] = a.pts[(size_t)(a.p0 + pc) * a.dt + n.imap[i]]; }
                for (int tr0 = 0; tr0 < NR; tr0 += TB) {
                    if (16 * (tr0 >> 2) >= n_out) break;
                    double z[TB][C], wl[TB];
                    PINN_UNROLL for (int b = 0; b < TB; ++b) {
                        const int tr = tr0 + b, m = 16 * (tr >> 2) + 4 * (tr & 3) + q, mc = m < n_out ? m : n_out - 1;
                        if (lyr == 0) {
                            double z0 = B[mc];
                            for (int i = 0; i < n.d; ++i) z0 = vfma(W[mc + (size_t)i * n_out], x[i], z0);
                            PINN_UNROLL for (int c = 0; c < C; ++c) z[b][c] = 0.0;
                            z[b][0] = z0;
                            PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf) { const double wv = W[mc + (size_t)J::first_axis(kf) * n_out]; z[b][J::CH_FIRST + kf] = (J::CH_FIRST + kf < ce_v) ? wv : 0.0; }
                        } else {
                            PINN_UNROLL for (int c = 0; c < C; ++c) { const double v = S[f64m_six(a, (size_t)n.r_rec[lyr] + (size_t)mc * C + c, p)]; z[b][c] = (c < ce_v) ? v : 0.0; }
                        }
                        wl[b] = (lyr == L - 1 && m < n_out) ? WL[mc] : 0.0;
                    }
                    PINN_UNROLL for (int b = 0; b < TB; ++b) {
                        const int tr = tr0 + b, m = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                        const bool valid = m < n_out;
                        const double a0 = act_value<SIN>(f64_act(n, lyr), z[b][0]);
                        z[b][0] = act_record<SIN>(z[b][0], a0);
                        if (rev && valid) {
                            if (lyr == 0) { PINN_UNROLL for (int c = 0; c < C; ++c) if (c < ce) S[f64m_six(a, (size_t)n.r_rec[lyr] + (size_t)m * C + c, p)] = z[b][c]; }
                            else if (!SIN) S[f64m_six(a, (size_t)n.r_rec[lyr] + (size_t)m * C, p)] = z[b][0];
                        }
                        double dd[ND];
                        act_derivs_n<J::NORD - 1, SIN>(f64_act(n, lyr), z[b][0], dd);
                        jet_forward<J>(z[b], dd);
                        z[b][0] = a0;
                        if (lyr < L - 1) {
                            // (value-only tanh / sigmoid terms: r_post == r_rec, the activation IS the record — the store above already wrote it when rev)
                            if (valid && !(a.post_alias && rev)) { PINN_UNROLL for (int c = 0; c < C; ++c) if (c < ce) S[f64m_six(a, (size_t)n.r_post[lyr] + (size_t)m * C + c, p)] = z[b][c]; }
                        } else {
                            PINN_UNROLL for (int c = 0; c < C; ++c) u(l, c) = vfma(wl[b], z[b][c], u(l, c));
                        }
                    }
                }
            }
        }
        if (L == 0) {                                            // a single Dense layer: u = W x + b
            PINN_LANES(l) {
                const int p = pbase + (l & 15), pc = p < a.npts ? p : a.npts - 1;
                if ((l >> 4) == 0) {
                    for (int i = 0; i < n.d; ++i) u(l, 0) = vfma(WL[i], a.pts[(size_t)(a.p0 + pc) * a.dt + n.imap[i]], u(l, 0));
                    PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf) u(l, J::CH_FIRST + kf) = WL[J::first_axis(kf)];
                }
            }
        }
        PINN_UNROLL for (int c = 0; c < C; ++c) lv_qsum(u, c);
        const double bo = a.theta[n.boff[L]];
        PINN_LANES(l) {
            if ((l >> 4) == 0) { PINN_UNROLL for (int c = 0; c < C; ++c) ulds[(ni * C + c) * 16 + (l & 15)] = u(l, c) + (c == 0 ? bo : 0.0); }
        }
    }
    // =========================== residual tape (lane group 0: one lane per point) ===========================
    LVd<1 + MAX_PARAMS> TS;
    PINN_LANES(l) { PINN_UNROLL for (int e = 0; e < 1 + MAX_PARAMS; ++e) TS(l, e) = 0.0; }
    PINN_LANES(l) {
        const int q = l >> 4, j = l & 15;
        const int R0 = a.dt + a.np + a.nslots;
        const int p = pbase + j, pc = p < a.npts ? p : a.npts - 1, gp = a.p0 + pc;
        const bool live = p < a.npts;
        if (q == 0) {
            double v[F64_MAX_ROWS];
            for (int i = 0; i < a.dt; ++i) v[i] = a.pts[(size_t)gp * a.dt + i];
            for (int k = 0; k < a.np; ++k) v[a.dt + k] = k < a.ne ? a.theta[a.p_off + k] : a.pdef[k];
            for (int s = 0; s < a.nslots; ++s) v[a.dt + a.np + s] = ulds[(a.slot_net[s] * C + a.slot_chan[s]) * 16 + j];
            for (int o = 0; o < a.nops; ++o) {
                const rp::Instr ins = a.prog[o];
                const double va = rp::is_nullary(ins.code) ? 0.0 : v[ins.a];
                const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
                v[R0 + o] = (ins.code == rp::OP_DATA) ? a.data[(size_t)(int)a.imm[o] * (size_t)a.N + (size_t)gp] : rp::apply<double, double>(ins.code, va, vb, a.imm[o]);
            }
            const double r = v[a.out_row];
            if (a.mode == 2) { if (live) a.resid[gp] = r; }
            else {
                const double sw = a.pw ? (double)a.pw[gp] : 1.0;
                const double rs = r * sw;
                if (live) TS(l, 0) = rs * rs;
                if (a.mode == 0) {
                    double g[F64_MAX_ROWS];
                    for (int o = 0; o < R0 + a.nops; ++o) g[o] = 0.0;
                    g[a.out_row] = 1.0;
                    for (int o = a.nops - 1; o >= 0; --o) {
                        const rp::Instr ins = a.prog[o];
                        if (rp::is_nullary(ins.code)) continue;
                        const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
                        double da, db;
                        rp::adjoint<double, double>(ins.code, v[ins.a], vb, v[R0 + o], a.imm[o], g[R0 + o], da, db);
                        g[ins.a] += da;
                        if (rp::is_binary(ins.code)) g[ins.b] += db;
                    }
                    const double rbar = live ? rs * a.scale * sw : 0.0;
                    PINN_UNROLL for (int k = 0; k < MAX_PARAMS; ++k) if (k < a.ne) TS(l, 1 + k) = rbar * g[a.dt + k];
                    // the seeds of every network's output jets replace its outputs
                    for (int ni = 0; ni < a.nnets; ++ni)
                        for (int c = 0; c < C; ++c) ulds[(ni * C + c) * 16 + j] = 0.0;
                    for (int s = 0; s < a.nslots; ++s) ulds[(a.slot_net[s] * C + a.slot_chan[s]) * 16 + j] += rbar * g[a.dt + a.np + s];
                }
            }
        }
    }
    if (a.mode == 2) return;
    PINN_UNROLL for (int e = 0; e < 1 + MAX_PARAMS; ++e) { if (e > a.ne) break; lv_rowsum(TS, e); }
    PINN_LANES(l) {
        if (l == 0) {
            TP[a.tp_p + a.ne] = TS(l, 0);
            PINN_UNROLL for (int k = 0; k < MAX_PARAMS; ++k) if (k < a.ne && a.mode == 0) TP[a.tp_p + k] = TS(l, 1 + k);
        }
    }
    if (a.mode != 0) return;
    // =========================== reverse sweep, network by network ===========================
    for (int ni = 0; ni < a.nnets; ++ni) {
        const F64Net& n = a.net[ni];
        const int L = n.nl - 1;
        const int ce = n.ceff;
        {                                                        // output-bias gradient: the tile's sum of the value seeds
            LVd<1> bl;
            PINN_LANES(l) { bl(l, 0) = (l >> 4) == 0 ? ulds[(ni * C) * 16 + (l & 15)] : 0.0; }
            lv_rowsum(bl, 0);
            PINN_LANES(l) { if (l == 0) TP[n.tp0 + (n.d + 1) * n.sizes[1] + n.sizes[n.nl - 1]] = bl(l, 0); }
        }
        for (int lyr = L - 1; lyr >= 0; --lyr) {
            const int H = n.sizes[lyr + 1];
            const int n_next = n.sizes[lyr + 2];
            const double* Wn = a.theta + n.woff[lyr + 1];            // W_{lyr+1}[m + k * n_next]
            // G = W_{lyr+1}^T dZ_{lyr+1} into this layer's dZ rows (the output layer's row vector: formed per element below)
            if (lyr < L - 1) gemm(Wn, n_next, 1, H, n_next, n.r_dz[lyr + 1], n.r_dz[lyr], nullptr, ce);
            for (int tr0 = 0; tr0 < NR; tr0 += TB) {
                if (16 * (tr0 >> 2) >= H) break;
                LVd<6 * TB> T6;                                      // per row of the batch: [0, d) dW_0[m][i], [4] db_0[m], [5] dW_L[m]
                PINN_LANES(l) {
                    const int q = l >> 4, j = l & 15;
                    const int p = pbase + j, pc = p < a.npts ? p : a.npts - 1;
                    const int ce_v = opaque_lane(ce);
                    double s[TB][C], gq[TB][C], ub[C], xin[4] = {0.0, 0.0, 0.0, 0.0};
                    if (lyr == 0) { for (int i = 0; i < n.d; ++i) xin[i] = a.pts[(size_t)(a.p0 + pc) * a.dt + n.imap[i]]; }
                    if (lyr == L - 1) { PINN_UNROLL for (int c = 0; c < C; ++c) ub[c] = ulds[(ni * C + c) * 16 + j]; }
                    PINN_UNROLL for (int b = 0; b < TB; ++b) {
                        const int tr = tr0 + b, k = 16 * (tr >> 2) + 4 * (tr & 3) + q, kc = k < H ? k : H - 1;
                        PINN_UNROLL for (int c = 0; c < C; ++c) { const double v = S[f64m_six(a, (size_t)n.r_rec[lyr] + (size_t)kc * C + c, p)]; s[b][c] = (c < ce_v) ? v : 0.0; }
                        if (lyr == L - 1) {
                            const double w = k < H ? Wn[kc] : 0.0;
                            PINN_UNROLL for (int c = 0; c < C; ++c) gq[b][c] = w * ub[c];
                        } else {
                            PINN_UNROLL for (int c = 0; c < C; ++c) { const double v = S[f64m_six(a, (size_t)n.r_dz[lyr] + (size_t)kc * C + c, p)]; gq[b][c] = (c < ce_v) ? v : 0.0; }
                        }
                    }
                    PINN_UNROLL for (int b = 0; b < TB; ++b) {
                        const int tr = tr0 + b, k = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                        const bool valid = k < H, st = valid && p < a.npts;
                        double t6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, dd[ND];
                        act_derivs_n<J::NORD, SIN>(f64_act(n, lyr), s[b][0], dd);
                        if (lyr == L - 1) {                          // dW_L[k] += sum_c ubar_c * (post-activation jet c of neuron k): the forward rule on the record
                            double pz[C];
                            PINN_UNROLL for (int c = 0; c < C; ++c) pz[c] = s[b][c];
                            jet_forward<J>(pz, dd);
                            pz[0] = SIN ? act_value<SIN>(f64_act(n, lyr), s[b][0]) : s[b][0];
                            double t2 = 0.0;
                            PINN_UNROLL for (int c = 0; c < C; ++c) t2 = vfma(ub[c], pz[c], t2);
                            t6[5] = st ? t2 : 0.0;
                        }
                        jet_adjoint<J>(gq[b], s[b], dd);
                        if (valid && lyr > 0) { PINN_UNROLL for (int c = 0; c < C; ++c) if (c < ce) S[f64m_six(a, (size_t)n.r_dz[lyr] + (size_t)k * C + c, p)] = st ? gq[b][c] : 0.0; }
                        if (lyr == 0) {
                            const double z0 = st ? gq[b][0] : 0.0;
                            t6[4] = z0;
                            PINN_UNROLL for (int i = 0; i < 4; ++i) {
                                if (i >= n.d) break;
                                double t2 = z0 * xin[i];
                                const int ck = a.first_ch[i];
                                PINN_UNROLL for (int c = 1; c < C; ++c) if (c == ck) t2 += st ? gq[b][c] : 0.0;
                                t6[i] = t2;
                            }
                        }
                        PINN_UNROLL for (int i = 0; i < 6; ++i) T6(l, 6 * b + i) = t6[i];
                    }
                }
                PINN_UNROLL for (int b = 0; b < TB; ++b) {
                    if (lyr == 0) {
                        PINN_UNROLL for (int i = 0; i < 4; ++i) { if (i >= n.d) break; lv_rowsum(T6, 6 * b + i); }
                        lv_rowsum(T6, 6 * b + 4);
                    }
                    if (lyr == L - 1) lv_rowsum(T6, 6 * b + 5);
                }
                PINN_LANES(l) {
                    PINN_UNROLL for (int b = 0; b < TB; ++b) {
                        const int tr = tr0 + b, k = 16 * (tr >> 2) + 4 * (tr & 3) + (l >> 4);
                        if ((l & 15) == 0 && k < H) {
                            const int n1 = n.sizes[1];
                            if (lyr == 0) {
                                PINN_UNROLL for (int i = 0; i < 4; ++i) if (i < n.d) TP[n.tp0 + i * n1 + k] = T6(l, 6 * b + i);
                                TP[n.tp0 + n.d * n1 + k] = T6(l, 6 * b + 4);
                            }
                            if (lyr == L - 1) TP[n.tp0 + (n.d + 1) * n1 + k] = T6(l, 6 * b + 5);
                        }
                    }
                }
            }
        }
    }
}

// ---- kernel B2'': hidden-to-hidden weight gradients with the OUTPUT tiles split across the workgroup's four waves (HT = 8: wave w owns output
// tiles 2 w, 2 w + 1 and every input tile: 64 accumulators; family 4m's kernel gives every wave all HT^2 tiles of a quarter of the points).  Each
// wave walks ALL 16-point steps of the block; its slab entries are its own: no combine. ----
template <int HT>
DEV void f64s_dwt_wave(int ni, int lyr, int b, int w, const F64Args& a) {
    constexpr int OW = HT / F64M_DWT_WAVES;                      // output tiles per wave
    static_assert(OW >= 1 && OW * F64M_DWT_WAVES == HT, "the split dW kernel needs HT a multiple of the wave count");
    const F64Net& n = a.net[ni];
    const int n_out = n.sizes[lyr + 1], n_in = n.sizes[lyr], C = a.C, CE = n.ceff;      // rows of channels >= CE are never written (F64Net::ceff)
    const int lo = b * F64_BLOCK, hi = (lo + F64_BLOCK < a.npts) ? lo + F64_BLOCK : a.npts;
    const double* S = a.scratch;
    LVd<OW * HT * 4> acc;
    LVd<OW> bsum;
    PINN_LANES(l) {
        PINN_UNROLL for (int e = 0; e < OW * HT * 4; ++e) acc(l, e) = 0.0;
        PINN_UNROLL for (int t = 0; t < OW; ++t) bsum(l, t) = 0.0;
    }
    const int nsteps = ((hi - lo + 15) / 16) * CE;
    for (int i = 0; i < nsteps; ++i) {
        const int p = lo + 16 * (i / CE), c = i % CE;
#if PINN_F64S_DWT_SYNC && !defined(PINN_EMU)
        // (experiment: the four waves read the SAME input rows of a step; kept in step, three of the four reads could be cache hits — the barrier costs more)
        __syncthreads();
#endif
        LVd<OW * 4> A_;
        LVd<HT * 4> B_;
        PINN_LANES(l) {
            const int pp = p + 4 * (l >> 4);
            PINN_UNROLL for (int t = 0; t < OW; ++t) {
                const int m = 16 * (w * OW + t) + (l & 15);
                double za[4];
                ld4_f64(S + f64m_six(a, (size_t)n.r_dz[lyr] + (size_t)(m < n_out ? m : 0) * C + c, pp), za);
                PINN_UNROLL for (int s4 = 0; s4 < 4; ++s4) A_(l, 4 * t + s4) = (m < n_out && pp + s4 < hi) ? za[s4] : 0.0;
            }
            PINN_UNROLL for (int t = 0; t < HT; ++t) {
                const int m = 16 * t + (l & 15);
                double ia[4];
                ld4_f64(S + f64m_six(a, (size_t)n.r_post[lyr - 1] + (size_t)(m < n_in ? m : 0) * C + c, pp), ia);
                PINN_UNROLL for (int s4 = 0; s4 < 4; ++s4) B_(l, 4 * t + s4) = (m < n_in && pp + s4 < hi) ? ia[s4] : 0.0;
            }
        }
        PINN_UNROLL for (int s4 = 0; s4 < 4; ++s4)
            PINN_UNROLL for (int to = 0; to < OW; ++to) {
                if (16 * (w * OW + to) >= n_out) break;
                PINN_UNROLL for (int t = 0; t < HT; ++t) {
                    if (16 * t >= n_in) break;
                    mfma_f64(acc, (to * HT + t) * 4, 1, A_, 4 * to + s4, B_, 4 * t + s4);
                }
            }
        if (c == 0) {
            PINN_LANES(l) {
                PINN_UNROLL for (int to = 0; to < OW; ++to) bsum(l, to) += (A_(l, 4 * to) + A_(l, 4 * to + 1)) + (A_(l, 4 * to + 2) + A_(l, 4 * to + 3));
            }
        }
    }
    PINN_LANES(l) {
        PINN_UNROLL for (int e = 0; e < OW * HT * 4; ++e) {
            const int to = e / (HT * 4), t = (e / 4) % HT, r = e % 4;
            const int m = 16 * (w * OW + to) + (l >> 4) + 4 * r, k = 16 * t + (l & 15);
            if (m < n_out && k < n_in) a.slab[(size_t)b * a.nent + n.ent0 + (n.woff[lyr] - n.theta0) + m + (size_t)k * n_out] = acc(l, e);
        }
    }
    PINN_UNROLL for (int to = 0; to < OW; ++to) lv_qsum(bsum, to);
    PINN_LANES(l) {
        PINN_UNROLL for (int to = 0; to < OW; ++to) {
            const int m = 16 * (w * OW + to) + (l & 15);
            if ((l >> 4) == 0 && m < n_out) a.slab[(size_t)b * a.nent + n.ent0 + (n.boff[lyr] - n.theta0) + m] = bsum(l, to);
        }
    }
}

#ifdef PINN_EMU
template <class J, int HT, int CG> void launch_f64s_tile(const F64Args& a, plat_stream) {
    const int nt = (a.npts + 15) / 16;
    std::vector<double> ulds((size_t)F64_MAX_NETS * J::C * 16);
    for (int t = 0; t < nt; ++t) f64s_tile<J, HT, CG, ACT_TANH>(t, a, ulds.data());
}
template <int HT> void launch_f64s_dwt(const F64Args& a, plat_stream) {
    const int nb = (a.npts + F64_BLOCK - 1) / F64_BLOCK, nl = f64m_num_layers(a);
    for (int b = 0; b < nb; ++b)
        for (int e = 0; e < a.ntp; ++e) f64m_tsum_entry(e, b, a);
    if (a.mode != 0) return;
    for (int b = 0; b < nb; ++b)
        for (int li = 0; li < nl; ++li) {
            int ni = 0, lyr = 1;
            if (!f64m_dwt_locate(li, a, ni, lyr)) continue;
            for (int w = 0; w < F64M_DWT_WAVES; ++w) f64s_dwt_wave<HT>(ni, lyr, b, w, a);
        }
}
#else
template <class J, int HT, int CG> __global__ void __launch_bounds__(64, PINN_F64S_WAVES) k_f64s_tile(const F64Args a) {
    __shared__ double ulds[F64_MAX_NETS * J::C * 16];
    f64s_tile<J, HT, CG, ACT_TANH>((int)blockIdx.x, a, ulds);
}
template <int HT> __global__ void __launch_bounds__(64 * F64M_DWT_WAVES, 2) k_f64s_dwt(const F64Args a) {
    int ni = 0, lyr = 1;
    if (!f64m_dwt_locate((int)blockIdx.x, a, ni, lyr)) {
        for (int e = (int)threadIdx.x; e < a.ntp; e += 64 * F64M_DWT_WAVES) f64m_tsum_entry(e, (int)blockIdx.y, a);
        return;
    }
    if (a.mode != 0) return;
    f64s_dwt_wave<HT>(ni, lyr, (int)blockIdx.y, (int)(threadIdx.x >> 6), a);
}
template <class J, int HT, int CG> void launch_f64s_tile(const F64Args& a, plat_stream st) {
    const int nt = (a.npts + 15) / 16;
    hipLaunchKernelGGL((k_f64s_tile<J, HT, CG>), dim3(nt), dim3(64), 0, st, a);
}
template <int HT> void launch_f64s_dwt(const F64Args& a, plat_stream st) {
    const int nb = (a.npts + F64_BLOCK - 1) / F64_BLOCK, nl = f64m_num_layers(a);
    hipLaunchKernelGGL((k_f64s_dwt<HT>), dim3(nl + 1, nb), dim3(64 * F64M_DWT_WAVES), 0, st, a);
}
#endif

// the dW kernel of a sliced entry: family 4m's (every wave all tiles of a quarter of the points, LDS combine) while its HT^2 x 4 accumulators fit a
// wave (HT <= 4), the output-split one above for wider layers
template <int HT> void launch_f64s_dwt_any(const F64Args& a, plat_stream st) {
    if constexpr (HT <= 4) launch_f64m_dwt<HT>(a, st);
    else launch_f64s_dwt<HT>(a, st);
}

template <int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, int HT> F64MKernel make_f64s_kernel() {
    using J = JetSet<D1MASK, PAIRS, NPAIR, HI>;
    static_assert(J::NLAP == 0, "the float64 kernels carry plain derivative channels (no forward-Laplacian channel)");
    constexpr int CGMAX = PINN_F64S_ZMAX / HT;                   // accumulators 4 HT x CG <= 4 ZMAX doubles per lane (96 by default)
    constexpr int CG = (J::C < CGMAX) ? J::C : CGMAX;
    static_assert(CG >= 1, "layer too wide for the sliced float64 kernels");
    F64MKernel k;
    k.D = D; k.NPAIR = NPAIR; k.HT = HT; k.PG = 1; k.D1MASK = D1MASK; k.HI = HI; k.PAIRS = PAIRS;
    k.sliced = 1;
    k.launch_tile = &launch_f64s_tile<J, HT, CG>;
    k.launch_dwt = &launch_f64s_dwt_any<HT>;
    return k;
}
#define PINN_INSTANTIATE_F64S(NAME, D, D1MASK, PAIRS, NPAIR, HI, HT) \
    namespace { pk::F64MRegistrar NAME##_regf64s(pk::make_f64s_kernel<D, D1MASK, PAIRS, NPAIR, HI, HT>()); }

}  // namespace pk
