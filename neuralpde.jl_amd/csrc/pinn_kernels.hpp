// pinn_kernels.hpp — the PINN residual/loss hot path for gfx950 (CDNA4), one wavefront per point tile.
//
// Replaces, per collocation point (reference file:line):
//   Phi / Lux Dense forward                      src/pinn_types.jl:79-90
//   numeric_derivative (FD stencils)             src/pinn_types.jl:445-482   -> exact forward-mode Taylor jets
//   generated residual function                  src/discretize.jl:163-175   -> rprog.hpp tape
//   mean(abs2, residual)                         src/training_strategies.jl:220,280,380
//   weighted sum of terms                        src/discretize.jl:582-588
//   Zygote reverse mode of all of the above      src/discretize.jl:778       -> hand-derived reverse sweep
//
// Data flow of one wave-tile (TP = 16*PG points, C jet channels => NG = C*PG column groups of 16):
//   * every activation tensor of a layer lives in registers in the MFMA C/D layout:
//       A[q][m][r]  (vfloat4 per (column group q, 16-neuron tile m)):  neuron 16m + 4*(lane>>4) + r,
//       column (point) lane&15 of group q.   q = pg*C + channel.
//   * hidden GEMMs use v_mfma_f32_16x16x4_f32 with the layer weights as the A operand, pre-packed on
//     the device in fragment order, and the previous layer's D registers fed straight back in as the B
//     operand (the contraction index is permuted, k-step (mi,r') <-> neurons 16mi+4g+r', so no LDS
//     round trip between layers).
//   * the tanh/sigmoid Taylor-jet rules run lane-local on the VALU (all channels of a (neuron, point)
//     are in the same lane).
//   * (a, z_i, z_ij) of every hidden layer is parked in a per-wave scratch slab (lane-native, 16 B per
//     lane per store) for the reverse sweep; it is re-read by the same wave a few microseconds later
//     (L2 / Infinity-Cache resident).
//   * reverse sweep: activation adjoints lane-local; dA = W^T dZ by the same register-chained MFMA
//     scheme with the transposed packed weights; dW += dZ A^T needs both operands with the point index
//     on the contraction axis, i.e. transposed — done through a per-wave XOR-swizzled LDS buffer
//     (conflict-free ds_write_b128 / ds_read_b128), accumulating into persistent MFMA accumulators
//     that live in registers across all tiles of the wave (deterministic, no float atomics).
//   * quadrature sum of r^2 and the per-wave gradient slab are reduced in a fixed order by
//     k_reduce (pinn_aux_kernels.hpp) => bit-identical results run to run (BFGS callers need that,
//     test/NNPDE1/nnpde__pde_vi_pde_with_mixed_derivative.jl:77-80).
#pragma once
#include "rprog.hpp"
#include "vec.hpp"

#ifndef PINN_F1_PREFETCH
#define PINN_F1_PREFETCH 1
#endif
// Small nets keep their operands in REGISTERS (r04): a family-1 workgroup runs one wave per SIMD, so every lane owns the whole 512-entry
// register file, of which a 3 x 32 network's kernel used 280.  Bit 0: every hidden->hidden weight fragment (forward and transposed), the
// biases and the first layer are loaded ONCE per launch, in front of the tile loop — the k-steps of the GEMMs no longer wait for one L2 round
// trip each (32 dependent round trips per tile of a 3 x 32 net; a one-tile launch is nothing but latency).  Bit 1: the records of the hidden
// layers stay in registers between the forward and the reverse sweep instead of a store / load pair through the per-wave scratch slab.
// Same arithmetic in the same order either way: results are bit-identical to the streaming form (PINN_F1_RESIDENT=0).
#ifndef PINN_F1_RESIDENT
#define PINN_F1_RESIDENT 3
#endif

namespace pk {
using namespace wv;

// The record of a layer keeps r0 = phi(z) for tanh / sigmoid (their derivatives are polynomials in phi) and r0 = z for sin
// (cos z cannot be recovered from sin z); act_derivs_n / act_from_record take that r0.
enum Act : int { ACT_TANH = 0, ACT_SIGMOID = 1, ACT_SIN = 2,
                 ACT_MIXED = 3 /* kernel variant only: tanh / sigmoid chosen per hidden layer at run time, branch-free */ };
// FUSED: forward + residual tape + reverse sweep.  RESID: forward + tape, writes r.  FWD: forward only, writes the jet
// channels per point.  GRADIN: forward + reverse sweep seeded with per-point d(loss)/d(jet) read from memory (equations that
// couple several networks: the tape runs in k_expr between the FWD and GRADIN launches of every network involved).
// FWDREC / GRADREC (family 2): as FWD / GRADIN, but the forward launch keeps every hidden layer's record in HBM (per tile) and
// the reverse launch reads them back instead of running the forward pass a second time — 288 GB of HBM make the 8 KB/point affordable.
// LOSS: forward + tape + the per-term sums of squares, nothing else (no records, no reverse sweep, no gradient slabs): what the
// reference's per-term closures cost when they are only evaluated — callbacks, adaptive-weight rules, rejected line-search trials
// (src/training_strategies.jl:215-221).  The forward arithmetic is the FUSED kernel's, so the sums are the same numbers.
enum Mode : int { MODE_FUSED = 0, MODE_RESID = 1, MODE_FWD = 2, MODE_GRADIN = 3, MODE_FWDREC = 4, MODE_GRADREC = 5, MODE_LOSS = 6 };
constexpr bool mode_is_forward_only(int mode) { return mode == MODE_RESID || mode == MODE_FWD || mode == MODE_FWDREC || mode == MODE_LOSS; }

constexpr int MAX_GROUP_TERMS = 12;
constexpr int ND = 8;            // activation-derivative array: d[1] .. d[7] (jets up to order 6 need phi^(7) in the reverse sweep)
constexpr int MAX_PARAMS = 4;
constexpr int LIN_MAX_C = 8, LIN_MAX_SRC = 4;      // TermDev::linear covers up to 8 jet channels and 4 sources

// ------------------------------------------------------------------------------------------------
// The jet-channel set of a kernel: which derivatives of the trial function travel through the layers.
//   channel 0                        u
//   1 .. NFIRST                      du/dx_a            for the axes a in D1MASK
//   then NPAIR                       d2u/dx_a dx_b      pairs (a <= b) packed one per byte in PAIRS
//   then NLAP (0 or 1)               sum_a d2u/dx_a^2   over the axes in LAP = bits 24..31 of HI ("forward Laplacian": one channel
//                                                        carries the whole sum, a_L = phi'' sum_a z_a^2 + phi' z_L, instead of one
//                                                        channel per pure second derivative — used when a residual needs them only summed)
//   then N3                          d3u/dx_a^3         axes whose nibble in HI (bits 0..23) is >= 3
//   then N4                          d4u/dx_a^4         axes whose nibble in HI is 4
// (pure third / fourth derivatives need the lower pure derivatives of the same axis: first(a), pair(a,a), third(a).)
// ------------------------------------------------------------------------------------------------
template <unsigned D1MASK_, unsigned long long PAIRS_, int NPAIR_, unsigned HI_>
struct JetSet {
    static constexpr int popc(unsigned x) { int n = 0; while (x) { n += x & 1; x >>= 1; } return n; }
    static constexpr int NFIRST = popc(D1MASK_);
    static constexpr int NPAIR = NPAIR_;
    static constexpr int first_axis(int k) {
        int cnt = 0;
        for (int a = 0; a < 8; ++a)
            if (D1MASK_ & (1u << a)) { if (cnt == k) return a; ++cnt; }
        return -1;
    }
    static constexpr int first_rank(int axis) {
        int cnt = 0;
        for (int a = 0; a < axis; ++a) if (D1MASK_ & (1u << a)) ++cnt;
        return cnt;
    }
    static constexpr int pair_a(int p) { return (int)((PAIRS_ >> (8 * p)) & 0xF); }
    static constexpr int pair_b(int p) { return (int)((PAIRS_ >> (8 * p + 4)) & 0xF); }
    static constexpr int pair_index(int a, int b) {
        for (int p = 0; p < NPAIR_; ++p) if (pair_a(p) == a && pair_b(p) == b) return p;
        return -1;
    }
    static constexpr unsigned LAP = (HI_ >> 24) & 0xFFu;
    static constexpr int NLAP = LAP ? 1 : 0;
    static constexpr int hi_order(int axis) { return axis < 6 ? (int)((HI_ >> (4 * axis)) & 0xF) : 0; }
    static constexpr int hi_count(int k) { int n = 0; for (int a = 0; a < 8; ++a) if (hi_order(a) >= k) ++n; return n; }
    static constexpr int hi_axis(int k, int idx) {
        int cnt = 0;
        for (int a = 0; a < 8; ++a)
            if (hi_order(a) >= k) { if (cnt == idx) return a; ++cnt; }
        return -1;
    }
    static constexpr int hi_rank(int k, int axis) { int n = 0; for (int a = 0; a < axis; ++a) if (hi_order(a) >= k) ++n; return n; }
    static constexpr int N3 = hi_count(3), N4 = hi_count(4);
    static constexpr int C = 1 + NFIRST + NPAIR_ + NLAP + N3 + N4;
    static constexpr int CH_FIRST = 1, CH_PAIR = 1 + NFIRST, CH_LAP = CH_PAIR + NPAIR_, CH_3 = CH_LAP + NLAP, CH_4 = CH_3 + N3;
    static constexpr bool valid() {
        for (int a = 0; a < 8; ++a) {
            const int h = hi_order(a);
            if (h != 0 && h != 3 && h != 4) return false;
            if (h && (!(D1MASK_ & (1u << a)) || pair_index(a, a) < 0)) return false;
        }
        if ((LAP & D1MASK_) != LAP) return false;        // the Laplacian recurrence reads the first-derivative channels of its axes
        return true;
    }
    static_assert(valid(), "third/fourth derivative channels need first(a) and pair(a,a) of the same axis; Laplacian axes need first(a)");
    // derivatives of the activation needed by the reverse sweep (one more than the highest jet order)
    static constexpr int NORD = N4 > 0 ? 5 : (N3 > 0 ? 4 : 3);
    // general multi-index channel sets (mixed derivatives of order >= 3, orders 5-6) are SPECIALISATIONS of this template generated at
    // run time (jit.cpp: bit 31 of HI + a set id); gen_channel(i) = the multi-index of channel i, 0 for the fixed categories above
    static constexpr bool GEN = false;
    static constexpr unsigned gen_channel(int) { return 0u; }
    // multi-index of channel ch (nibble 0 = order, nibbles 1.. = sorted axes); the Laplacian channel is a sum, not a multi-index
    static constexpr unsigned channel_mi(int ch) {
        if (ch == 0) return 0u;
        if (ch < CH_PAIR) return 1u | ((unsigned)first_axis(ch - CH_FIRST) << 4);
        if (ch < CH_LAP) return 2u | ((unsigned)pair_a(ch - CH_PAIR) << 4) | ((unsigned)pair_b(ch - CH_PAIR) << 8);
        if (ch < CH_3) return 0xFFFFFFFFu;
        if (ch < CH_4) { const unsigned ax = (unsigned)hi_axis(3, ch - CH_3); return 3u | (ax << 4) | (ax << 8) | (ax << 12); }
        const unsigned ax = (unsigned)hi_axis(4, ch - CH_4);
        return 4u | (ax << 4) | (ax << 8) | (ax << 12) | (ax << 16);
    }
};

// ------------------------------------------------------------------------------------------------
// compile-time description of one kernel family member
// ------------------------------------------------------------------------------------------------
template <int HP_, int NHH_, int D_, unsigned D1MASK_, unsigned long long PAIRS_, int NPAIR_, int PG_, unsigned HI_ = 0>
struct Spec {
    using J = JetSet<D1MASK_, PAIRS_, NPAIR_, HI_>;
    static constexpr unsigned HI = HI_;
    static constexpr int HP = HP_;            // padded hidden width (multiple of 16)
    static constexpr int MT = HP_ / 16;       // 16-neuron tiles per layer
    static constexpr int NHH = NHH_;          // hidden->hidden layers
    static constexpr int LH = NHH_ + 1;       // hidden layers
    static constexpr int D = D_;              // input dimension
    static constexpr unsigned D1MASK = D1MASK_;
    static constexpr unsigned long long PAIRS = PAIRS_;
    static constexpr int NPAIR = NPAIR_;
    static constexpr int PG = PG_;            // 16-point groups per tile
    static constexpr int NFIRST = J::NFIRST;
    static constexpr int C = J::C;
    static constexpr int NG = C * PG_;
    static constexpr int TP = 16 * PG_;       // points per tile
    static constexpr int first_axis(int k) { return J::first_axis(k); }       // k-th first-order axis
    static constexpr int first_rank(int axis) { return J::first_rank(axis); } // rank of axis among first-order channels
    static constexpr int pair_a(int p) { return J::pair_a(p); }
    static constexpr int pair_b(int p) { return J::pair_b(p); }
    // packed parameter buffer (floats)
    static constexpr int OFF_W1 = 0;                       // [D][HP]
    static constexpr int OFF_B = OFF_W1 + D_ * HP_;        // [LH][HP]
    static constexpr int OFF_WL = OFF_B + LH * HP_;        // [HP]
    static constexpr int OFF_BL = OFF_WL + HP_;            // [4] (1 used)
    static constexpr int OFF_WPK = OFF_BL + 4;             // [NHH][HP*HP]  forward fragments
    static constexpr int OFF_WTPK = OFF_WPK + NHH_ * HP_ * HP_;   // [NHH][HP*HP] transposed fragments
    static constexpr int PACKED = OFF_WTPK + NHH_ * HP_ * HP_;
    // dW ownership: with 4 neuron tiles (HP = 64) the four waves of a workgroup each own one 16-row block of every
    // layer's dW (COOP); smaller nets keep all of dW per wave.
    static constexpr bool COOP = (MT == 4);
    static constexpr int WT = COOP ? 1 : MT;                     // dW row tiles held by one wave
    // per-WORKGROUP gradient slab (floats) = [shared section SH][4 x per-wave section PW]
    static constexpr int O_WBAR = 0;                                         // [NHH][MT(to)][MT(ti)][64][4]
    static constexpr int O_BFRH = NHH_ * HP_ * HP_;                          // [NHH][MT][16]  biases of hidden layers >= 1
    static constexpr int SH = COOP ? (NHH_ * HP_ * HP_ + NHH_ * HP_) : 0;
    static constexpr int PB = COOP ? 0 : (NHH_ * HP_ * HP_ + NHH_ * HP_);
    static constexpr int O_BFR0 = PB;                                        // [MT][16]     bias of hidden layer 0
    static constexpr int O_W1 = O_BFR0 + HP_;                                // [D][MT][16]
    static constexpr int O_WL = O_W1 + D_ * HP_;                             // [MT][4][4]
    static constexpr int O_BL = O_WL + HP_;                                  // [1]
    static constexpr int O_P = O_BL + 1;                                     // [MAX_PARAMS]
    static constexpr int PW = ((O_P + MAX_PARAMS + 63) / 64) * 64;
    static constexpr int SLAB = SH + 4 * PW;
    // per-wave activation scratch (floats): [LH][NG][MT][64][4]
    static constexpr int SCR = LH * NG * MT * 256;
    // LDS (floats).  Per wave private: residual tape (32 value rows + 32 adjoint rows), aliased by the wave-private
    // transpose buffers of the reverse sweep, + the tile's coordinates.  COOP adds a workgroup-shared double buffer of
    // the four waves' (dZ^T, A^T) 16-column chunks.
    static constexpr int LDS_T = (16 * HP_ > 1024) ? 16 * HP_ : 1024;
    static constexpr int LDS_X = ((16 * PG_ * D_ + 63) / 64) * 64;
    static constexpr int LDS_PRIV = 4 * LDS_T + LDS_X;
    static constexpr int LDS_SHARED = COOP ? 2 * 4 * 2 * LDS_T : 0;
    static constexpr int LDS_WG = LDS_SHARED + 4 * LDS_PRIV;
    // register-resident operands (PINN_F1_RESIDENT): registers per lane of the hidden->hidden fragments (forward + transposed), of the
    // stored records, and of what the tile loop keeps live anyway (seven activation-sized tensors + the dW accumulators)
    static constexpr int WREGS = 2 * NHH_ * 4 * MT * MT;
    static constexpr int RREGS = NHH_ * NG * MT * 4;
    static constexpr int LIVE = 7 * NG * MT * 4 + NHH_ * WT * MT * 4 + (LH + D_ + 1) * MT * 4;
    static constexpr bool W_RESIDENT = (PINN_F1_RESIDENT & 1) && NHH_ > 0 && MT < 4 && LIVE + WREGS <= 400;
    static constexpr bool R_RESIDENT = (PINN_F1_RESIDENT & 2) && NHH_ > 0 && LIVE + (W_RESIDENT ? WREGS : 0) + RREGS <= 400;
};

struct TermDev {
    const float* pts;        // d x N, point-major (Julia d×N column-major matrix)
    int N;                   // points of this term on this device
    int tile0;               // first tile of this term inside the group
    int ntiles;
    int prog_off;            // into GroupArgs::prog
    int nops;
    int out_row;
    int term_id;             // global term index (loss partial column)
    float scale;             // 2 * w_k / N_k(global)   — reverse-sweep seed of mean(abs2, r)
    float* out;              // MODE_RESID: residual r[N];  MODE_FWD: jets [C][N]
    const float* in;         // MODE_GRADIN: d(loss)/d(jet) [C][N]
    const float* src;        // [nsrc][N]: coordinate-only subexpressions of the residual, evaluated when the point set was installed
    int nsrc;                // tape rows dt+NP+C .. dt+NP+C+nsrc-1
    // tail launch of an equation that couples several networks (plan.cpp: Coupled::tail): the source rows are the OTHER networks' jet channels
    // (written by their forward launches); the fused kernel also delivers d(loss)/d(source row) = the seeds of those networks' reverse launches
    float* src_bar;          // [nsrc][N], nullptr: plain sources (no adjoints wanted)
    // The term binds dt coordinates (rows of its point matrix, tape rows 0..dt-1); input i of the group's network is
    // coordinate imap[i] of the term (src/discretize.jl:111-131: every depvar gets its own `cord` rows).  hetero = the map is
    // not the identity over dt == D coordinates (systems whose dependent variables take different arguments).
    int dt;
    int hetero;
    int imap[4];
    // optional per-point factors s_i = sqrt(N w_i) (pinn_set_point_weights): the term's loss becomes sum_i w_i r_i^2 — a quadrature
    // rule — instead of the plain mean; nullptr: mean(abs2, r)
    const float* pw;
    // residuals that are AFFINE in the jet channels and the hoisted sources with constant coefficients — every Dirichlet boundary term
    // (u - g), the Poisson interior term (lap u - f), linear PDEs with fixed coefficients — skip the tape interpreter in the neuron-split
    // kernels: r = lin_k + sum_c lin_a[c] U_c + sum_j lin_b[j] src_j, seeds dr/dU_c = lin_a[c]   (plan.cpp: detect_linear)
    int linear;
    float lin_k;
    float lin_a[LIN_MAX_C];
    float lin_b[LIN_MAX_SRC];
};

struct GroupArgs {
    const float* packed;         // Spec::PACKED floats (this net, this call's theta)
    const float* params;         // MAX_PARAMS floats (theta.p / default_p)
    const rp::Instr* prog;
    float* slabs;                // [nblocks][SLAB]
    double* losspart;            // [nwaves][nterms_total]
    float* scratch;              // [nwaves][SCR]
    float* rec;                  // MODE_FWDREC / MODE_GRADREC: [tile slots][Spec2::REC] records of hidden layers 1 .. LH-1
    int nterms_total;
    int nterms;                  // terms in this group
    int ntiles;
    int nparams;                 // NP rows in the tape
    int nparams_estim;           // first NE params get adjoints
    int act;                     // ACT_TANH / ACT_SIGMOID / ACT_SIN for all hidden layers, or ACT_MIXED:
    int act_layers;              //   kind of hidden layer l (tanh / sigmoid) in bits 4l .. 4l+3
    // family 3 (DGM, pinn_kernels3.hpp): `packed` points at the network's parameters inside theta (unpacked), `scratch` at the
    // point-major scratch rows [Spec3::ROWS][dgm_npad]
    int dgm_modes;               // real number of modes M (<= the padded MP of the kernel)
    int dgm_npad;                // points per scratch row (tiles x 64)
    int dgm_slab;                // floats per block of the gradient slab
    int dgm_nparams;             // parameters of the network (slab entries [0, nparams) = theta order; then MAX_PARAMS PDE-parameter sums)
    int chain;                   // family 2: add this launch's gradient onto the slab contents an earlier launch group of the same
                                 // network left behind (one slab set and one reduction input for several launch groups)
    int sub_terms0, sub_tiles0;  // merged launch (family 2, wave_main2m): terms [0, sub_terms0) / tiles [0, sub_tiles0) run the first
                                 // kernel-family member, the rest the second
    int scr_stride;              // floats between two workgroups' record scratch (0: the kernel's own Spec2::SCR).  A merged launch sets
                                 // the larger of its members' sizes: workgroups are in different members' tile lists at the same time, so
                                 // both members must address the scratch with ONE stride or their slots overlap
    TermDev terms[MAX_GROUP_TERMS];
};

// derivatives phi', ..., phi^(NORD) of the activation from the record value r0 (see Act)   (d[0] unused)
// MIXED (ACT_MIXED kernels): `act` is the run-time kind of THIS layer (tanh or sigmoid).  Both are one function family,
// f(z) = m s(m z) + 1 - m with s = logistic and m = 2 (tanh) or 1 (sigmoid), so the layer kind enters only through the scalar m:
// no branch behind the GEMMs (a run-time branch there is what the unmixed kernels avoid by taking the kind as a template parameter).
// s^(6) / (s (1 - s)) and s^(7) / (s (1 - s)) of the logistic function as polynomials in s (Horner form)
// (r04) the activation / jet rules below are templates over the lane-value type V: vfloat in the wave kernels (fp32), plain double in
// the per-point float64 kernels (pinn_kernels4.hpp) — one statement of the mathematics for both precisions
template <class V> DEV V sig_poly6(V s) {
    return vfma(vfma(vfma(vfma(vfma(V(-720.0f), s, V(1800.0f)), s, V(-1560.0f)), s, V(540.0f)), s, V(-62.0f)), s, V(1.0f));
}
template <class V> DEV V sig_poly7(V s) {
    return vfma(vfma(vfma(vfma(vfma(vfma(V(5040.0f), s, V(-15120.0f)), s, V(16800.0f)), s, V(-8400.0f)), s, V(1806.0f)), s, V(-126.0f)), s, V(1.0f));
}
template <int NORD, bool SINACT, bool MIXED = false, class V = vfloat>
DEV void act_derivs_n(int act, V a, V (&d)[ND]) {
    if (MIXED) {
        const float m = 2.0f - (float)act, im = 0.5f + 0.5f * (float)act, m2 = m * m;      // act in {ACT_TANH = 0, ACT_SIGMOID = 1}
        const V s = (a + V(m - 1.0f)) * V(im);                              // logistic value behind the record
        const V s1 = s * (V(1.0f) - s);
        const V s2 = s1 * vfma(V(-2.0f), s, V(1.0f));
        const V s3 = s1 * vfma(V(-6.0f), s1, V(1.0f));
        d[1] = V(m2) * s1;
        d[2] = V(m2 * m) * s2;
        d[3] = V(m2 * m2) * s3;
        if (NORD >= 4) d[4] = V(m2 * m2 * m) * (s2 * vfma(V(-12.0f), s1, V(1.0f)));
        if (NORD >= 5) d[5] = V(m2 * m2 * m2) * vfma(s3, vfma(V(-12.0f), s1, V(1.0f)), V(-12.0f) * s2 * s2);
        if (NORD >= 6) d[6] = V(m2 * m2 * m2 * m) * s1 * sig_poly6(s);
        if (NORD >= 7) d[7] = V(m2 * m2 * m2 * m2) * s1 * sig_poly7(s);
        return;
    }
    if (SINACT) {
        V sn, cs;
        vsincos(a, sn, cs);
        d[1] = cs;
        d[2] = V(0.f) - sn;
        d[3] = V(0.f) - cs;
        if (NORD >= 4) d[4] = sn;
        if (NORD >= 5) d[5] = cs;
        if (NORD >= 6) d[6] = V(0.f) - sn;
        if (NORD >= 7) d[7] = V(0.f) - cs;
    } else if (act == ACT_TANH) {
        const V a2 = a * a;
        d[1] = vfma(-a, a, V(1.0f));                           // 1 - a^2 with one rounding (the negation is an operand modifier)
        d[2] = V(-2.0f) * a * d[1];
        d[3] = d[1] * vfma(V(6.0f), a2, V(-2.0f));
        if (NORD >= 4) d[4] = d[1] * a * vfma(V(-24.0f), a2, V(16.0f));
        if (NORD >= 5) d[5] = d[1] * (vfma(vfma(V(120.0f), a2, V(-120.0f)), a2, V(16.0f)));
        if (NORD >= 6) d[6] = d[1] * a * vfma(vfma(V(-720.0f), a2, V(960.0f)), a2, V(-272.0f));
        if (NORD >= 7) d[7] = d[1] * vfma(vfma(vfma(V(5040.0f), a2, V(-8400.0f)), a2, V(3696.0f)), a2, V(-272.0f));
    } else {
        d[1] = a * (V(1.0f) - a);
        d[2] = d[1] * vfma(V(-2.0f), a, V(1.0f));
        d[3] = d[1] * vfma(V(-6.0f), d[1], V(1.0f));
        if (NORD >= 4) d[4] = d[2] * vfma(V(-12.0f), d[1], V(1.0f));
        if (NORD >= 5) d[5] = vfma(d[3], vfma(V(-12.0f), d[1], V(1.0f)), V(-12.0f) * d[2] * d[2]);
        if (NORD >= 6) d[6] = d[1] * sig_poly6(a);
        if (NORD >= 7) d[7] = d[1] * sig_poly7(a);
    }
}
// One element's jet through the activation (Faa di Bruno), in place: z[0] = a = phi(z0) on entry, z[k>0] = pre-activation
// channels; on exit z[k] = post-activation channels.  Higher channels first: they read the lower pre-activation values.
template <class J, class V>
DEV void jet_forward(V (&z)[J::C], const V (&d)[ND]) {
    PINN_UNROLL for (int k = 0; k < J::N4; ++k) {
        const int ax = J::hi_axis(4, k);
        const V z1 = z[J::CH_FIRST + J::first_rank(ax)], z2 = z[J::CH_PAIR + J::pair_index(ax, ax)], z3 = z[J::CH_3 + J::hi_rank(3, ax)];
        const V z11 = z1 * z1;
        V r = d[1] * z[J::CH_4 + k];
        r = vfma(V(4.0f) * d[2] * z1, z3, r);
        r = vfma(V(3.0f) * d[2] * z2, z2, r);
        r = vfma(V(6.0f) * d[3] * z11, z2, r);
        r = vfma(d[4] * z11, z11, r);
        z[J::CH_4 + k] = r;
    }
    PINN_UNROLL for (int k = 0; k < J::N3; ++k) {
        const int ax = J::hi_axis(3, k);
        const V z1 = z[J::CH_FIRST + J::first_rank(ax)], z2 = z[J::CH_PAIR + J::pair_index(ax, ax)];
        V r = d[1] * z[J::CH_3 + k];
        r = vfma(V(3.0f) * d[2] * z1, z2, r);
        r = vfma(d[3] * z1 * z1, z1, r);
        z[J::CH_3 + k] = r;
    }
    PINN_UNROLL for (int p = 0; p < J::NPAIR; ++p) {
        const int ca = J::CH_FIRST + J::first_rank(J::pair_a(p)), cb = J::CH_FIRST + J::first_rank(J::pair_b(p));
        z[J::CH_PAIR + p] = vfma(d[2] * z[ca], z[cb], d[1] * z[J::CH_PAIR + p]);
    }
    if (J::NLAP) {
        V sq = V(0.f);
        PINN_UNROLL for (int a = 0; a < 8; ++a)
            if (J::LAP & (1u << a)) sq = vfma(z[J::CH_FIRST + J::first_rank(a)], z[J::CH_FIRST + J::first_rank(a)], sq);
        z[J::CH_LAP] = vfma(d[2], sq, d[1] * z[J::CH_LAP]);
    }
    PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf) z[J::CH_FIRST + kf] = d[1] * z[J::CH_FIRST + kf];
}
// Adjoint of jet_forward for one element: g[k] = adjoint of post-activation channel k on entry, of pre-activation channel k
// on exit; s = the record (s[0] = a, s[k>0] = pre-activation channels).
template <class J, class V>
DEV void jet_adjoint(V (&g)[J::C], const V (&s)[J::C], const V (&d)[ND]) {
#if !defined(PINN_EMU)
#pragma clang fp contract(fast)      // reverse-sweep only: nothing here feeds a residual, so the compiler may fuse what it finds (vec.hpp)
#endif
    V zv = d[1] * g[0];
    V zf[J::NFIRST > 0 ? J::NFIRST : 1], zp[J::NPAIR > 0 ? J::NPAIR : 1], z3b[J::N3 > 0 ? J::N3 : 1];
    PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf) {
        const V gk = g[J::CH_FIRST + kf];
        zv = vfma(d[2] * s[J::CH_FIRST + kf], gk, zv);
        zf[kf] = d[1] * gk;
    }
    PINN_UNROLL for (int p = 0; p < J::NPAIR; ++p) {
        const int ka = J::first_rank(J::pair_a(p)), kb = J::first_rank(J::pair_b(p));
        const V za = s[J::CH_FIRST + ka], zb = s[J::CH_FIRST + kb], gp = g[J::CH_PAIR + p];
        zv = vfma(vfma(d[3] * za, zb, d[2] * s[J::CH_PAIR + p]), gp, zv);
        zf[ka] = vfma(d[2] * zb, gp, zf[ka]);
        zf[kb] = vfma(d[2] * za, gp, zf[kb]);
        zp[p] = d[1] * gp;
    }
    if (J::NLAP) {
        const V gl = g[J::CH_LAP];
        V sq = V(0.f);
        PINN_UNROLL for (int a = 0; a < 8; ++a)
            if (J::LAP & (1u << a)) {
                const V za = s[J::CH_FIRST + J::first_rank(a)];
                sq = vfma(za, za, sq);
                zf[J::first_rank(a)] = vfma(V(2.0f) * d[2] * za, gl, zf[J::first_rank(a)]);
            }
        zv = vfma(vfma(d[3], sq, d[2] * s[J::CH_LAP]), gl, zv);
        g[J::CH_LAP] = d[1] * gl;
    }
    PINN_UNROLL for (int k = 0; k < J::N3; ++k) {
        const int ax = J::hi_axis(3, k), k1 = J::first_rank(ax), p2 = J::pair_index(ax, ax);
        const V z1 = s[J::CH_FIRST + k1], z2 = s[J::CH_PAIR + p2], z3 = s[J::CH_3 + k], g3 = g[J::CH_3 + k];
        z3b[k] = d[1] * g3;
        zp[p2] = vfma(V(3.0f) * d[2] * z1, g3, zp[p2]);
        zf[k1] = vfma(V(3.0f) * vfma(d[3] * z1, z1, d[2] * z2), g3, zf[k1]);
        zv = vfma(vfma(d[4] * z1 * z1, z1, vfma(V(3.0f) * d[3] * z1, z2, d[2] * z3)), g3, zv);
    }
    PINN_UNROLL for (int k = 0; k < J::N4; ++k) {
        const int ax = J::hi_axis(4, k), k1 = J::first_rank(ax), p2 = J::pair_index(ax, ax), k3 = J::hi_rank(3, ax);
        const V z1 = s[J::CH_FIRST + k1], z2 = s[J::CH_PAIR + p2], z3 = s[J::CH_3 + k3], z4 = s[J::CH_4 + k], g4 = g[J::CH_4 + k];
        const V z11 = z1 * z1;
        g[J::CH_4 + k] = d[1] * g4;
        z3b[k3] = vfma(V(4.0f) * d[2] * z1, g4, z3b[k3]);
        zp[p2] = vfma(V(6.0f) * vfma(d[3], z11, d[2] * z2), g4, zp[p2]);
        zf[k1] = vfma(V(4.0f) * vfma(d[4] * z11, z1, vfma(V(3.0f) * d[3] * z1, z2, d[2] * z3)), g4, zf[k1]);
        V t = d[2] * z4;
        t = vfma(V(4.0f) * d[3] * z1, z3, t);
        t = vfma(V(3.0f) * d[3] * z2, z2, t);
        t = vfma(V(6.0f) * d[4] * z11, z2, t);
        t = vfma(d[5] * z11, z11, t);
        zv = vfma(t, g4, zv);
    }
    g[0] = zv;
    PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf) g[J::CH_FIRST + kf] = zf[kf];
    PINN_UNROLL for (int p = 0; p < J::NPAIR; ++p) g[J::CH_PAIR + p] = zp[p];
    PINN_UNROLL for (int k = 0; k < J::N3; ++k) g[J::CH_3 + k] = z3b[k];
}

template <bool SINACT, bool MIXED = false, class V = vfloat>
DEV V act_value(int act, V z) {
    if (MIXED) {
        const float m = 2.0f - (float)act;
        return vfma(vsigmoid_fast(V(m) * z), V(m), V(1.0f - m));
    }
    if (SINACT) { V sn, cs; vsincos(z, sn, cs); return sn; }
    if (act == ACT_TANH) return vtanh_fast(z);
    return vsigmoid_fast(z);
}
// the four elements of an accumulator fragment at once (tanh: the pairwise form, vec.hpp vtanh_fast4 — same bits)
template <bool SINACT>
DEV vfloat4 act_value4(int act, vfloat4 z) {
    if (!SINACT && act == ACT_TANH) return vtanh_fast4(z);
    vfloat4 a;
    PINN_UNROLL for (int r = 0; r < 4; ++r) a[r] = act_value<SINACT>(act, z[r]);
    return a;
}
// record value r0 of an element with pre-activation z and activation a, and the activation back from r0
template <bool SINACT, class V> DEV V act_record(V z, V a) { return SINACT ? z : a; }
template <bool SINACT, class V>
DEV V act_from_record(V r0) {
    if (!SINACT) return r0;
    V sn, cs;
    vsincos(r0, sn, cs);
    return sn;
}

// XOR-swizzled address inside a [16 columns][HP] transpose buffer: neuron n of column `col`
template <class S>
DEV vint tr_addr(vint col, vint slot) {
    // slots of 4 floats; 4*MT slots per row
    return col * S::HP + (((slot ^ col) & (4 * S::MT - 1)) << 2);
}

// ------------------------------------------------------------------------------------------------
// The wave program.  Workgroup `blk` of `nblocks` (persistent grid), wave `w` (0..3) of the workgroup.
// All four waves of a workgroup run the same number of tile iterations (tiles past the end are fully masked
// dummies) because the COOP dW phase synchronises them with workgroup barriers.
// ------------------------------------------------------------------------------------------------
// WTHRU (training kernel, pinn_train.hpp): what other workgroups read inside the launch — gradient slabs, loss partials — is stored write-through,
// and the weight image, which other workgroups rewrote, is read with agent-scope loads: no fences around the grid barriers
template <class S, int MODE, int ACTK, bool WTHRU = false>
DEV void wave_main(const GroupArgs& ga, int blk, int nblocks, int w, float* lds_wg) {
    const int wave = blk * 4 + w;
    float* lds = lds_wg + S::LDS_SHARED + w * S::LDS_PRIV;     // wave-private LDS
    float* lds_sh = lds_wg;                                    // workgroup-shared chunk buffers (COOP)
    constexpr bool BWD = (MODE == MODE_FUSED || MODE == MODE_GRADIN);       // modes that run the reverse sweep
    constexpr bool COOP = S::COOP && BWD;
    constexpr int WT = S::WT;
    constexpr int HP = S::HP, MT = S::MT, NHH = S::NHH, LH = S::LH, D = S::D, C = S::C, PG = S::PG, NG = S::NG;
    constexpr int NFIRST = S::NFIRST;
    using J = typename S::J;
    const vint lane = lane_id();
    const vint g = lane >> 4;
    const vint c = lane & vint(15);
    const vbool g0 = veq(g, 0);
    const float* P = ga.packed;
    // the activation kind is a template parameter: every kernel is straight-line code behind its GEMMs (no activation branches for the
    // optimiser to hoist); sin variants are compiled only for the specs registered with PINN_INSTANTIATE*_SIN
    constexpr bool SINACT = (ACTK == ACT_SIN);
    constexpr bool MIXED = (ACTK == ACT_MIXED);             // tanh / sigmoid per hidden layer (GroupArgs::act: one nibble per layer), branch-free
    auto act_of = [&](int layer) -> int { return MIXED ? ((ga.act_layers >> (4 * layer)) & 15) : ACTK; };

    // ---- persistent per-wave gradient accumulators (registers / AGPRs across all tiles) ----
    vfloat4 wbar[NHH > 0 ? NHH : 1][WT][MT];      // COOP: this wave's row block (to == w) only
    vfloat bfrh[NHH > 0 ? NHH : 1][WT];           // bias grads of hidden layers >= 1 (fragment form)
    vfloat bfr0[MT];                              // bias grad of hidden layer 0
    vfloat w1fr[D][MT];
    vfloat4 wLbar[MT];
    vfloat bLbar = vfloat(0.f);
    vfloat pbar[MAX_PARAMS];
    PINN_UNROLL for (int l = 0; l < (NHH > 0 ? NHH : 1); ++l)
        PINN_UNROLL for (int a = 0; a < WT; ++a) {
            bfrh[l][a] = vfloat(0.f);
            PINN_UNROLL for (int b = 0; b < MT; ++b) wbar[l][a][b] = vzero4();
        }
    PINN_UNROLL for (int a = 0; a < MT; ++a) bfr0[a] = vfloat(0.f);
    PINN_UNROLL for (int i = 0; i < D; ++i)
        PINN_UNROLL for (int a = 0; a < MT; ++a) w1fr[i][a] = vfloat(0.f);
    PINN_UNROLL for (int a = 0; a < MT; ++a) wLbar[a] = vzero4();
    PINN_UNROLL for (int i = 0; i < MAX_PARAMS; ++i) pbar[i] = vfloat(0.f);

    vdacc lsum = vdacc_zero();     // per-lane double sums of squared residuals
    int cur_term = -1;      // index into ga.terms of the term whose loss is being accumulated
    int cq = 0;             // running chunk counter: parity selects the shared LDS chunk buffer (COOP)

    const ubuf SB = ub_make(ga.scratch + (size_t)wave * S::SCR, S::SCR);      // this wave's activation scratch
    const ubuf PB = ub_make(P, S::PACKED);                                         // packed weights of this net
    auto pld = [&](int soff, vint voff) -> vfloat { return WTHRU ? ub_load_sc1(PB, soff, voff) : ub_load(PB, soff, voff); };
    auto pld4 = [&](int soff, vint voff) -> vfloat4 { return WTHRU ? ub_load4_sc1(PB, soff, voff) : ub_load4(PB, soff, voff); };
    auto lp_store = [&](int term_id, double v) {                                   // this wave's loss-partial column of a term
        double* q = ga.losspart + (size_t)wave * ga.nterms_total + term_id;
        if (WTHRU) ustore_wt(q, v); else *q = v;
    };
    float* xs = lds + 4 * S::LDS_T;           // coords of the tile: [pg][pt][i]

    // output layer weights in D layout
    vfloat4 wL[MT];
    PINN_UNROLL for (int m = 0; m < MT; ++m) wL[m] = pld4(S::OFF_WL + 16 * m, g << 2);
    const float bL = lane0(pld(S::OFF_BL, lane & vint(0)));      // (a vector load: the training kernel rewrites the image inside the launch)
    // register-resident weights (Spec::W_RESIDENT): [layer][k-step][tile] forward / transposed fragments, biases, first layer
    constexpr bool WRES = S::W_RESIDENT, RRES = S::R_RESIDENT && BWD;
    vfloat wres[WRES ? NHH : 1][WRES ? 4 * MT : 1][MT], tres[(WRES && BWD) ? NHH : 1][(WRES && BWD) ? 4 * MT : 1][MT];
    vfloat4 bres[WRES ? LH : 1][MT], w1res[WRES ? D : 1][MT];
    if (WRES) {
        PINN_UNROLL for (int l = 0; l < LH; ++l)
            PINN_UNROLL for (int m = 0; m < MT; ++m) bres[l][m] = pld4(S::OFF_B + l * HP + 16 * m, g << 2);
        PINN_UNROLL for (int i = 0; i < D; ++i)
            PINN_UNROLL for (int m = 0; m < MT; ++m) w1res[i][m] = pld4(S::OFF_W1 + i * HP + 16 * m, g << 2);
        PINN_UNROLL for (int hl = 0; hl < NHH; ++hl)
            PINN_UNROLL for (int ks = 0; ks < 4 * MT; ++ks)
                PINN_UNROLL for (int mo = 0; mo < MT; ++mo) {
                    wres[hl][ks][mo] = pld(S::OFF_WPK + hl * HP * HP + ks * 64 * MT + mo, lane * MT);
                    if (BWD) tres[hl][ks][mo] = pld(S::OFF_WTPK + hl * HP * HP + ks * 64 * MT + mo, lane * MT);
                }
    }

    constexpr bool SUMS = (MODE == MODE_FUSED || MODE == MODE_LOSS);      // modes that deliver the per-term sums of squares
    if (SUMS)                    // this wave's loss columns start at zero (no host-side memset per evaluation)
        for (int j = 0; j < ga.nterms; ++j) lp_store(ga.terms[j].term_id, 0.0);
    const int niter = (ga.ntiles + 4 * nblocks - 1) / (4 * nblocks);
    for (int it = 0; it < niter; ++it) {
        const int t = (it * nblocks + blk) * 4 + w;          // >= ntiles: dummy tile of the last term, all points masked
        // ---- locate the term of this tile (terms are tile-contiguous) ----
        int k = 0;
        for (int j = 1; j < ga.nterms; ++j)
            if (t >= ga.terms[j].tile0) k = j;
        if (k != cur_term) {
            if (cur_term >= 0) {
                double s = wave_sum_dd(lsum, g0);
                if (SUMS) lp_store(ga.terms[cur_term].term_id, s);
            }
            lsum = vdacc_zero();
            cur_term = k;
        }
        const TermDev& T = ga.terms[k];
        const int pbase = (t - T.tile0) * S::TP;

        // ---- coordinates ----
        vfloat x[PG][D];
        vbool valid[PG];
        PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
            vint p = vint(pbase + 16 * pg) + c;
            valid[pg] = vlt(p, T.N);
            PINN_UNROLL for (int i = 0; i < D; ++i) {
                x[pg][i] = gload_masked(T.pts, p * T.dt + vint(T.imap[i]), valid[pg]);
                if (BWD) lds_store(xs, (vint(16 * pg) + c) * D + vint(i), x[pg][i]);
            }
        }

        // =========================== forward Taylor-jet sweep ===========================
        vfloat4 A[NG][MT];
        // ---- layer 1: d -> HP on the VALU (K = d is tiny) ----
        PINN_UNROLL for (int m = 0; m < MT; ++m) {
            vfloat4 b1 = WRES ? bres[0][m] : pld4(S::OFF_B + 16 * m, g << 2);
            vfloat4 w1[D];
            PINN_UNROLL for (int i = 0; i < D; ++i) w1[i] = WRES ? w1res[i][m] : pld4(S::OFF_W1 + i * HP + 16 * m, g << 2);
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                vfloat4 z = b1;
                PINN_UNROLL for (int i = 0; i < D; ++i)
                    PINN_UNROLL for (int r = 0; r < 4; ++r) z[r] = vfma(w1[i][r], x[pg][i], z[r]);
                A[pg * C][m] = z;
                PINN_UNROLL for (int kf = 0; kf < NFIRST; ++kf) A[pg * C + 1 + kf][m] = w1[S::first_axis(kf)];
                PINN_UNROLL for (int ch = 1 + NFIRST; ch < C; ++ch) A[pg * C + ch][m] = vzero4();       // second and higher derivatives of an affine map
            }
        }

        // raw (a, z_i, z_ij) record of the LAST hidden layer stays in registers for the reverse sweep
        vfloat4 Rlast[NG][MT];
        vfloat4 Rec[RRES ? NHH : 1][NG][MT];           // Spec::R_RESIDENT: the records of the hidden layers below the last, in registers
        // activation jets in place + park (a, z_i, z_ij) of the other layers in the scratch slab
        auto act_forward = [&](vfloat4 (&Z)[NG][MT], int layer /*0-based hidden layer*/) {
            const int act = act_of(layer);
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int m = 0; m < MT; ++m) {
                    vfloat4 av;                                       // sin: activation values; Z[pg*C] holds the RECORD value z meanwhile
                    PINN_UNROLL for (int r = 0; r < 4; ++r) {
                        const vfloat z0 = Z[pg * C][m][r];
                        av[r] = act_value<SINACT, MIXED>(act, z0);
                        Z[pg * C][m][r] = SINACT ? z0 : av[r];
                    }
                    if (BWD) {
                        if (layer == LH - 1) {
                            PINN_UNROLL for (int ch = 0; ch < C; ++ch) Rlast[pg * C + ch][m] = Z[pg * C + ch][m];
                        } else if (RRES) {
                            PINN_UNROLL for (int ch = 0; ch < C; ++ch) Rec[layer][pg * C + ch][m] = Z[pg * C + ch][m];
                        } else {
                            PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                                ub_store4(SB, (((layer * NG) + pg * C + ch) * MT + m) * 256, lane << 2, Z[pg * C + ch][m]);
                            store_pad();                      // the stored channels are updated in place below (vec.hpp: store_pad)
                            PINN_UNROLL for (int ch = 0; ch < C; ++ch) keep_alive(Z[pg * C + ch][m]);
                            sched_fence();
                        }
                    }
                    PINN_UNROLL for (int r = 0; r < 4; ++r) {
                        vfloat zz[C], dd[ND];
                        PINN_UNROLL for (int ch = 0; ch < C; ++ch) zz[ch] = Z[pg * C + ch][m][r];
                        act_derivs_n<J::NORD - 1, SINACT, MIXED>(act, zz[0], dd);
                        jet_forward<J>(zz, dd);
                        PINN_UNROLL for (int ch = 1; ch < C; ++ch) Z[pg * C + ch][m][r] = zz[ch];
                    }
                    if (SINACT) Z[pg * C][m] = av;
                }
        };
        // k-steps (mi, rr) of hidden layer hl's GEMM; the fragment of step ks+1 is requested before the MFMAs of step ks (software pipeline:
        // with one wave per SIMD nothing else hides the L2 latency of a just-in-time load)
        auto load_wf = [&](int hl, int ks, vfloat (&wf)[MT]) {
            const int Wf = S::OFF_WPK + hl * HP * HP;
            if (MT == 4) {
                vfloat4 w4 = pld4(Wf + ks * 64 * MT, lane << 2);
                PINN_UNROLL for (int mo = 0; mo < MT; ++mo) wf[mo] = w4[mo & 3];
            } else {
                PINN_UNROLL for (int mo = 0; mo < MT; ++mo) wf[mo] = pld(Wf + ks * 64 * MT + mo, lane * MT);
            }
        };
        // PINN_F1_PREFETCH: a layer's bias and its FIRST weight fragment are requested before the activation function of the layer below
        // runs (a lone wave per SIMD otherwise sits out one L2 round trip at the top of every layer: most of a one-tile launch's time)
        vfloat wfirst[MT];
        vfloat4 bvn[MT];
        auto prefetch_layer = [&](int hl) {
            if (WRES) {
                PINN_UNROLL for (int m = 0; m < MT; ++m) bvn[m] = bres[hl + 1][m];
                return;
            }
            PINN_UNROLL for (int m = 0; m < MT; ++m) bvn[m] = pld4(S::OFF_B + (hl + 1) * HP + 16 * m, g << 2);
            load_wf(hl, 0, wfirst);
        };
        if (PINN_F1_PREFETCH && NHH > 0) prefetch_layer(0);
        act_forward(A, 0);

        // ---- hidden -> hidden layers on the matrix cores ----
        PINN_UNROLL for (int hl = 0; hl < NHH; ++hl) {
            vfloat4 Zn[NG][MT];
            if (!PINN_F1_PREFETCH) prefetch_layer(hl);
            PINN_UNROLL for (int m = 0; m < MT; ++m) {
                PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                    Zn[pg * C][m] = bvn[m];
                    PINN_UNROLL for (int ch = 1; ch < C; ++ch) Zn[pg * C + ch][m] = vzero4();
                }
            }
            vfloat wcur[MT], wnxt[MT];
            if (!WRES) { PINN_UNROLL for (int mo = 0; mo < MT; ++mo) wcur[mo] = wfirst[mo]; }
            PINN_UNROLL for (int ks = 0; ks < 4 * MT; ++ks) {
                const int mi = ks >> 2, rr = ks & 3;
                if (!WRES && ks + 1 < 4 * MT) load_wf(hl, ks + 1, wnxt);
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int mo = 0; mo < MT; ++mo) Zn[q][mo] = mfma16(WRES ? wres[hl][WRES ? ks : 0][mo] : wcur[mo], A[q][mi][rr], Zn[q][mo]);
                if (!WRES) { PINN_UNROLL for (int mo = 0; mo < MT; ++mo) wcur[mo] = wnxt[mo]; }
            }
            if (PINN_F1_PREFETCH && hl + 1 < NHH) prefetch_layer(hl + 1);
            act_forward(Zn, hl + 1);
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int m = 0; m < MT; ++m) A[q][m] = Zn[q][m];
        }

        // ---- output layer HP -> 1 (identity): per-lane partial dot + reduction over the 4 row groups ----
        vfloat U[PG][C];
        PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
            PINN_UNROLL for (int ch = 0; ch < C; ++ch) {
                vfloat s = vfloat(0.f);
                PINN_UNROLL for (int m = 0; m < MT; ++m)
                    PINN_UNROLL for (int r = 0; r < 4; ++r) s = vfma(wL[m][r], A[pg * C + ch][m][r], s);
                s = xrow_allsum(s);
                U[pg][ch] = (ch == 0) ? s + vfloat(bL) : s;
            }

        if (MODE == MODE_FWD) {
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                vint p = vint(pbase + 16 * pg) + c;
                PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                    gstore_masked(T.out, vint(ch * T.N) + p, U[pg][ch], vand(valid[pg], g0));
            }
            continue;
        }

        auto load_raw = [&](vfloat4 (&Sr)[NG][MT], int layer) {
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int m = 0; m < MT; ++m)
                    Sr[q][m] = RRES ? Rec[RRES ? layer : 0][q][m] : ub_load4(SB, (((layer * NG) + q) * MT + m) * 256, lane << 2);
        };
        // prefetch the raw record of the layer feeding the last hidden->hidden GEMM: it lands while the tape runs
        vfloat4 SrN[NG][MT];
        if (BWD && NHH > 0) load_raw(SrN, NHH - 1);

        // =========================== residual tape (values, then adjoints) ===========================
        vfloat ubar[PG][C];
        if (MODE == MODE_GRADIN) {
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                vint p = vint(pbase + 16 * pg) + c;
                PINN_UNROLL for (int ch = 0; ch < C; ++ch) ubar[pg][ch] = gload_masked(T.in, vint(ch * T.N) + p, valid[pg]);
            }
        } else if (T.linear && C <= LIN_MAX_C) {
            // affine residual (r04, as in the neuron-split kernels): r = k + sum_c a_c U_c + sum_j b_j src_j with constant coefficients —
            // every Dirichlet term, lap u - f, linear PDEs with fixed coefficients (plan.cpp: detect_linear): C + nsrc FMAs and the seeds
            // are the coefficients; no interpreter, no LDS rows
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                const vint p = vint(pbase + 16 * pg) + c;
                vfloat r = vfloat(T.lin_k);
                PINN_UNROLL for (int ch = 0; ch < C; ++ch) r = vfma(vfloat(T.lin_a[ch < LIN_MAX_C ? ch : 0]), U[pg][ch], r);
                PINN_UNROLL for (int j = 0; j < LIN_MAX_SRC; ++j)
                    if (j < T.nsrc) r = vfma(vfloat(T.lin_b[j]), gload_masked(T.src, vint(j * T.N) + p, valid[pg]), r);
                if (MODE == MODE_RESID) {
                    gstore_masked(T.out, p, r, vand(valid[pg], g0));
                    continue;
                }
                vfloat sw = vfloat(1.0f);
                if (T.pw) sw = gload_masked(T.pw, p, valid[pg]);
                const vfloat rm = vselect(valid[pg], r * sw, vfloat(0.f));
                lsum = vdacc_fma(rm, vselect(g0, rm, vfloat(0.f)), lsum);
                if (MODE == MODE_LOSS) continue;
                const vfloat rbar = rm * vfloat(T.scale) * sw;
                PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                    ubar[pg][ch] = vselect(valid[pg], rbar * vfloat(T.lin_a[ch < LIN_MAX_C ? ch : 0]), vfloat(0.f));
            }
        } else {
            wave_fence();
            float* tv = lds;                    // value rows  [row][64]
            float* ta = lds + 2 * S::LDS_T;     // adjoint rows
            const int NP = ga.nparams;
            const int DT = T.dt;                // tape rows: [coordinates DT | params NP | jet channels C | sources | ops]
            const int R0 = DT + NP + C + T.nsrc;
            const rp::Instr* prog = ga.prog + T.prog_off;
            const int nrows = R0 + T.nops;
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                if (!T.hetero) {
                    PINN_UNROLL for (int i = 0; i < D; ++i) lds_store(tv, vint(i * 64) + lane, x[pg][i]);
                } else {
                    for (int j = 0; j < DT; ++j)
                        lds_store(tv, vint(j * 64) + lane, gload_masked(T.pts, (vint(pbase + 16 * pg) + c) * DT + vint(j), valid[pg]));
                }
                for (int j = 0; j < NP; ++j) lds_store(tv, vint((DT + j) * 64) + lane, vfloat(ga.params[j]));
                PINN_UNROLL for (int ch = 0; ch < C; ++ch) lds_store(tv, vint((DT + NP + ch) * 64) + lane, U[pg][ch]);
                for (int j = 0; j < T.nsrc; ++j)
                    lds_store(tv, vint((DT + NP + C + j) * 64) + lane, gload_masked(T.src, vint(j * T.N + pbase + 16 * pg) + c, valid[pg]));
                for (int q = 0; q < T.nops; ++q) {
                    const rp::Instr ins = rp::fetch_uniform(prog, q);
                    vfloat va = lds_load(tv, vint(ins.a * 64) + lane);            // unused operands point at row 0
                    vfloat vb = lds_load(tv, vint(ins.b * 64) + lane);
                    vfloat vo;
                    if (rp::is_bilinear(ins.code)) vo = rp::apply_bilinear<vfloat>(ins, va, vb);
                    else vo = rp::apply<vfloat>(ins.code, va, vb, ins.imm);
                    lds_store(tv, vint((R0 + q) * 64) + lane, vo);
                }
                vfloat r = lds_load(tv, vint(T.out_row * 64) + lane);
                if (MODE == MODE_RESID) {
                    vint p = vint(pbase + 16 * pg) + c;
                    gstore_masked(T.out, p, r, vand(valid[pg], g0));
                    continue;
                }
                vfloat sw = vfloat(1.0f);
                if (T.pw) sw = gload_masked(T.pw, vint(pbase + 16 * pg) + c, valid[pg]);
                vfloat rm = vselect(valid[pg], r * sw, vfloat(0.f));
                lsum = vdacc_fma(rm, vselect(g0, rm, vfloat(0.f)), lsum);
                if (MODE == MODE_LOSS) continue;
                vfloat rbar = rm * vfloat(T.scale) * sw;
                for (int q = 0; q < nrows; ++q) lds_store(ta, vint(q * 64) + lane, vfloat(0.f));
                lds_store(ta, vint(T.out_row * 64) + lane, vfloat(1.0f));
                for (int q = T.nops - 1; q >= 0; --q) {
                    const rp::Instr ins = rp::fetch_uniform(prog, q);
                    vfloat gq = lds_load(ta, vint((R0 + q) * 64) + lane);
                    vfloat va = lds_load(tv, vint(ins.a * 64) + lane);
                    vfloat vb = lds_load(tv, vint(ins.b * 64) + lane);
                    vfloat da, db;
                    if (rp::is_bilinear(ins.code)) rp::adjoint_bilinear<vfloat>(ins, va, vb, gq, da, db);      // unused operands: zero adjoint into row 0
                    else {
                        vfloat vo = lds_load(tv, vint((R0 + q) * 64) + lane);
                        rp::adjoint<vfloat>(ins.code, va, vb, vo, ins.imm, gq, da, db);
                    }
                    lds_store(ta, vint(ins.a * 64) + lane, lds_load(ta, vint(ins.a * 64) + lane) + da);
                    lds_store(ta, vint(ins.b * 64) + lane, lds_load(ta, vint(ins.b * 64) + lane) + db);
                }
                PINN_UNROLL for (int ch = 0; ch < C; ++ch)       // masked points may hold inf/NaN (their sources read as 0)
                    ubar[pg][ch] = vselect(valid[pg], rbar * lds_load(ta, vint((DT + NP + ch) * 64) + lane), vfloat(0.f));
                for (int j = 0; j < ga.nparams_estim; ++j) {
                    vfloat pj = vselect(vand(g0, valid[pg]), rbar * lds_load(ta, vint((DT + j) * 64) + lane), vfloat(0.f));
                    PINN_UNROLL for (int jj = 0; jj < MAX_PARAMS; ++jj) if (jj == j) pbar[jj] += pj;
                }
            }
            wave_fence();
        }
        if (MODE == MODE_RESID || MODE == MODE_LOSS) continue;

        // =========================== reverse sweep ===========================
        // post-activation jet of channel ch from the raw (a, z_i, z_ij) record
        auto ajet = [&](const vfloat4 (&Sr)[NG][MT], int pg, int ch, int m, int act) -> vfloat4 {
            vfloat4 out;
            PINN_UNROLL for (int r = 0; r < 4; ++r) {
                vfloat zz[C], dd[ND];
                PINN_UNROLL for (int k = 0; k < C; ++k) zz[k] = Sr[pg * C + k][m][r];
                if (ch > 0) {
                    act_derivs_n<J::NORD - 1, SINACT, MIXED>(act, zz[0], dd);
                    jet_forward<J>(zz, dd);                 // (ch is a constant after unrolling: the other channels are dead code)
                }
                out[r] = (ch == 0) ? act_from_record<SINACT>(zz[0]) : zz[ch];
            }
            return out;
        };
        // activation adjoint in place: G (dA jets) -> dZ jets
        auto act_adjoint = [&](vfloat4 (&G)[NG][MT], const vfloat4 (&Sr)[NG][MT], int act) {
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int m = 0; m < MT; ++m)
                    PINN_UNROLL for (int r = 0; r < 4; ++r) {
                        vfloat gg[C], ss[C], dd[ND];
                        PINN_UNROLL for (int k = 0; k < C; ++k) { gg[k] = G[pg * C + k][m][r]; ss[k] = Sr[pg * C + k][m][r]; }
                        act_derivs_n<J::NORD, SINACT, MIXED>(act, ss[0], dd);
                        jet_adjoint<J>(gg, ss, dd);
                        PINN_UNROLL for (int k = 0; k < C; ++k) G[pg * C + k][m][r] = gg[k];
                    }
        };

        vfloat4 G[NG][MT];     // adjoint chain (dA, then dZ in place)
        {
            // output layer: dW_out += sum ubar * a_jets (A still holds the last hidden layer's jets); dA = w_out * ubar
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                bLbar += vselect(g0, ubar[pg][0], vfloat(0.f));
                PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                    PINN_UNROLL for (int m = 0; m < MT; ++m) {
                        PINN_UNROLL for (int r = 0; r < 4; ++r) {
                            wLbar[m][r] = vfma(ubar[pg][ch], A[pg * C + ch][m][r], wLbar[m][r]);
                            G[pg * C + ch][m][r] = wL[m][r] * ubar[pg][ch];
                        }
                    }
            }
            act_adjoint(G, Rlast, act_of(LH - 1));
        }

        PINN_UNROLL for (int hl = NHH - 1; hl >= 0; --hl) {
            // layer (hl+1) -> (hl+2) weights; inputs are hidden layer hl's a-jets (record prefetched into SrN)
            const int act = act_of(hl);
            vfloat4 Sr[NG][MT];
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int m = 0; m < MT; ++m) Sr[q][m] = SrN[q][m];
            // W^T fragments of the dA GEMM below; the first one is requested here (PINN_F1_PREFETCH): its L2 round trip hides under the dW staging
            const int Wt = S::OFF_WTPK + hl * HP * HP;
            auto load_wt = [&](int ks, vfloat (&wf)[MT]) {
                if (MT == 4) {
                    vfloat4 w4 = pld4(Wt + ks * 64 * MT, lane << 2);
                    PINN_UNROLL for (int mi = 0; mi < MT; ++mi) wf[mi] = w4[mi & 3];
                } else {
                    PINN_UNROLL for (int mi = 0; mi < MT; ++mi) wf[mi] = pld(Wt + ks * 64 * MT + mi, lane * MT);
                }
            };
            vfloat tcur[MT];
            if (PINN_F1_PREFETCH && !WRES) load_wt(0, tcur);
            // ---- dW += dZ A^T through the swizzled LDS transpose, 16 columns at a time ----
            if (COOP) {
                // Every wave publishes its (dZ^T, A^T) chunk in the workgroup-shared double buffer; after one barrier
                // each wave accumulates ITS 16-row block of dW (out neurons 4i+w) over the chunks of all four waves,
                // in the fixed order ws = 0..3 => deterministic, and only MT accumulator tiles per layer stay resident.
                PINN_UNROLL for (int q = 0; q < NG; ++q) {
                    float* cb = lds_sh + (cq & 1) * (4 * 2 * S::LDS_T);
                    ++cq;
                    float* myzt = cb + w * 2 * S::LDS_T;
                    float* myat = myzt + S::LDS_T;
                    const int pg = q / C, ch = q % C;
                    PINN_UNROLL for (int m = 0; m < MT; ++m) {
                        const vint ad = tr_addr<S>(c, vint(4 * m) + g);
                        lds_store4(myzt, ad, G[q][m]);
                        lds_store4(myat, ad, ajet(Sr, pg, ch, m, act));
                    }
                    wg_barrier();
                    // operands of source wave ws+1 are requested before the 16 MFMAs of source wave ws
                    vfloat zc[4], zn[4];
                    vfloat4 ac[4], an[4];
                    auto load_ops = [&](int ws, vfloat (&zf)[4], vfloat4 (&af)[4]) {
                        const float* zt = cb + ws * 2 * S::LDS_T;
                        const float* at = zt + S::LDS_T;
                        PINN_UNROLL for (int kk = 0; kk < 4; ++kk) {
                            const vint ad = tr_addr<S>(vint(4 * kk) + g, c);
                            zf[kk] = lds_load(zt, ad + vint(w));              // dZ[out neuron 4c + w][column 4kk + g]
                            af[kk] = lds_load4(at, ad);                       // A[in neurons 4c .. 4c+3][column 4kk + g]
                        }
                    };
                    load_ops(0, zc, ac);
                    PINN_UNROLL for (int ws = 0; ws < 4; ++ws) {
                        if (ws + 1 < 4) load_ops(ws + 1, zn, an);
                        PINN_UNROLL for (int kk = 0; kk < 4; ++kk) {
                            PINN_UNROLL for (int ti = 0; ti < MT; ++ti) wbar[hl][0][ti] = mfma16(zc[kk], ac[kk][ti & 3], wbar[hl][0][ti]);
                            if (ch == 0) bfrh[hl][0] += zc[kk];
                        }
                        PINN_UNROLL for (int kk = 0; kk < 4; ++kk) { zc[kk] = zn[kk]; ac[kk] = an[kk]; }
                    }
                }
            } else {
                PINN_UNROLL for (int q = 0; q < NG; ++q) {
                    float* zt = lds + (q & 1) * 2 * S::LDS_T;
                    float* at = zt + S::LDS_T;
                    const int pg = q / C, ch = q % C;
                    PINN_UNROLL for (int m = 0; m < MT; ++m) {
                        const vint ad = tr_addr<S>(c, vint(4 * m) + g);
                        lds_store4(zt, ad, G[q][m]);
                        lds_store4(at, ad, ajet(Sr, pg, ch, m, act));
                    }
                    wave_fence();
                    PINN_UNROLL for (int kk = 0; kk < 4; ++kk) {
                        const vint row = vint(4 * kk) + g;
                        vfloat zf[MT], af[MT];
                        if (MT == 4) {
                            const vint ad = tr_addr<S>(row, c);
                            vfloat4 z4 = lds_load4(zt, ad), a4 = lds_load4(at, ad);
                            PINN_UNROLL for (int to = 0; to < MT; ++to) { zf[to] = z4[to & 3]; af[to] = a4[to & 3]; }
                        } else {
                            PINN_UNROLL for (int to = 0; to < MT; ++to) {
                                const vint n = c * MT + vint(to);
                                const vint ad = tr_addr<S>(row, n >> 2) + (n & vint(3));
                                zf[to] = lds_load(zt, ad);
                                af[to] = lds_load(at, ad);
                            }
                        }
                        PINN_UNROLL for (int to = 0; to < WT; ++to)
                            PINN_UNROLL for (int ti = 0; ti < MT; ++ti) wbar[hl][to][ti] = mfma16(zf[to], af[ti], wbar[hl][to][ti]);
                        if (ch == 0) PINN_UNROLL for (int to = 0; to < WT; ++to) bfrh[hl][to] += zf[to];
                    }
                    wave_fence();
                }
            }
            // prefetch the next layer's record: its latency hides under the dA GEMM below
            if (hl > 0) load_raw(SrN, hl - 1);
            // ---- dA_prev = W^T dZ on the matrix cores (register-chained) ----
            vfloat4 Gn[NG][MT];
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int m = 0; m < MT; ++m) Gn[q][m] = vzero4();
            vfloat tnxt[MT];
            if (!PINN_F1_PREFETCH && !WRES) load_wt(0, tcur);
            PINN_UNROLL for (int ks = 0; ks < 4 * MT; ++ks) {
                const int mo = ks >> 2, rr = ks & 3;
                if (!WRES && ks + 1 < 4 * MT) load_wt(ks + 1, tnxt);
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int mi = 0; mi < MT; ++mi)
                        Gn[q][mi] = mfma16((WRES && BWD) ? tres[(WRES && BWD) ? hl : 0][(WRES && BWD) ? ks : 0][mi] : tcur[mi], G[q][mo][rr], Gn[q][mi]);
                if (!WRES) { PINN_UNROLL for (int mi = 0; mi < MT; ++mi) tcur[mi] = tnxt[mi]; }
            }
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int m = 0; m < MT; ++m) G[q][m] = Gn[q][m];
            act_adjoint(G, Sr, act);
        }

        // ---- layer 1 (d -> HP): dW1, db1 from the transposed dZ fragments ----
        PINN_UNROLL for (int q = 0; q < NG; ++q) {
            const int pg = q / C, ch = q % C;
            if (ch > NFIRST) continue;     // second-order channels of layer 1 do not depend on W1 (z_ij = 0)
            float* zt = lds + (q & 1) * 2 * S::LDS_T;
            PINN_UNROLL for (int m = 0; m < MT; ++m) lds_store4(zt, tr_addr<S>(c, vint(4 * m) + g), G[q][m]);
            wave_fence();
            PINN_UNROLL for (int kk = 0; kk < 4; ++kk) {
                const vint row = vint(4 * kk) + g;
                vfloat zf[MT];
                if (MT == 4) {
                    vfloat4 z4 = lds_load4(zt, tr_addr<S>(row, c));
                    PINN_UNROLL for (int to = 0; to < MT; ++to) zf[to] = z4[to & 3];
                } else {
                    PINN_UNROLL for (int to = 0; to < MT; ++to) {
                        const vint n = c * MT + vint(to);
                        zf[to] = lds_load(zt, tr_addr<S>(row, n >> 2) + (n & vint(3)));
                    }
                }
                if (ch == 0) {
                    PINN_UNROLL for (int to = 0; to < MT; ++to) bfr0[to] += zf[to];
                    PINN_UNROLL for (int i = 0; i < D; ++i) {
                        vfloat xc = lds_load(xs, (vint(16 * pg) + row) * D + vint(i));
                        PINN_UNROLL for (int to = 0; to < MT; ++to) w1fr[i][to] = vfma(zf[to], xc, w1fr[i][to]);
                    }
                } else {
                    const int axis = S::first_axis(ch - 1);
                    PINN_UNROLL for (int i = 0; i < D; ++i)
                        if (i == axis) PINN_UNROLL for (int to = 0; to < MT; ++to) w1fr[i][to] += zf[to];
                }
            }
            wave_fence();
        }
    }  // tiles

    if (MODE == MODE_LOSS && cur_term >= 0) lp_store(ga.terms[cur_term].term_id, wave_sum_dd(lsum, g0));
    if (!BWD) return;

    // =========================== epilogue: gradient slab ===========================
    if (cur_term >= 0) {
        double s = wave_sum_dd(lsum, g0);
        lp_store(ga.terms[cur_term].term_id, s);
    }
    // per-workgroup slab: [shared section | 4 x per-wave section]; COOP: dW / hidden-bias rows are written by their
    // owner wave into the shared section, everything else (and everything for small nets) is per wave.
    float* slab_wg = ga.slabs + (size_t)blk * S::SLAB;
    float* mine = slab_wg + S::SH + w * S::PW;
    float* big = S::COOP ? slab_wg : mine;
    const ubuf SLW = ub_make(slab_wg, S::SLAB);
    auto st4 = [&](float* p, vint i, const vfloat4& x) { if (WTHRU) ub_store4_wt(SLW, vint((int)(p - slab_wg)) + i, x); else gstore4(p, i, x); };
    auto stm = [&](float* p, vint i, vfloat x, vbool m) { if (WTHRU) gstore_masked_wt(p, i, x, m); else gstore_masked(p, i, x, m); };
    PINN_UNROLL for (int hl = 0; hl < NHH; ++hl)
        PINN_UNROLL for (int to = 0; to < WT; ++to) {
            const int trow = S::COOP ? w : to;
            PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                st4(big + S::O_WBAR + hl * HP * HP, vint((trow * MT + ti) * 256) + (lane << 2), wbar[hl][to][ti]);
            vfloat v = bfrh[hl][to];
            v = xrow_allsum(v);
            stm(big + S::O_BFRH, vint((hl * MT + trow) * 16) + c, v, g0);
        }
    PINN_UNROLL for (int to = 0; to < MT; ++to) {
        vfloat v = bfr0[to];
        v = xrow_allsum(v);
        stm(mine + S::O_BFR0, vint(to * 16) + c, v, g0);
    }
    PINN_UNROLL for (int i = 0; i < D; ++i)
        PINN_UNROLL for (int to = 0; to < MT; ++to) {
            vfloat v = w1fr[i][to];
            v = xrow_allsum(v);
            stm(mine + S::O_W1, vint((i * MT + to) * 16) + c, v, g0);
        }
    const vbool c0 = veq(c, 0);
    PINN_UNROLL for (int m = 0; m < MT; ++m)
        PINN_UNROLL for (int r = 0; r < 4; ++r) {
            vfloat v = wLbar[m][r];
            v = row_allsum16(v);
            stm(mine + S::O_WL, ((vint(m * 4) + g) << 2) + vint(r), v, c0);
        }
    {
        vbool all = vlt(lane, 64);
        float s = (float)wave_sum_d(bLbar, all);
        stm(mine + S::O_BL, vint(0), vfloat(s), veq(lane, 0));
        PINN_UNROLL for (int j = 0; j < MAX_PARAMS; ++j) {
            // (no estimated PDE parameters: the sums are zero — four 64-lane double reductions, 48 dependent cross-lane moves, off a one-tile launch)
            float sp = ga.nparams_estim > 0 ? (float)wave_sum_d(pbar[j], all) : 0.f;
            stm(mine + S::O_P, vint(j), vfloat(sp), veq(lane, 0));
        }
    }
}

}  // namespace pk
