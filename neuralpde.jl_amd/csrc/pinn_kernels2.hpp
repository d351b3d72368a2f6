// pinn_kernels2.hpp — "family 2" of the PINN residual/loss kernel: neuron-split workgroups.
//
// Same mathematics, same replaced reference code (see pinn_kernels.hpp header) — different mapping onto the CU:
//   * one workgroup (NW = 4 waves; 8 at H = 128 for NG <= 6 column groups: one workgroup per CU either way, so 8 waves = 2 per SIMD) owns one tile of
//     TP = 16*PG points; wave w owns the 16-neuron tiles {w*MTW .. w*MTW+MTW-1} of EVERY layer (MTW = HP/(16 NW)), for all
//     NG = C*PG column groups of the tile;
//   * between layers the activation jets are exchanged through LDS in MFMA-B-fragment order
//     X[column group][neuron tile][lane][4] (ping-pong buffers, one s_barrier per layer); a wave's accumulators are
//     only NG*MTW vfloat4 (20 registers for the 2-D Poisson interior term) instead of 2 x 80, so the kernel stays under
//     256 VGPR+AGPR and TWO workgroups (8 waves = 2 per SIMD) are resident per CU: one wave's VALU/LDS/HBM latency
//     hides under the other's MFMA issue (family 1 runs a single 512-register wave per SIMD and overlaps nothing);
//   * dW rows are naturally wave-owned (its neuron tiles x all inputs): MTW*MT accumulator tiles per layer stay resident
//     across all tiles of the workgroup; the transposed operands go through LDS: dZ^T wave-private, A^T cooperatively;
//   * the residual tape runs in vector registers (vtape, VGPR-index mode): wave pg interprets it for point group pg and broadcasts the
//     seeds through LDS;
//   * the activation kind (tanh / sigmoid / sin) is a template parameter: straight-line code behind every GEMM.
#pragma once
#include "pinn_kernels.hpp"

// Profiling build only (tools/stamp_profile.sh): per-wave cycle counters per phase of a tile, written to the tail of the
// workgroup's gradient slab (beyond the reduced entries).  Never defined for the product or the emulation library.
#if defined(PINN_STAMP) && !defined(PINN_EMU)
#define STAMP_MEMBERS unsigned st_acc[24]; unsigned long long st_last;
#define STAMP_INIT(ac) { for (int i_ = 0; i_ < 24; ++i_) (ac).st_acc[i_] = 0; (ac).st_last = __builtin_amdgcn_s_memtime(); }
#define STAMP(i) { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); ac.st_acc[i] += (unsigned)(st_now - ac.st_last); ac.st_last = st_now; }
#define STAMP_EXTRA 128
#else
#define STAMP_MEMBERS
#define STAMP_INIT(ac)
#define STAMP(i)
#define STAMP_EXTRA 0
#endif

// ---- build switches of this header (A/B measurements; every default is the product) ----
#ifndef PINN_F2_REC_LDS
#define PINN_F2_REC_LDS 1               // keep the records of the stored hidden layers in LDS as far as they fit (Spec2::NRQ)
#endif
// Explicit software pipeline of the split-operand GEMMs: the B-operand (dW: both operands') LDS reads of group i + 1 are issued in front of
// a scheduling fence, the six MFMAs of group i behind it — the compiler cannot sink the reads next to their uses (what it does when left
// alone, and what sched_group_barrier requests did not change at 256 registers), so a wave's MFMAs no longer wait one LDS round trip per
// group.  Bit mask: 1 forward GEMM, 2 dA GEMM, 4 dW GEMM.
#ifndef PINN_F2_SWP
#define PINN_F2_SWP 7
#endif
#ifndef PINN_PROBE
#define PINN_PROBE 0                    // timing probes (tools only, wrong numbers): 1 half the MFMAs, 2 no residual pieces, 4 no workgroup barriers, 8 no weight loads, 16 a third of the operand reads
#endif
#ifndef PINN_F2_ADJ_IL
#define PINN_F2_ADJ_IL 1                // transpose-read schedules: activation adjoint issued between the MFMA groups of the dW GEMM (1: H = 64 only, 2: H = 128 too)
#endif
#ifndef PINN_F2_TR_FWDIMG
#define PINN_F2_TR_FWDIMG 1             // transpose-read kernels: the forward pass's last exchange image serves as the first dW's a-jet operand
#endif
// Split products: the five SMALL piece products (hm mh hl mm lh: 2^-8 ... 2^-16 of the result) in an accumulator of their own, added to the big one
// (bias / running sum + the hh products) by one VALU add per GEMM.  v_mfma_f32_16x16x32_bf16 does not ROUND products that are small against
// its accumulator, it truncates them towards -infinity below ~2^-31 of the accumulator: six pieces on one accumulator shift every GEMM output
// by -0.6 ... -1.0e-8 (|z| ~ 1.6) — a coherent offset of every pre-activation, which a TRAINED theta amplifies (tools/micro/split_bias_probe.hip,
// profiles/r05_split_bias_probe.txt: own accumulator = zero mean AND a third of the rms error).  Bit mask: 1 forward GEMM, 2 dA GEMM,
// 4 dW GEMM with register-resident sums (H = 64), 8 dW GEMM with slab-resident sums (H = 128: not enabled — 32 more live registers in a kernel
// that already spills).  (A variant that also formed the hh products of a dW tile in fresh accumulators, input tile as the outer loop, gave
// the same parity figures to three digits and was not bit-reproducible in stand-alone launches on the hardware: removed, profiles/r05_experiments.txt.)
// r06: FORWARD ONLY (1).  Measured on MI355X at the trained fixtures (tools/r05/theta_ab_gpu.py, profiles/r06_experiments.txt section 1): masks 1, 3, 5 and 7
// give the same errors to three digits (cfg2 adam2000 / adam6000 / cfg3 adam2000 gradient against the oracle at float32(theta): 1.34e-5 / 4.06e-3 /
// 4.83e-6; mask 0: 1.75e-5 / 5.27e-3 / 5.66e-6) — the coherent offset matters where it feeds the activations' cancellation, i.e. in the
// pre-activations; the reverse GEMMs' outputs are summed over thousands of points with mixed signs.  10 fewer spilled registers, -1.5 %.
#ifndef PINN_F2_SPLIT_ACC2
#define PINN_F2_SPLIT_ACC2 1
#endif
#ifndef PINN_F2_WACC_PRELOAD
#define PINN_F2_WACC_PRELOAD 2          // slab-resident dW sums (H = 128) loaded as the dW GEMM's initial accumulators (2: fp32-MFMA kernels too)
#endif
// Which GEMM arithmetic a translation unit instantiates for its family-2 kernels (bit mask: 1 = split-operand bf16 products, 2 = fp32
// MFMAs).  The library carries both (run-time choice per handle: pinn_set_option(h, "gemm", ...)); run-time specialised units compile
// only the one their handle asked for.
#ifndef PINN_F2_MODES
#define PINN_F2_MODES 3
#endif

namespace pk {

// ---- GEMM arithmetic of the hidden-layer products of a family-2 kernel (template parameter GEMM_ of Spec2; SpecInfo::gemm) ----
//   GEMM_SPLIT (default): SPLIT-OPERAND GEMMs on the bf16 matrix pipe (64- and 128-wide kernels): an fp32 product from three bf16 pieces
//     per operand,  a b ~ ah bh + ah bm + am bh + ah bl + am bm + al bh   (a = ah + am + al to 24 bits; the dropped terms are below
//     2^-24 a b), accumulated in fp32 by v_mfma_f32_16x16x32_bf16: 12 MFMAs of ~17 cycles for a 16x16 output tile over K = 64 instead
//     of 16 fp32 MFMAs of 32 cycles (2.56x measured, tools/micro/mfma_split_bench.hip).  Forward and dA = W^T dZ: weights split once per
//     evaluation by k_pack_bf16, activations / dZ split by the publishing wave (3 x 8-byte LDS stores instead of one 16-byte store).
//     dW = dZ A^T WITHOUT any transposed staging (Spec2::BFX_TR): the exchange images are "planes" — a fragment's two neuron-tile halves
//     in separate 512-byte planes of 8-byte slots, slot(g, c) = 16 g + (c ^ 4 (g >> 1)) — so that gfx950's LDS transpose read
//     (ds_read_b64_tr_b16: a 16-lane group turns a [4 points][16 neurons] block into per-neuron columns) delivers both operands in MFMA
//     operand order (K = 32 points = two column groups) STRAIGHT OUT OF the B-operand images: dZ is already there for the dA GEMM, the
//     a-jets are published like a forward activation.  The plane layout with the XOR on the point index is conflict-free for the transpose
//     reads, the plain 8-byte reads of the forward / dA B operands and the 8-byte piece stores (tools/micro/tr_probe.hip).  Shapes
//     outside BFX_TR (64-wide nets deeper than 7 hidden layers) keep dW on fp32 MFMAs with staged operands.
//   GEMM_FP32: v_mfma_f32_16x16x4_f32 everywhere (rounds 1-2): bit-for-bit an fmaf chain per product — the exact-fp32 path for callers
//     that need the last bit (quasi-Newton finishing); ~1.35x slower on the bench workload.
// Error budget of the two modes against the float64 oracle: DESIGN.md section 6.
constexpr int GEMM_FP32 = 0, GEMM_SPLIT = 1;
// settled design parameters (measured in rounds 1-3)
constexpr int F2_WAVES128 = 8;          // waves per workgroup of the H = 128 kernels: two waves per SIMD share one workgroup's LDS tiles
constexpr int F2_NW8_MAXNG = 6;         // ... for up to this many column groups (beyond: 4 waves with 512 registers)
constexpr int F2_GEMM_AHEAD = 2;       // fp32-MFMA GEMMs: B-fragment LDS reads issued this many MFMA groups ahead of their use
constexpr int F2_SPRE_MAX = 24;         // records parked in the scratch slab are requested one phase early when they take <= this many registers


// What a wave's persistent gradient accumulators depend on: the network shape and the neuron split — NOT the jet-channel set or the
// points per tile.  Kernels of one network with different channel sets (the interior and the boundary terms of one PINN) therefore
// share the accumulators, the gradient-slab layout and the packed weight image, which is what lets ONE launch walk both tile lists
// (wave_main2m below).
template <class A, class B> struct same_type { static constexpr bool value = false; };
template <class A> struct same_type<A, A> { static constexpr bool value = true; };
template <int HP_, int NHH_, int D_, int NW_, bool WBAR_REG_>
struct Shape2 {
    static constexpr int HP = HP_, MT = HP_ / 16, NHH = NHH_, LH = NHH_ + 1, D = D_, NW = NW_, MTW = MT / NW_;
    static constexpr bool WBAR_REG = WBAR_REG_;
};
template <class SH>
struct Acc2 {
    vfloat4 wbar[(SH::WBAR_REG && SH::NHH > 0) ? SH::NHH : 1][SH::MTW][SH::MT];
    vfloat4 bbar[SH::LH][SH::MTW];
    vfloat4 w1bar[SH::D][SH::MTW];
    vfloat4 wLbar[SH::MTW];
    vfloat bLbar;
    vfloat pbar[MAX_PARAMS];
    vdacc lsum;                  // running sum of squares of the term being processed (tape waves), double per lane
    int cur_term;                // index into GroupArgs::terms of that term, -1: none yet
    STAMP_MEMBERS
};

template <int HP_, int NHH_, int D_, unsigned D1MASK_, unsigned long long PAIRS_, int NPAIR_, int PG_, unsigned HI_ = 0, int GEMM_ = GEMM_SPLIT>
struct Spec2 {
    using J = JetSet<D1MASK_, PAIRS_, NPAIR_, HI_>;
    static constexpr unsigned HI = HI_;
    static constexpr int FAMILY = 2;
    static constexpr int GEMM = GEMM_;
    static constexpr int HP = HP_, MT = HP_ / 16, NHH = NHH_, LH = NHH_ + 1, D = D_, NPAIR = NPAIR_, PG = PG_;
    static constexpr unsigned D1MASK = D1MASK_;
    static constexpr unsigned long long PAIRS = PAIRS_;
    // waves per workgroup: 8 where the workgroup's LDS tiles (3 x NG x MT KB) leave room for only one workgroup per CU anyway and
    // a wave's state fits 256 registers tolerably (measured: NG <= 4, cfg4 44.5 -> 39.0 ms; NG = 6, the forward-Laplacian kernel of
    // cfg5: 114 spilled registers, yet 38.8 -> 34.6 ms against the 4-wave, 512-register build — the second wave per SIMD wins)
    static constexpr int NW = (HP_ >= 128 && J::C * PG_ <= F2_NW8_MAXNG) ? F2_WAVES128 : 4;
    static constexpr int MTW = MT / NW;                // neuron tiles per wave
    static_assert(MT % NW == 0 && MT % 4 == 0, "family 2 needs a hidden width that is a multiple of 64 (16 x waves per workgroup)");
    static constexpr int NFIRST = J::NFIRST;
    static constexpr int C = J::C;
    static constexpr int NG = C * PG_;
    static constexpr int TP = 16 * PG_;
    static constexpr int first_axis(int k) { return J::first_axis(k); }
    static constexpr int first_rank(int axis) { return J::first_rank(axis); }
    static constexpr int pair_a(int p) { return J::pair_a(p); }
    static constexpr int pair_b(int p) { return J::pair_b(p); }
    // packed parameter buffer (floats); fragment images are [layer][tile a][tile b][lane][4 k-steps]
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_B = OFF_W1 + D_ * HP_;
    static constexpr int off_b(int layer) { return OFF_B + layer * HP_; }
    static constexpr int OFF_WL = OFF_B + LH * HP_;
    static constexpr int OFF_BL = OFF_WL + HP_;
    static constexpr int OFF_WPK = OFF_BL + 4;                       // [NHH][mo][mi][64][4]: W[16mo+(l&15)][16mi+4(l>>4)+rr]
    static constexpr int OFF_WTPK = OFF_WPK + NHH_ * HP_ * HP_;      // [NHH][mi][mo][64][4]: W[16mo+4(l>>4)+rr][16mi+(l&15)]
    static constexpr int PACKED0 = OFF_WTPK + NHH_ * HP_ * HP_;
    // split-operand images (k_pack_bf16): [NHH][out tile][k-block][piece][64 lanes][8 bf16] = 256 floats per fragment, forward then transposed
    static constexpr int BF_LAYER = (HP_ / 16) * (HP_ / 32) * 3 * 256;
    static constexpr int OFF_WB = PACKED0;
    static constexpr int OFF_WTB = OFF_WB + NHH_ * BF_LAYER;
    static constexpr bool BFIMG = (GEMM_ == GEMM_SPLIT) && (HP_ == 64 || HP_ == 128);   // the net's weight image carries the bf16 pieces
    static constexpr bool HAS_SPLIT = (HP_ == 64 || HP_ == 128);     // the shape exists in both GEMM modes (SpecInfo::twin)
    static constexpr int PACKED = BFIMG ? OFF_WTB + NHH_ * BF_LAYER : PACKED0;
    // per-workgroup gradient slab: every entry is written by exactly one wave
    static constexpr int O_WBAR = 0;                                 // [NHH][to][ti][64][4]
    static constexpr int O_BH = NHH_ * HP_ * HP_;                    // [LH][HP]  natural neuron order
    static constexpr int O_W1 = O_BH + LH * HP_;                     // [D][HP]
    static constexpr int O_WL = O_W1 + D_ * HP_;                     // [HP]
    static constexpr int O_BL = O_WL + HP_;
    static constexpr int O_P = O_BL + 1;
    static constexpr int SLAB = ((O_P + MAX_PARAMS + 63) / 64) * 64 + STAMP_EXTRA;
    // per-workgroup record scratch, [layer][q][tile][lane][4]; layer LH-1 stays in registers, layer 0 is recomputed
    static constexpr int SCR = (LH > 2 ? LH - 2 : 1) * NG * MT * 256;      // hidden layers 1 .. LH-2
    // per-TILE record store of the two-launch path for coupled equations (MODE_FWDREC -> MODE_GRADREC): hidden layers 1 .. LH-1
    static constexpr int REC = (LH - 1) * NG * MT * 256;
    // LDS (floats): X0 | X1 (activation / dZ exchange, A^T) | ZT (4 x private dZ^T) | output partials | coords
    static constexpr int XSZ = NG * MT * 256;
    // split-operand GEMMs: the exchange buffers hold B operands as three bf16 pieces, [q][k-block of 32][piece][lane][8 bf16]
    static constexpr int KB = MT / 2;                                // k-blocks of 32 per layer
    static constexpr int XSZ_BF = NG * KB * 3 * 256;
    static constexpr int UP_SZ = (((NW + 1) * NG * 16 + 63) / 64) * 64;
    // TR_SHAPE: no dW staging at all — dW reads its operands out of the exchange images with the LDS transpose read (H = 64: weight
    // fragments prefetched, dW sums in registers; H = 128: 8-wave workgroups, dW sums in the slab)
    static constexpr bool TR_SHAPE = (HP_ == 64 && NW == 4 && (NHH_ * (MT / NW) * MT * 4 <= 96)) || (HP_ == 128 && NW == 8);
    // the bigger exchange buffers must leave the un-chunked fp32 dW staging in place where the shape still stages (BFX without BFX_TR)
    static constexpr bool BFX = BFIMG && ((2 * XSZ_BF + (TR_SHAPE ? 0 : NG * MT * 256) + UP_SZ) * 4 <= 160 * 1024);
    static constexpr int XSZB = BFX ? XSZ_BF : XSZ;                  // floats of one exchange buffer
    // dW operands by LDS transpose reads out of the plane-layout exchange images; a shape-level decision — every member of a merged launch
    // stores its dW tiles in the same (natural) order.  An odd number of column groups pads the last K = 32 block with zeros.
    static constexpr bool BFX_TR = BFX && TR_SHAPE;
    static constexpr bool DW_NATURAL = BFX_TR;                       // dW tiles in natural order: column c of tile ti = input neuron 16 ti + c
    static constexpr int ZTW = BFX_TR ? 0 : NG * (MT / NW) * 256;    // floats of one wave's private dZ^T staging
    static constexpr int LDS_UP = (((NW + 1) * NG * 16 + 63) / 64) * 64;    // NW x output partials + seed broadcast (UB)
    static_assert(PG_ >= 1 && PG_ <= 4, "one tape wave per point group");
    // when X0 | X1 | ZT would not fit in 160 KiB (H = 128 with 8 jet channels) the dW operands are staged one column group
    // at a time in a double buffer carved out of X1: [A^T chunk 16 x HP | 4 x dZ^T chunk]
    static constexpr bool CHUNKED = (2 * XSZB + NW * ZTW + LDS_UP) * 4 > 160 * 1024;
    static constexpr int CH_AT = 16 * HP_;
    static constexpr int CH_ZT = MTW * 256;
    static constexpr int CHSZ = CH_AT + NW * CH_ZT;
    static_assert(!CHUNKED || 2 * CHSZ <= XSZ, "chunk double buffer must fit inside X1");
    static constexpr int OFF_X1 = XSZB;                                                                          // LDS offsets (floats)
    static constexpr int OFF_ZT = 2 * XSZB;
    static constexpr int OFF_UP = 2 * XSZB + (CHUNKED ? 0 : NW * ZTW);
    static constexpr int LDS_BASE = OFF_UP + LDS_UP;
    // (r05 experiment, -DPINN_F2_WG3=1 with -DPINN_F2_REC_LDS=0: THREE workgroups per CU — 168 registers per wave, <= 53 KB LDS per workgroup;
    // profiles/r05_experiments.txt)
#ifndef PINN_F2_WG3
#define PINN_F2_WG3 0
#endif
    static constexpr int WG_PER_CU = (NW == 8) ? 1 : ((PINN_F2_WG3 && LDS_BASE * 4 <= 53 * 1024) ? 3 : ((LDS_BASE * 4 <= 80 * 1024) ? 2 : 1));
    // the records of the stored hidden layers stay in LDS instead of the per-workgroup scratch slab in global memory — as many
    // (layer, column group) slices as fit without lowering the number of resident workgroups, from layer LH-2 (needed first by the
    // reverse sweep) downwards.  Each wave writes and reads its own part of a slice only.  4x64, NG = 4: 7 of the 8 slices.
    static constexpr int RECQ = MT * 256;                            // one column group of one layer's record of a tile (floats)
    static constexpr int LDS_CAP = (WG_PER_CU == 3 ? 53 : (WG_PER_CU == 2 ? 80 : 160)) * 256 - 64;    // floats per workgroup
    static constexpr int NRQ_ALL = (LH > 2 ? LH - 2 : 0) * NG;
    static constexpr int NRQ_FIT = (LDS_CAP - LDS_BASE) / RECQ;
    static constexpr int NRQ = PINN_F2_REC_LDS ? (NRQ_FIT < NRQ_ALL ? (NRQ_FIT > 0 ? NRQ_FIT : 0) : NRQ_ALL) : 0;
    static constexpr int LDS_WG = LDS_BASE + NRQ * RECQ;
    static constexpr int OCC = WG_PER_CU * NW / 4;                       // waves per SIMD the kernel is compiled for
    // FORWARD-ONLY launches (MODE_FWD / FWDREC / RESID / LOSS: no reverse sweep) use the two exchange buffers and the output partials
    // only and a fraction of the registers: they are compiled for up to four waves per SIMD (the loss-only evaluation, pinn_phi,
    // pinn_residual, the forward launches of coupled equations)
    static constexpr int LDS_FWD = 2 * XSZB + LDS_UP;
    static constexpr int WG_FWD_LDS = (160 * 1024 - 1024) / (LDS_FWD * 4);
    static constexpr int WG_FWD = (WG_FWD_LDS * NW >= 16) ? 16 / NW : (WG_FWD_LDS < 1 ? 1 : WG_FWD_LDS);
    static constexpr int OCC_FWD = WG_FWD * NW / 4;
    // dW accumulators: resident in registers across tiles when they fit (4x64: 48 registers); for wide/deep nets they
    // are accumulated per tile into this workgroup's slab instead (read-modify-write, L2; same wave owns the same tiles)
#ifndef PINN_F2_NO_WBAR_REG
#define PINN_F2_NO_WBAR_REG 0           // (r05 experiment: the dW sums of the H = 64 kernels in the slab as well — 48 registers fewer per wave)
#endif
    static constexpr bool WBAR_REG = !PINN_F2_NO_WBAR_REG && (NHH_ * MTW * MT * 4 <= 96);
    using Shape = Shape2<HP_, NHH_, D_, NW, WBAR_REG>;
};

// ---- persistent gradient accumulators of this wave's neuron tiles: zero, or (chained launch group) the sums an earlier launch group of
// this network stored in this workgroup's slab ----
template <class S>
DEV void acc2_init(Acc2<typename S::Shape>& ac, const GroupArgs& ga, int blk, int w, bool bwd) {
    constexpr int HP = S::HP, MT = S::MT, MTW = S::MTW, NHH = S::NHH, LH = S::LH, D = S::D;
    const vint lane = lane_id();
    const vint g = lane >> 4;
    const vint c = lane & vint(15);
    float* slab = ga.slabs + (size_t)blk * S::SLAB;
    PINN_UNROLL for (int l = 0; l < ((S::WBAR_REG && NHH > 0) ? NHH : 1); ++l)
        PINN_UNROLL for (int t = 0; t < MTW; ++t)
            PINN_UNROLL for (int ti = 0; ti < MT; ++ti) ac.wbar[l][t][ti] = vzero4();
    if (!S::WBAR_REG && bwd && !ga.chain)    // slab-resident dW: this wave's tiles start at zero (or on top of the chained group's sums)
        for (int hl = 0; hl < NHH; ++hl)
            PINN_UNROLL for (int t = 0; t < MTW; ++t)
                PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                    gstore4(slab + S::O_WBAR, vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2), vzero4());
    PINN_UNROLL for (int l = 0; l < LH; ++l)
        PINN_UNROLL for (int t = 0; t < MTW; ++t) ac.bbar[l][t] = vzero4();
    PINN_UNROLL for (int i = 0; i < D; ++i)
        PINN_UNROLL for (int t = 0; t < MTW; ++t) ac.w1bar[i][t] = vzero4();
    PINN_UNROLL for (int t = 0; t < MTW; ++t) ac.wLbar[t] = vzero4();
    ac.bLbar = vfloat(0.f);
    PINN_UNROLL for (int i = 0; i < MAX_PARAMS; ++i) ac.pbar[i] = vfloat(0.f);
    ac.lsum = vdacc_zero();
    ac.cur_term = -1;
    STAMP_INIT(ac)
    if (bwd && ga.chain) {
        // chained launch group: the accumulators continue from the sums an earlier launch group of this network stored in this slab
        // (same wave, same entries), so one slab set — one reduction input — carries both groups.  Lane c = 0 of a row group holds the
        // stored value, the other column lanes start at zero: the row sums of the epilogue then include it exactly once.
        if (S::WBAR_REG)
            PINN_UNROLL for (int hl = 0; hl < NHH; ++hl)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                        ac.wbar[hl][t][ti] = gload4(slab + S::O_WBAR, vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2));
        const vbool c0i = veq(c, 0);
        PINN_UNROLL for (int t = 0; t < MTW; ++t)
            PINN_UNROLL for (int r = 0; r < 4; ++r) {
                const vint n = vint(16 * (w * MTW + t) + r) + (g << 2);
                PINN_UNROLL for (int l = 0; l < LH; ++l) ac.bbar[l][t][r] = gload_masked(slab + S::O_BH + l * HP, n, c0i);
                PINN_UNROLL for (int i = 0; i < D; ++i) ac.w1bar[i][t][r] = gload_masked(slab + S::O_W1 + i * HP, n, c0i);
                ac.wLbar[t][r] = gload_masked(slab + S::O_WL, n, c0i);
            }
        if (w == 0) {
            const vbool l0 = veq(lane, 0);
            ac.bLbar = gload_masked(slab + S::O_BL, lane & vint(0), l0);
            PINN_UNROLL for (int j = 0; j < MAX_PARAMS; ++j) ac.pbar[j] = gload_masked(slab + S::O_P, vint(j) + (lane & vint(0)), l0);
        }
    }
}

// ---- the tile loop: workgroup `blk` of `nblocks` takes the tiles tix = blk (mod nblocks) of [tile_lo, tile_hi), which belong to the terms
// [term_lo, term_hi) of the launch (one launch group: everything; a merged launch: one call per kernel family member) ----
template <class S, int MODE, int ACTK>
DEV void wave_tiles2(const GroupArgs& ga, int blk, int nblocks, int w, float* lds, Acc2<typename S::Shape>& ac,
                     int term_lo, int term_hi, int tile_lo, int tile_hi) {
    constexpr int HP = S::HP, MT = S::MT, MTW = S::MTW, NHH = S::NHH, LH = S::LH, D = S::D, C = S::C, PG = S::PG, NG = S::NG;
    constexpr int NFIRST = S::NFIRST;
    using J = typename S::J;
    constexpr bool BWD = (MODE == MODE_FUSED || MODE == MODE_GRADIN || MODE == MODE_GRADREC);
    constexpr bool SUMS = (MODE == MODE_FUSED || MODE == MODE_LOSS);       // modes that deliver the per-term sums of squares
    constexpr bool TAPE_ONLY = (MODE == MODE_RESID || MODE == MODE_LOSS);  // forward + tape, no reverse sweep
    constexpr bool RECOUT = (MODE == MODE_FWDREC), RECIN = (MODE == MODE_GRADREC);
    constexpr bool IS_FWD = (MODE == MODE_FWD || MODE == MODE_FWDREC), IS_GRADIN = (MODE == MODE_GRADIN || MODE == MODE_GRADREC);
    constexpr bool WPRE = (MT * MTW * 4 <= 16);        // prefetch a layer's weight fragments when they take <= 16 registers (H = 64)
    // level 3 in the H = 64 schedule (dA and dW behind ONE barrier): the exchange buffers stay in use until the end of the dW GEMM, so the
    // barrier that frees them sits in front of the next publish (the H = 128 schedule keeps its barrier behind the GEMMs)
    constexpr bool TR_OVL = S::BFX_TR && WPRE && !S::CHUNKED;
    const int wave = blk * S::NW + w;
    const vint lane = lane_id();
    const vint g = lane >> 4;
    const vint c = lane & vint(15);
    const vbool g0 = veq(g, 0);
    // this lane's 8-byte slot in a plane of a split-operand exchange image (S::BFX_TR; floats), else its 16-byte slot
    const vint sw = S::BFX_TR ? (((g << 4) + (c ^ ((g >> 1) << 2))) << 1) : (lane << 2);
    // transpose-read addresses of the dW operands (S::BFX_TR): lane (g, c) of a 16-lane group reads the slot of row group c & 3 at point
    // 8 (g & 1) + 4 jh + (c >> 2) of column group 2 qp + (g >> 1) and receives neuron c of the tile at the points k = 8 g + 4 jh + 0..3 of the
    // K = 32 block (column group 2 qp + (k >> 4), point k & 15)
    vint trb[2];
    PINN_UNROLL for (int jh = 0; jh < 2; ++jh) {
        const vint gg = c & vint(3), pc = ((g & vint(1)) << 3) + vint(4 * jh) + (c >> 2);
        trb[jh] = (((gg << 4) + (pc ^ ((gg >> 1) << 2))) << 1);
    }
    vint trbq[2];                                            // ... + the offset of the pair's second column group for k >= 16
    PINN_UNROLL for (int jh = 0; jh < 2; ++jh) trbq[jh] = trb[jh] + (g >> 1) * vint(S::KB * 3 * 256);
    const vbool klo = vlt(g, 2);                             // lanes whose k < 16 (first column group of a pair)
    const float* P = ga.packed;
    // the activation kind is a template parameter: every kernel is straight-line code behind its GEMMs (no activation branches for the
    // optimiser to hoist); sin variants are compiled only for the specs registered with PINN_INSTANTIATE*_SIN
    constexpr int act = ACTK;
    static_assert(ACTK != ACT_MIXED, "per-layer activation kinds are compiled for family 1 (small nets) only");
    constexpr bool SINACT = (ACTK == ACT_SIN);
    const ubuf PB = ub_make(P, S::PACKED);
    const ubuf SB = ub_make(ga.scratch + (size_t)blk * (ga.scr_stride ? ga.scr_stride : S::SCR), S::SCR);
    float* X0 = lds;
    float* X1 = lds + S::OFF_X1;
    float* ZT = lds + S::OFF_ZT + w * S::ZTW;                 // wave-private dZ^T: [q][t][16 columns][16 neurons] 
    float* UP = lds + (BWD ? S::OFF_UP : 2 * S::XSZB);        // output-layer partial sums [wave][q][16] (forward-only launches: LDS_FWD)
    float* RL = lds + S::LDS_BASE;                            // LDS-resident record slices [(LH-2-layer)*NG + q][tile][lane][4]
    auto rec_in_lds = [](int layer, int q) { return layer >= 1 && layer <= LH - 2 && (LH - 2 - layer) * NG + q < S::NRQ; };   // (same-kernel records only)
    auto rl_off = [&](int layer, int q, int t) { return vint((((LH - 2 - layer) * NG + q) * MT + w * MTW + t) * 256) + (lane << 2); };

    float* slab = ga.slabs + (size_t)blk * S::SLAB;
    auto& wbar = ac.wbar;
    auto& bbar = ac.bbar;
    auto& w1bar = ac.w1bar;
    auto& wLbar = ac.wLbar;
    vfloat& bLbar = ac.bLbar;
    auto& pbar = ac.pbar;
    vdacc& lsum = ac.lsum;
    int& cur_term = ac.cur_term;

    // weight fragments of this wave's neuron tile t: forward A operand of k-block mi, transposed A operand of output tile mo, bias
    auto ld_wf = [&](int hl, int t, int mi) -> vfloat4 {
        return ub_load4(PB, S::OFF_WPK + ((hl * MT + w * MTW + t) * MT + mi) * 256, lane << 2);
    };
    auto ld_wt = [&](int hl, int t, int mo) -> vfloat4 {
        return ub_load4(PB, S::OFF_WTPK + ((hl * MT + w * MTW + t) * MT + mo) * 256, lane << 2);
    };
    auto ld_bias = [&](int layer, int t) -> vfloat4 { return ub_load4(PB, S::off_b(layer) + 16 * (w * MTW + t), g << 2); };

    vfloat4 wL[MTW];
    PINN_UNROLL for (int t = 0; t < MTW; ++t) wL[t] = ub_load4(PB, S::OFF_WL + 16 * (w * MTW + t), g << 2);
    // (PINN_PROBE & 8, timing probe: one fragment, loaded once, stands for all — fp32-MFMA kernels have no bf16 image: OFF_WB is the image's end)
    const vbf8 wb_probe = ((PINN_PROBE & 8) && S::BFIMG) ? ub_load_bf8(PB, S::OFF_WB, lane << 2) : vbf8{};
    const float bL = P[S::OFF_BL];

    wave_prio(1);
    // (no dummy tiles: every wave of a workgroup works on the same tile, so a workgroup simply stops after its last one)
    const int first = tile_lo + (((blk - tile_lo) % nblocks) + nblocks) % nblocks;
    for (int tix = first; tix < tile_hi; tix += nblocks) {
        STAMP(15)
        int k = term_lo;
        for (int j = term_lo + 1; j < term_hi; ++j)
            if (tix >= ga.terms[j].tile0) k = j;
        if (k != cur_term) {
            if (cur_term >= 0 && SUMS)
                ga.losspart[(size_t)wave * ga.nterms_total + ga.terms[cur_term].term_id] = wave_sum_dd(lsum, g0);
            lsum = vdacc_zero();
            cur_term = k;
        }
        const TermDev& T = ga.terms[k];
        const int pbase = (tix - T.tile0) * S::TP;
        const ubuf RB = ub_make((RECOUT || RECIN) ? ga.rec + (size_t)tix * S::REC : ga.scratch, S::REC);    // this tile's record slot

        vfloat x[PG][D];
        vbool valid[PG];
        PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
            vint p = vint(pbase + 16 * pg) + c;
            valid[pg] = vlt(p, T.N);
            PINN_UNROLL for (int i = 0; i < D; ++i) x[pg][i] = gload_masked(T.pts, p * T.dt + vint(T.imap[i]), valid[pg]);
        }

        // jet activation in place + record: layer LH-1 stays in registers (Rlast), the others go to the scratch slab
        vfloat4 Rlast[NG][MTW];
        auto act_forward = [&](vfloat4 (&Z)[NG][MTW], int layer) {
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                    vfloat4 av;                                       // sin: activation values; Z[pg*C] holds the RECORD value z meanwhile
                    av = act_value4<SINACT>(act, Z[pg * C][t]);
                    if (!SINACT) Z[pg * C][t] = av;
                    if (RECOUT && layer > 0)
                        PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                            ub_store4(RB, (((layer - 1) * NG + pg * C + ch) * MT + w * MTW + t) * 256, lane << 2, Z[pg * C + ch][t]);
                    if (BWD) {
                        if (layer == LH - 1) {
                            PINN_UNROLL for (int ch = 0; ch < C; ++ch) Rlast[pg * C + ch][t] = Z[pg * C + ch][t];
                        } else if (layer > 0) {          // layer 0's record is recomputed from the coordinates (no storage)
                            PINN_UNROLL for (int ch = 0; ch < C; ++ch) {
                                if (rec_in_lds(layer, pg * C + ch)) lds_store4(RL, rl_off(layer, pg * C + ch, t), Z[pg * C + ch][t]);
                                else ub_store4(SB, (((layer - 1) * NG + pg * C + ch) * MT + w * MTW + t) * 256, lane << 2, Z[pg * C + ch][t]);
                            }
                        }
                    }
                    if ((RECOUT && layer > 0) || (BWD && layer > 0 && layer != LH - 1 && !rec_in_lds(layer, pg * C + C - 1))) {    // the stored channels are updated in place below
                        store_pad();
                        PINN_UNROLL for (int ch = 0; ch < C; ++ch) keep_alive(Z[pg * C + ch][t]);
                        sched_fence();
                    }
                    PINN_UNROLL for (int r = 0; r < 4; ++r) {
                        vfloat zz[C], dd[ND];
                        PINN_UNROLL for (int ch = 0; ch < C; ++ch) zz[ch] = Z[pg * C + ch][t][r];
                        act_derivs_n<J::NORD - 1, SINACT>(act, zz[0], dd);
                        jet_forward<J>(zz, dd);
                        PINN_UNROLL for (int ch = 1; ch < C; ++ch) Z[pg * C + ch][t][r] = zz[ch];
                    }
                    if (SINACT) Z[pg * C][t] = av;
                }
        };
        // publish this wave's tiles of a [NG][MT] tensor in B-fragment order: X[q][tile][lane][4]
        // split-operand GEMMs (S::BFX): three bf16 pieces per value in the operand order of the 16x16x32 MFMA, X[q][k-block][piece][lane][8]:
        // k = 8 g + j of a k-block <-> neuron tile 2 kb + (j >> 2), row 4 g + (j & 3) — a lane's own four rows fill its half of the slot
        auto publish = [&](float* X, const vfloat4 (&A)[NG][MTW]) {
            if (S::BFX) {
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                        const int tile = w * MTW + t;
                        vbf4 ph, pm, pl;
#if PINN_PROBE & 2
                        split1_bf16(A[q][t], ph); pm = ph; pl = ph;      // timing probe: no residual pieces (wrong numbers)
#else
                        split3_bf16(A[q][t], ph, pm, pl);
#endif
                        const vint at = vint(((q * S::KB + (tile >> 1)) * 3) * 256 + (tile & 1) * (S::BFX_TR ? 128 : 2)) + sw;
                        lds_store_bf4(X, at, ph);
                        lds_store_bf4(X, at + vint(256), pm);
                        lds_store_bf4(X, at + vint(512), pl);
                    }
                return;
            }
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    lds_store4(X, vint(((q * MT + w * MTW + t) * 64) * 4) + (lane << 2), A[q][t]);
        };
        // B operand (this lane's point, k = 8 g + j) of fragment `frag` = (column group, k-block, piece) of an exchange image
        auto ld_bfrag = [&](const float* X, int frag) -> vbf8 {
#if PINN_PROBE & 16
            frag = (frag / 3) * 3;                      // timing probe: one LDS read per group serves all three pieces (wrong numbers)
#endif
            if (S::BFX_TR) return cat_bf8(lds_load_bf4(X, vint(frag * 256) + sw), lds_load_bf4(X, vint(frag * 256 + 128) + sw));
            return lds_load_bf8(X, vint(frag * 256) + (lane << 2));
        };
        // six bf16 MFMAs = one fp32-accurate 16x16 (x) 16x32 product (GEMM_SPLIT)
        auto mfma_split = [&](const vbf8 (&a)[3], const vbf8 (&b)[3], vfloat4 c) -> vfloat4 {
            c = mfma16x32bf(a[0], b[0], c);
            c = mfma16x32bf(a[0], b[1], c);
            c = mfma16x32bf(a[1], b[0], c);
#if PINN_PROBE & 1
            return c;                                   // timing probe: half the matrix work (wrong numbers)
#endif
            c = mfma16x32bf(a[0], b[2], c);
            c = mfma16x32bf(a[1], b[1], c);
            c = mfma16x32bf(a[2], b[0], c);
            return c;
        };

        // the same with the five small piece products on an accumulator of their own (PINN_F2_SPLIT_ACC2), smallest first
        auto mfma_split2 = [&](const vbf8 (&a)[3], const vbf8 (&b)[3], vfloat4& big, vfloat4& sm) {
#if !(PINN_PROBE & 1)
            sm = mfma16x32bf(a[2], b[0], sm);
            sm = mfma16x32bf(a[1], b[1], sm);
            sm = mfma16x32bf(a[0], b[2], sm);
#endif
            sm = mfma16x32bf(a[1], b[0], sm);
            sm = mfma16x32bf(a[0], b[1], sm);
            big = mfma16x32bf(a[0], b[0], big);
        };
        // C[q][t] += W[kb][t] (x) X[q][kb] over every column group and k-block, software-pipelined (PINN_F2_SWP): group i + 1's three piece
        // fragments are requested before group i's MFMAs
        // acc: 0 all six piece products on Cc (r03-r04); 1 the five small ones on accumulators of their own, merged at the end (PINN_F2_SPLIT_ACC2);
        // 2 TWO PASSES over the groups on the one accumulator — every small piece product of every k-block first, the hh products last — for
        // the kernels without 4 NG MTW registers to spare (H = 128 with >= 5 column groups: cfg5 20.0 -> 21.8 ms with mode 1, gpurun r05u); Cc must
        // start at ZERO (the caller adds the bias afterwards), the hh pass re-reads the h piece of every group's B operand
        auto gemm_swp = [&](const float* X, const vbf8 (&wfr)[S::BFX ? S::KB : 1][S::BFX ? MTW : 1][3], vfloat4 (&Cc)[NG][MTW], int acc) {
            constexpr int NGRP = S::KB * NG;
            constexpr int NB = 2;                                               // operand buffers in rotation (three measured in r04: no gain)
            constexpr int RD = S::BFX_TR ? 2 : 1;                                // LDS reads per piece fragment
            if (acc == 2) {
                vbf8 bb[NB][3];
                PINN_UNROLL for (int sp = 0; sp < 3; ++sp) bb[0][sp] = ld_bfrag(X, sp);
                PINN_UNROLL for (int i = 0; i < 2 * NGRP; ++i) {
                    const int gi = i % NGRP, kb = gi / NG, q = gi % NG;
                    const bool small_pass = i < NGRP;
                    if (i + 1 < 2 * NGRP) {
                        const int g2 = (i + 1) % NGRP, kb2 = g2 / NG, q2 = g2 % NG;
                        const int np = (i + 1 < NGRP) ? 3 : 1;                   // the hh pass needs the h piece only
                        PINN_UNROLL for (int sp = 0; sp < 3; ++sp) if (sp < np) bb[(i + 1) % NB][sp] = ld_bfrag(X, (q2 * S::KB + kb2) * 3 + sp);
                    }
                    sched_fence();
                    if (i + 1 >= 2 * NGRP) lds_wait<0>(); else if (i + 1 < NGRP) lds_wait<3 * RD>(); else lds_wait<RD>();
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                        const vbf8 (&a_)[3] = wfr[kb][t];
                        const vbf8 (&b_)[3] = bb[i % NB];
                        if (small_pass) {
                            Cc[q][t] = mfma16x32bf(a_[2], b_[0], Cc[q][t]);
                            Cc[q][t] = mfma16x32bf(a_[1], b_[1], Cc[q][t]);
                            Cc[q][t] = mfma16x32bf(a_[0], b_[2], Cc[q][t]);
                            Cc[q][t] = mfma16x32bf(a_[1], b_[0], Cc[q][t]);
                            Cc[q][t] = mfma16x32bf(a_[0], b_[1], Cc[q][t]);
                        } else {
                            Cc[q][t] = mfma16x32bf(a_[0], b_[0], Cc[q][t]);
                        }
                    }
                    chain_fence();
                }
                return;
            }
            vfloat4 Cs[NG][MTW];
            if (acc == 1)
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) Cs[q][t] = vzero4();
            vbf8 bb[NB][3];
            PINN_UNROLL for (int sp = 0; sp < 3; ++sp) bb[0][sp] = ld_bfrag(X, sp);          // group 0: kb = 0, q = 0
            PINN_UNROLL for (int i = 0; i < NGRP; ++i) {
                const int kb = i / NG, q = i % NG;
                if (i + 1 < NGRP) {
                    const int kb2 = (i + 1) / NG, q2 = (i + 1) % NG;
                    PINN_UNROLL for (int sp = 0; sp < 3; ++sp) bb[(i + 1) % NB][sp] = ld_bfrag(X, (q2 * S::KB + kb2) * 3 + sp);
                }
                sched_fence();
                // every piece of this group's B operand has arrived before the first MFMA of the chain (see lds_wait): only the next group's
                // reads may still be in flight
                if (i + 1 < NGRP) lds_wait<3 * RD>(); else lds_wait<0>();
                PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                    if (acc == 1) mfma_split2(wfr[kb][t], bb[i % NB], Cc[q][t], Cs[q][t]);
                    else Cc[q][t] = mfma_split(wfr[kb][t], bb[i % NB], Cc[q][t]);
                }
                chain_fence();                                              // (nothing of the next group moves up into this chain)
            }
            if (acc == 1)
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int e = 0; e < 4; ++e) Cc[q][t][e] += Cs[q][t][e];
        };
        // which form a GEMM site takes (bit of PINN_F2_SPLIT_ACC2: 1 forward, 2 dA)
#ifndef PINN_F2_ACC_2PASS_ALL
#define PINN_F2_ACC_2PASS_ALL 0         // (A/B: the two-pass form for every split-GEMM kernel)
#endif
        constexpr bool ACC_2PASS = PINN_F2_ACC_2PASS_ALL || (HP >= 128 && NG >= 5);
        auto acc_mode = [&](int bit) -> int { return (PINN_F2_SPLIT_ACC2 & bit) ? (ACC_2PASS ? 2 : 1) : 0; };

        // =========================== forward ===========================
        vfloat4 A[NG][MTW];
        vfloat U[PG][C];
        constexpr int SRC_PRE = 2;                     // source channels requested one phase ahead of the tape (the rest at tape time)
        vfloat srcv[SRC_PRE];
        PINN_UNROLL for (int j = 0; j < SRC_PRE; ++j) srcv[j] = vfloat(0.f);
        vbool valid_w = valid[0];
        PINN_UNROLL for (int pg = 1; pg < PG; ++pg) if (w == pg) valid_w = valid[pg];
        if (!RECIN) {
            PINN_UNROLL for (int t = 0; t < MTW; ++t) {                          // hidden layer 0: d -> HP on the VALU
                const int n0 = 16 * (w * MTW + t);
                vfloat4 b1 = ub_load4(PB, S::off_b(0) + n0, g << 2);
                vfloat4 w1[D];
                PINN_UNROLL for (int i = 0; i < D; ++i) w1[i] = ub_load4(PB, S::OFF_W1 + i * HP + n0, g << 2);
                PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                    vfloat4 z = b1;
                    PINN_UNROLL for (int i = 0; i < D; ++i)
                        PINN_UNROLL for (int r = 0; r < 4; ++r) z[r] = vfma(w1[i][r], x[pg][i], z[r]);
                    A[pg * C][t] = z;
                    PINN_UNROLL for (int kf = 0; kf < NFIRST; ++kf) A[pg * C + 1 + kf][t] = w1[S::first_axis(kf)];
                    PINN_UNROLL for (int ch = 1 + NFIRST; ch < C; ++ch) A[pg * C + ch][t] = vzero4();      // second and higher derivatives of an affine map
                }
            }
            act_forward(A, 0);
            STAMP(0)
            PINN_UNROLL for (int hl = 0; hl < NHH; ++hl) {
                float* Xin = (hl & 1) ? X1 : X0;
                // this wave's weight fragments + bias of the layer: issued before the exchange so that their L2 latency hides
                // under publish + barrier instead of stalling the first MFMA of every k-block
                vfloat4 wf[WPRE ? MT : 1][MTW], bv[MTW];
                vbf8 wb[S::BFX ? S::KB : 1][S::BFX ? MTW : 1][3];      // (H = 128: 48 registers; fetched per k-block their L2 latency showed at every k-block)
                if (S::BFX) {
                    // loads return in order: the bias (the accumulators' initial value) first, then the fragments in the order the GEMM uses them,
                    // so the first MFMA waits for the head of the stream, not for all of it
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) bv[t] = ld_bias(hl + 1, t);
                    PINN_UNROLL for (int kb = 0; kb < S::KB; ++kb)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            PINN_UNROLL for (int sp = 0; sp < 3; ++sp)
                                wb[kb][t][sp] = (PINN_PROBE & 8) ? wb_probe : ub_load_bf8(PB, S::OFF_WB + (((hl * MT + w * MTW + t) * S::KB + kb) * 3 + sp) * 256, lane << 2);
                    sched_fence();
                } else if (WPRE) {
                    PINN_UNROLL for (int mi = 0; mi < MT; ++mi)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            wf[mi][t] = ld_wf(hl, t, mi);
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) bv[t] = ld_bias(hl + 1, t);
                    sched_fence();
                }
                if (TR_OVL && BWD && hl == 0) wg_barrier();                     // the previous tile's last dW GEMM reads X0 / X1 (transpose reads)
                publish(Xin, A);
                wg_barrier();                                                   // layer hl activations complete in Xin
                STAMP(1)
                const bool bias_after = S::BFX && (PINN_F2_SWP & 1) && acc_mode(1) == 2;      // two-pass GEMM: accumulators start at zero
                PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                    if (!WPRE && !S::BFX) bv[t] = ld_bias(hl + 1, t);
                    PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                        A[pg * C][t] = bias_after ? vzero4() : bv[t];
                        PINN_UNROLL for (int ch = 1; ch < C; ++ch) A[pg * C + ch][t] = vzero4();
                    }
                }
                wave_prio(0);
                if (S::BFX && (PINN_F2_SWP & 1)) {
                    gemm_swp(Xin, wb, A, acc_mode(1));
                    if (bias_after)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                                PINN_UNROLL for (int e = 0; e < 4; ++e) A[pg * C][t][e] += bv[t][e];
                } else if (S::BFX) {
                    constexpr bool ACC2 = (PINN_F2_SPLIT_ACC2 & 1) != 0;
                    vfloat4 As[ACC2 ? NG : 1][ACC2 ? MTW : 1];
                    if (ACC2)
                        PINN_UNROLL for (int q = 0; q < NG; ++q)
                            PINN_UNROLL for (int t = 0; t < MTW; ++t) As[q][t] = vzero4();
                    PINN_UNROLL for (int kb = 0; kb < S::KB; ++kb)
                        PINN_UNROLL for (int q = 0; q < NG; ++q) {
                            vbf8 bb[3];
                            PINN_UNROLL for (int sp = 0; sp < 3; ++sp) bb[sp] = ld_bfrag(Xin, (q * S::KB + kb) * 3 + sp);
                            PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                                if (ACC2) mfma_split2(wb[kb][t], bb, A[q][t], As[ACC2 ? q : 0][ACC2 ? t : 0]);
                                else A[q][t] = mfma_split(wb[kb][t], bb, A[q][t]);
                            }
                        }
                    if (ACC2)
                        PINN_UNROLL for (int q = 0; q < NG; ++q)
                            PINN_UNROLL for (int t = 0; t < MTW; ++t)
                                PINN_UNROLL for (int e = 0; e < 4; ++e) A[q][t][e] += As[q][t][e];
                    if (WPRE && F2_GEMM_AHEAD > 0) sched_gemm_prefetch<S::KB * NG, MTW * 6, F2_GEMM_AHEAD, S::BFX_TR ? 6 : 3>();
                }
                PINN_UNROLL for (int mi = 0; mi < (S::BFX ? 0 : MT); ++mi) {
                    if (!WPRE)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            wf[0][t] = ld_wf(hl, t, mi);
                    vfloat4 b4[NG];                                          // all B fragments of this k-block first: their LDS latency overlaps
                    PINN_UNROLL for (int q = 0; q < NG; ++q) b4[q] = lds_load4(Xin, vint(((q * MT + mi) * 64) * 4) + (lane << 2));
                    PINN_UNROLL for (int q = 0; q < NG; ++q)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            PINN_UNROLL for (int rr = 0; rr < 4; ++rr) A[q][t] = mfma16(wf[WPRE ? mi : 0][t][rr], b4[q][rr], A[q][t]);
                }
                if (!S::BFX && WPRE && F2_GEMM_AHEAD > 0) sched_gemm_prefetch<MT * NG, MTW * 4, F2_GEMM_AHEAD>();
                wave_prio(1);
                STAMP(2)
                act_forward(A, hl + 1);
                STAMP(3)
            }
            // the tape waves request their hoisted source channels now: the latency hides under the output layer + its barrier
            // instead of sitting in the serialised tape phase (registers are held for one short phase only)
            if (MODE == MODE_FUSED || TAPE_ONLY)
                if (w < PG)
                    PINN_UNROLL for (int j = 0; j < SRC_PRE; ++j)
                        if (j < T.nsrc) srcv[j] = gload_masked(T.src, vint(j * T.N + pbase + 16 * w) + c, valid_w);
            // output layer HP -> 1: per-wave partial dot over its neurons, summed across the 4 waves through LDS
            {
                PINN_UNROLL for (int q = 0; q < NG; ++q) {
                    vfloat s = vfloat(0.f);
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int r = 0; r < 4; ++r) s = vfma(wL[t][r], A[q][t][r], s);
                    s = xrow_allsum(s);
                    lds_store(UP, vint((w * NG + q) * 16) + c, s);             // all four row groups hold the same value
                }
                wg_barrier();
            }
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int ch = 0; ch < C; ++ch) {
                    vfloat s = vfloat(0.f);
                    PINN_UNROLL for (int ws = 0; ws < S::NW; ++ws) s = s + lds_load(UP, vint((ws * NG + pg * C + ch) * 16) + c);
                    U[pg][ch] = (ch == 0) ? s + vfloat(bL) : s;
                }
        } else {
            // reverse launch of the two-launch path: no forward pass.  The last hidden layer's record comes from this tile's slot in
            // HBM; its post-activation jets (needed for the output layer's weight gradient) are recomputed lane-locally.
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int ch = 0; ch < C; ++ch) U[pg][ch] = vfloat(0.f);
            if (LH == 1) {
                PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                    const int n0 = 16 * (w * MTW + t);
                    vfloat4 b1 = ub_load4(PB, S::off_b(0) + n0, g << 2);
                    vfloat4 w1[D];
                    PINN_UNROLL for (int i = 0; i < D; ++i) w1[i] = ub_load4(PB, S::OFF_W1 + i * HP + n0, g << 2);
                    PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                        vfloat4 z = b1;
                        PINN_UNROLL for (int i = 0; i < D; ++i)
                            PINN_UNROLL for (int r = 0; r < 4; ++r) z[r] = vfma(w1[i][r], x[pg][i], z[r]);
                        if (!SINACT) z = act_value4<SINACT>(act, z);
                        Rlast[pg * C][t] = z;
                        PINN_UNROLL for (int kf = 0; kf < NFIRST; ++kf) Rlast[pg * C + 1 + kf][t] = w1[S::first_axis(kf)];
                        PINN_UNROLL for (int ch = 1 + NFIRST; ch < C; ++ch) Rlast[pg * C + ch][t] = vzero4();
                    }
                }
            } else {
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        Rlast[q][t] = ub_load4(RB, (((LH - 2) * NG + q) * MT + w * MTW + t) * 256, lane << 2);
            }
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int r = 0; r < 4; ++r) {
                        vfloat zz[C], dd[ND];
                        PINN_UNROLL for (int k2 = 0; k2 < C; ++k2) zz[k2] = Rlast[pg * C + k2][t][r];
                        act_derivs_n<J::NORD - 1, SINACT>(act, zz[0], dd);
                        jet_forward<J>(zz, dd);
                        A[pg * C][t][r] = act_from_record<SINACT>(Rlast[pg * C][t][r]);
                        PINN_UNROLL for (int k2 = 1; k2 < C; ++k2) A[pg * C + k2][t][r] = zz[k2];
                    }
        }
        STAMP(4)
        if (IS_FWD) {
            if (w == 0)
                PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                    vint p = vint(pbase + 16 * pg) + c;
                    PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                        gstore_masked(T.out, vint(ch * T.N) + p, U[pg][ch], vand(valid[pg], g0));
                }
            wg_barrier();                                                   // UP is rewritten by the next tile
            continue;
        }

        // =========================== residual tape (vector registers) ===========================
        // Wave pg interprets the program for point group pg (ONE copy of the interpreter in the instruction stream) and
        // broadcasts the seeds ubar = dL/d(jet channel) to the other waves through LDS.
        // records parked in the scratch slab are requested one phase before they are needed (SPRE: when they take <= 24 registers);
        // the first request (last stored layer) goes out before the wait for the tape waves' seeds
        constexpr bool SPRE = (NG * MTW * 4 <= F2_SPRE_MAX);
        vfloat4 Snext[SPRE ? NG : 1][SPRE ? MTW : 1];
        auto load_record = [&](int hl) {                                     // record of hidden layer hl (1 <= hl <= LH-2)
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    Snext[q][t] = (!RECIN && rec_in_lds(hl, q)) ? lds_load4(RL, rl_off(hl, q, t))
                                                             : ub_load4(RECIN ? RB : SB, (((hl - 1) * NG + q) * MT + w * MTW + t) * 256, lane << 2);
            sched_fence();
        };
        vfloat ubar[PG][C];
        if (IS_GRADIN) {
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                vint p = vint(pbase + 16 * pg) + c;
                PINN_UNROLL for (int ch = 0; ch < C; ++ch) ubar[pg][ch] = gload_masked(T.in, vint(ch * T.N) + p, valid[pg]);
            }
            if (SPRE && NHH - 1 >= 1) load_record(NHH - 1);
        } else {
            float* UB = UP + S::NW * NG * 16;
            if (w < PG) {
                wave_prio(3);                       // the other waves of the workgroup wait for this one
                vfloat xin[D], Uin[C];
                vbool vin = valid[0];
                PINN_UNROLL for (int i = 0; i < D; ++i) xin[i] = x[0][i];
                PINN_UNROLL for (int ch = 0; ch < C; ++ch) Uin[ch] = U[0][ch];
                PINN_UNROLL for (int pg = 1; pg < PG; ++pg)
                    if (w == pg) {
                        vin = valid[pg];
                        PINN_UNROLL for (int i = 0; i < D; ++i) xin[i] = x[pg][i];
                        PINN_UNROLL for (int ch = 0; ch < C; ++ch) Uin[ch] = U[pg][ch];
                    }
                if (T.linear && C <= LIN_MAX_C) {
                    // affine residual: no interpreter, no tape registers — a handful of FMAs and the seeds are the constant coefficients
                    vfloat r = vfloat(T.lin_k);
                    PINN_UNROLL for (int ch = 0; ch < C; ++ch) r = vfma(vfloat(T.lin_a[ch < LIN_MAX_C ? ch : 0]), Uin[ch], r);
                    PINN_UNROLL for (int j = 0; j < LIN_MAX_SRC; ++j)
                        if (j < T.nsrc) {
                            vfloat sv;
                            if (j < SRC_PRE) { PINN_UNROLL for (int jj = 0; jj < SRC_PRE; ++jj) if (jj == j) sv = srcv[jj]; }
                            else sv = gload_masked(T.src, vint(j * T.N + pbase + 16 * w) + c, vin);
                            r = vfma(vfloat(T.lin_b[j]), sv, r);
                        }
                    if (MODE == MODE_RESID) {
                        vint p = vint(pbase + 16 * w) + c;
                        gstore_masked(T.out, p, r, vand(vin, g0));
                    } else {
                        vfloat sw = vfloat(1.0f);
                        if (T.pw) sw = gload_masked(T.pw, vint(pbase + 16 * w) + c, vin);
                        vfloat rm = vselect(vin, r * sw, vfloat(0.f));
                        lsum = vdacc_fma(rm, vselect(g0, rm, vfloat(0.f)), lsum);
                        if (MODE != MODE_LOSS) {
                            vfloat rbar = rm * vfloat(T.scale) * sw;
                            PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                                lds_store(UB, vint((w * C + ch) * 16) + c, vselect(vin, rbar * vfloat(T.lin_a[ch < LIN_MAX_C ? ch : 0]), vfloat(0.f)));
                            if (T.src_bar)                                  // coupled equation: seeds of the other networks' reverse launches
                                PINN_UNROLL for (int j = 0; j < LIN_MAX_SRC; ++j)
                                    if (j < T.nsrc) gstore_masked(T.src_bar, vint(j * T.N + pbase + 16 * w) + c, rbar * vfloat(T.lin_b[j]), vand(vin, g0));
                        }
                    }
                } else {
                    const int NP = ga.nparams;
                    const int DT = T.dt;            // tape rows: [coordinates DT | params NP | jet channels C | sources | ops]
                    const int R0 = DT + NP + C + T.nsrc;
                    const rp::Instr* prog = ga.prog + T.prog_off;
                    vtape tv;
                    tape_zero(tv);
                    if (!T.hetero) {
                        PINN_UNROLL for (int i = 0; i < D; ++i) tape_set(tv, i, xin[i]);
                    } else {
                        for (int j = 0; j < DT; ++j) tape_set(tv, j, gload_masked(T.pts, (vint(pbase + 16 * w) + c) * DT + vint(j), vin));
                    }
                    for (int j = 0; j < NP; ++j) tape_set(tv, DT + j, vfloat(ga.params[j]));
                    PINN_UNROLL for (int ch = 0; ch < C; ++ch) tape_set(tv, DT + NP + ch, Uin[ch]);
                    for (int j = 0; j < T.nsrc; ++j) {
                        vfloat sv;
                        if (j < SRC_PRE) { PINN_UNROLL for (int jj = 0; jj < SRC_PRE; ++jj) if (jj == j) sv = srcv[jj]; }
                        else sv = gload_masked(T.src, vint(j * T.N + pbase + 16 * w) + c, vin);
                        tape_set(tv, DT + NP + C + j, sv);
                    }
                    for (int q = 0; q < T.nops; ++q) {
                        const rp::Instr ins = rp::fetch_uniform(prog, q);
                        const vfloat va = tape_get(tv, ins.a), vb = tape_get(tv, ins.b);      // unused operands point at row 0
                        vfloat vo;
                        if (rp::is_bilinear(ins.code)) vo = rp::apply_bilinear<vfloat>(ins, va, vb);
                        else vo = rp::apply<vfloat>(ins.code, va, vb, ins.imm);
                        tape_set(tv, R0 + q, vo);
                    }
                    vfloat r = tape_get(tv, T.out_row);
                    if (MODE == MODE_RESID) {
                        vint p = vint(pbase + 16 * w) + c;
                        gstore_masked(T.out, p, r, vand(vin, g0));
                    } else {
                        vfloat sw = vfloat(1.0f);
                        if (T.pw) sw = gload_masked(T.pw, vint(pbase + 16 * w) + c, vin);
                        vfloat rm = vselect(vin, r * sw, vfloat(0.f));
                        lsum = vdacc_fma(rm, vselect(g0, rm, vfloat(0.f)), lsum);   // this wave's share of the term's (weighted) sum of squares
                        if (MODE != MODE_LOSS) {
                        vfloat rbar = rm * vfloat(T.scale) * sw;
                        vtape ta;
                        tape_zero(ta);
                        tape_set(ta, T.out_row, vfloat(1.0f));
                        for (int q = T.nops - 1; q >= 0; --q) {
                            const rp::Instr ins = rp::fetch_uniform(prog, q);
                            const vfloat va = tape_get(tv, ins.a), vb = tape_get(tv, ins.b), gq = tape_get(ta, R0 + q);
                            vfloat da, db;
                            if (rp::is_bilinear(ins.code)) rp::adjoint_bilinear<vfloat>(ins, va, vb, gq, da, db);   // unused operands: zero adjoint into row 0
                            else rp::adjoint<vfloat>(ins.code, va, vb, tape_get(tv, R0 + q), ins.imm, gq, da, db);
                            tape_set(ta, ins.a, tape_get(ta, ins.a) + da);
                            tape_set(ta, ins.b, tape_get(ta, ins.b) + db);
                        }
                        PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                            lds_store(UB, vint((w * C + ch) * 16) + c, vselect(vin, rbar * tape_get(ta, DT + NP + ch), vfloat(0.f)));   // 4 row groups: same value; masked points may hold inf/NaN
                        if (T.src_bar)                                      // coupled equation: seeds of the other networks' reverse launches
                            for (int j = 0; j < T.nsrc; ++j)
                                gstore_masked(T.src_bar, vint(j * T.N + pbase + 16 * w) + c, rbar * tape_get(ta, DT + NP + C + j), vand(vin, g0));
                        for (int j = 0; j < ga.nparams_estim; ++j) {
                            vfloat pj = vselect(vand(g0, vin), rbar * tape_get(ta, DT + j), vfloat(0.f));
                            PINN_UNROLL for (int jj = 0; jj < MAX_PARAMS; ++jj) if (jj == j) pbar[jj] += pj;
                        }
                        }
                    }
                }
                }
            wave_prio(1);
            if (!TAPE_ONLY) {
                if (SPRE && NHH - 1 >= 1) load_record(NHH - 1);
                wg_barrier();                                                   // seeds of every point group are in UB
                PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                    PINN_UNROLL for (int ch = 0; ch < C; ++ch) ubar[pg][ch] = lds_load(UB, vint((pg * C + ch) * 16) + c);
            }
        }
        if (TAPE_ONLY) { wg_barrier(); continue; }
        STAMP(5)

        // =========================== reverse sweep ===========================
        // exchange buffers of the reverse sweep: XZ holds dZ (B operand of dA, A operand of dW), XA the a-jets (S::BFX_TR: B operand of dW).
        // Transpose-read kernels choose XA = the buffer in which the FORWARD pass left the last hidden layer's input image: for hl = NHH - 1
        // that image IS the a-jet operand, so the first layer of the sweep recomputes, splits and publishes nothing but its dZ.
        constexpr bool FWD_IMG = S::BFX_TR && !RECIN && NHH >= 1 && PINN_F2_TR_FWDIMG;
        float* XA = (S::BFX_TR && ((NHH - 1) & 1) == 0) ? X0 : X1;
        float* XZ = (S::BFX_TR && ((NHH - 1) & 1) == 0) ? X1 : X0;
        auto ajet = [&](const vfloat4 (&Sr)[NG][MTW], int pg, int ch, int t) -> vfloat4 {
            vfloat4 out;
            PINN_UNROLL for (int r = 0; r < 4; ++r) {
                vfloat zz[C], dd[ND];
                PINN_UNROLL for (int k = 0; k < C; ++k) zz[k] = Sr[pg * C + k][t][r];
                if (ch > 0) {
                    act_derivs_n<J::NORD - 1, SINACT>(act, zz[0], dd);
                    jet_forward<J>(zz, dd);                 // (ch is a constant after unrolling: the other channels are dead code)
                }
                out[r] = (ch == 0) ? act_from_record<SINACT>(zz[0]) : zz[ch];
            }
            return out;
        };
        auto act_adjoint = [&](vfloat4 (&G)[NG][MTW], const vfloat4 (&Sr)[NG][MTW]) {
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int r = 0; r < 4; ++r) {
                        vfloat gg[C], ss[C], dd[ND];
                        PINN_UNROLL for (int k = 0; k < C; ++k) { gg[k] = G[pg * C + k][t][r]; ss[k] = Sr[pg * C + k][t][r]; }
                        act_derivs_n<J::NORD, SINACT>(act, ss[0], dd);
                        jet_adjoint<J>(gg, ss, dd);
                        PINN_UNROLL for (int k = 0; k < C; ++k) G[pg * C + k][t][r] = gg[k];
                    }
        };

        vfloat4 G[NG][MTW];
        // PINN_F2_ADJ_IL (H = 64 transpose-read schedule): the activation adjoint of a layer depends on the dA GEMM only, so its PG x 4
        // independent (point group, row) pieces are issued BETWEEN the MFMA groups of the dW GEMM that follows — VALU work in the shadow of
        // the wave's own MFMAs instead of a phase of its own behind them
        // (the 8-wave H = 128 kernels gain most: both waves of a SIMD belong to one workgroup and sit in the same phase, so nothing else fills
        // the matrix pipe's shadow)
        constexpr bool ADJ_IL = S::BFX_TR && !S::CHUNKED && PINN_F2_ADJ_IL && MTW == 1 && (PINN_F2_SWP & 4) != 0 && (TR_OVL || PINN_F2_ADJ_IL >= 2);
        constexpr int ADJ_NCH = PG * 4, ADJ_NGRP = ((NG + 1) / 2) * MT;
        auto adj_piece = [&](const vfloat4 (&Sr)[NG][MTW], int j) {          // piece j = (pg, r) of act_adjoint(G, Sr)
            const int pg = j >> 2, r = j & 3;
            vfloat gg[C], ss[C], dd[ND];
            PINN_UNROLL for (int k = 0; k < C; ++k) { gg[k] = G[pg * C + k][0][r]; ss[k] = Sr[pg * C + k][0][r]; }
            act_derivs_n<J::NORD, SINACT>(act, ss[0], dd);
            jet_adjoint<J>(gg, ss, dd);
            PINN_UNROLL for (int k = 0; k < C; ++k) G[pg * C + k][0][r] = gg[k];
        };
        PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {                        // output layer
            if (w == 0) bLbar += vselect(g0, ubar[pg][0], vfloat(0.f));
            PINN_UNROLL for (int ch = 0; ch < C; ++ch)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int r = 0; r < 4; ++r) {
                        wLbar[t][r] = vfma(ubar[pg][ch], A[pg * C + ch][t][r], wLbar[t][r]);
                        G[pg * C + ch][t][r] = wL[t][r] * ubar[pg][ch];
                    }
        }
        act_adjoint(G, Rlast);
        STAMP(6)

        PINN_UNROLL for (int hl = NHH - 1; hl >= 0; --hl) {
            // G = dZ of hidden layer hl+1 (own tiles).  Inputs of that layer = a-jets of hidden layer hl.
            vfloat4 Sr[NG][MTW];
            if (hl == 0) {              // record of hidden layer 0: K = d, recomputed on the VALU
                PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                    const int n0 = 16 * (w * MTW + t);
                    vfloat4 b1 = ub_load4(PB, S::off_b(0) + n0, g << 2);
                    vfloat4 w1[D];
                    PINN_UNROLL for (int i = 0; i < D; ++i) w1[i] = ub_load4(PB, S::OFF_W1 + i * HP + n0, g << 2);
                    PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                        vfloat4 z = b1;
                        PINN_UNROLL for (int i = 0; i < D; ++i)
                            PINN_UNROLL for (int r = 0; r < 4; ++r) z[r] = vfma(w1[i][r], x[pg][i], z[r]);
                        if (!SINACT) z = act_value4<SINACT>(act, z);
                        Sr[pg * C][t] = z;
                        PINN_UNROLL for (int kf = 0; kf < NFIRST; ++kf) Sr[pg * C + 1 + kf][t] = w1[S::first_axis(kf)];
                        PINN_UNROLL for (int ch = 1 + NFIRST; ch < C; ++ch) Sr[pg * C + ch][t] = vzero4();
                    }
                }
            } else if (SPRE) {
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) Sr[q][t] = Snext[q][t];
            } else {
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        Sr[q][t] = (!RECIN && rec_in_lds(hl, q)) ? lds_load4(RL, rl_off(hl, q, t))
                                                              : ub_load4(RECIN ? RB : SB, (((hl - 1) * NG + q) * MT + w * MTW + t) * 256, lane << 2);
            }
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int r = 0; r < 4; ++r) bbar[hl + 1][t][r] += G[pg * C][t][r];
            if (TR_OVL && (hl < NHH - 1 || RECIN)) wg_barrier();             // every wave's dW GEMM of the layer above (RECIN: of the previous tile) has read X0 / X1
            publish(XZ, G);                                                  // dZ in B-fragment order for dA = W^T dZ
            if (S::BFX_TR && !(FWD_IMG && hl == NHH - 1)) {
                // the a-jets of hidden layer hl (this wave's neuron tiles) as bf16 pieces in XA, laid out like a forward activation: the
                // transpose reads of the dW GEMM turn XZ (dZ, own tile) and XA (every input tile) into its two operands
                vfloat4 AJ[NG][MTW];
                PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int r = 0; r < 4; ++r) {
                            vfloat zz[C], dd[ND];
                            PINN_UNROLL for (int k = 0; k < C; ++k) zz[k] = Sr[pg * C + k][t][r];
                            if (C > 1) {
                                act_derivs_n<J::NORD - 1, SINACT>(act, zz[0], dd);
                                jet_forward<J>(zz, dd);
                            }
                            AJ[pg * C][t][r] = act_from_record<SINACT>(Sr[pg * C][t][r]);
                            PINN_UNROLL for (int k = 1; k < C; ++k) AJ[pg * C + k][t][r] = zz[k];
                        }
                publish(XA, AJ);
            }
            // stage column group q: dZ^T (wave private, [t][column][16 neurons]) and A^T (cooperative, [column][HP]), slot-swizzled
            auto stage_q = [&](int q, float* zt, float* at) {
                PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                    lds_store4(zt, vint((t * 16) * 16) + c * 16 + ((g ^ (c & vint(3))) << 2), G[q][t]);
                    const vint slot = vint(4 * (w * MTW + t)) + g;
                    lds_store4(at, c * HP + (((slot ^ c) & vint(4 * MT - 1)) << 2), ajet(Sr, q / C, q % C, t));
                }
            };
            vfloat4 wacc[S::WBAR_REG ? 1 : MTW][S::WBAR_REG ? 1 : MT];
            constexpr bool WACC_PRELOAD = !S::WBAR_REG && (S::BFX || PINN_F2_WACC_PRELOAD >= 2) && !WPRE && !S::CHUNKED && PINN_F2_WACC_PRELOAD;
            if (!S::WBAR_REG && !WACC_PRELOAD)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int ti = 0; ti < MT; ++ti) wacc[t][ti] = vzero4();
            // dW[own rows][all inputs] += dZ A^T for one column group
            auto dw_q = [&](const float* zt, const float* at) {
                PINN_UNROLL for (int kk = 0; kk < 4; ++kk) {
                    const vint row = vint(4 * kk) + g;
                    vfloat zf[MTW];
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        zf[t] = lds_load(zt, vint((t * 16) * 16) + row * 16 + (((c >> 2) ^ (row & vint(3))) << 2) + (c & vint(3)));
                    PINN_UNROLL for (int hb = 0; hb < MT / 4; ++hb) {
                        const vint slot = vint(16 * hb) + c;
                        vfloat4 a4 = lds_load4(at, row * HP + (((slot ^ row) & vint(4 * MT - 1)) << 2));
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            PINN_UNROLL for (int e = 0; e < 4; ++e) {
                                if (S::WBAR_REG) wbar[hl][t][hb * 4 + e] = mfma16(zf[t], a4[e], wbar[hl][t][hb * 4 + e]);
                                else wacc[t][hb * 4 + e] = mfma16(zf[t], a4[e], wacc[t][hb * 4 + e]);
                            }
                    }
                }
            };
            // dW by LDS transpose reads (S::BFX_TR): column groups 2 qp, 2 qp + 1 are the K = 32 points of one MFMA; an odd tail pair takes
            // zeros for k >= 16 (A operand masked, B operand read from the pair's first column group: finite values)
            constexpr bool DW_ACC2 = S::BFX_TR && ((S::WBAR_REG && (PINN_F2_SPLIT_ACC2 & 4)) || (!S::WBAR_REG && (PINN_F2_SPLIT_ACC2 & 8)));
            vfloat4 ws[DW_ACC2 ? MTW : 1][DW_ACC2 ? MT : 1];
            if (DW_ACC2)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int ti = 0; ti < MT; ++ti) ws[t][ti] = vzero4();
            auto dw_merge = [&]() {                                         // after the last pair of the layer: sums += small pieces (one rounding each)
                if (!DW_ACC2) return;
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                        PINN_UNROLL for (int e = 0; e < 4; ++e) {
                            if (S::WBAR_REG) wbar[hl][t][ti][e] += ws[DW_ACC2 ? t : 0][DW_ACC2 ? ti : 0][e];
                            else wacc[t][ti][e] += ws[DW_ACC2 ? t : 0][DW_ACC2 ? ti : 0][e];
                        }
            };
            auto dw_pair_tr = [&](int qp) {
                const bool full = (2 * qp + 1 < NG);
                auto ld_tr = [&](const float* X, int frag, int h) -> vbf8 {
                    const int base = frag * 256 + h * 128;
                    if (full) return cat_bf8(lds_load_tr_bf4(X, vint(base) + trbq[0]), lds_load_tr_bf4(X, vint(base) + trbq[1]));
                    return cat_bf8(lds_load_tr_bf4(X, vint(base) + trb[0]), lds_load_tr_bf4(X, vint(base) + trb[1]));
                };
                vbf8 za[MTW][3];
                PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                    const int tile = w * MTW + t;
                    PINN_UNROLL for (int sp = 0; sp < 3; ++sp) {
                        za[t][sp] = ld_tr(XZ, (2 * qp * S::KB + (tile >> 1)) * 3 + sp, tile & 1);
                        if (!full) za[t][sp] = bf8_select(klo, za[t][sp]);
                    }
                }
                if (PINN_F2_SWP & 4) {
                    constexpr int NB = 2;
                    vbf8 ab[NB][3];
                    PINN_UNROLL for (int sp = 0; sp < 3; ++sp) ab[0][sp] = ld_tr(XA, (2 * qp * S::KB) * 3 + sp, 0);
                    PINN_UNROLL for (int ti = 0; ti < MT; ++ti) {
                        if (ti + 1 < MT)
                            PINN_UNROLL for (int sp = 0; sp < 3; ++sp) ab[(ti + 1) % NB][sp] = ld_tr(XA, (2 * qp * S::KB + ((ti + 1) >> 1)) * 3 + sp, (ti + 1) & 1);
                        sched_fence();
                        if (ti + 1 < MT) lds_wait<6>(); else lds_wait<0>();      // (this group's operands complete before its MFMA chain)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                            if (DW_ACC2 && S::WBAR_REG) mfma_split2(za[t], ab[ti % NB], wbar[hl][t][ti], ws[DW_ACC2 ? t : 0][DW_ACC2 ? ti : 0]);
                            else if (DW_ACC2) mfma_split2(za[t], ab[ti % NB], wacc[t][ti], ws[DW_ACC2 ? t : 0][DW_ACC2 ? ti : 0]);
                            else if (S::WBAR_REG) wbar[hl][t][ti] = mfma_split(za[t], ab[ti % NB], wbar[hl][t][ti]);
                            else wacc[t][ti] = mfma_split(za[t], ab[ti % NB], wacc[t][ti]);
                        }
                        chain_fence();                                      // (the adjoint pieces below stay BEHIND the chain, not inside it)
                        if (ADJ_IL) {                                       // this group's share of the activation adjoint (see adj_piece)
                            const int gidx = qp * MT + ti;
                            PINN_UNROLL for (int j = 0; j < ADJ_NCH; ++j)
                                if ((j * ADJ_NGRP) / ADJ_NCH == gidx) adj_piece(Sr, j);
                        }
                    }
                    return;
                }
                PINN_UNROLL for (int ti = 0; ti < MT; ++ti) {
                    vbf8 ab[3];
                    PINN_UNROLL for (int sp = 0; sp < 3; ++sp) ab[sp] = ld_tr(XA, (2 * qp * S::KB + (ti >> 1)) * 3 + sp, ti & 1);
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                        if (DW_ACC2 && S::WBAR_REG) mfma_split2(za[t], ab, wbar[hl][t][ti], ws[DW_ACC2 ? t : 0][DW_ACC2 ? ti : 0]);
                        else if (DW_ACC2) mfma_split2(za[t], ab, wacc[t][ti], ws[DW_ACC2 ? t : 0][DW_ACC2 ? ti : 0]);
                        else if (S::WBAR_REG) wbar[hl][t][ti] = mfma_split(za[t], ab, wbar[hl][t][ti]);
                        else wacc[t][ti] = mfma_split(za[t], ab, wacc[t][ti]);
                    }
                }
            };
            // W^T fragments for dA: issued ahead of the dW GEMM, which hides their latency
            vfloat4 wt[WPRE ? MT : 1][MTW];
            vbf8 wtb[S::BFX ? S::KB : 1][S::BFX ? MTW : 1][3];      // split-operand W^T fragments: [k-block of 32 output neurons][own input tile][piece]
            if (S::BFX) {
                PINN_UNROLL for (int kb = 0; kb < S::KB; ++kb)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int sp = 0; sp < 3; ++sp)
                            wtb[kb][t][sp] = (PINN_PROBE & 8) ? wb_probe : ub_load_bf8(PB, S::OFF_WTB + (((hl * MT + w * MTW + t) * S::KB + kb) * 3 + sp) * 256, lane << 2);
                sched_fence();
            } else if (WPRE) {
                PINN_UNROLL for (int mo = 0; mo < MT; ++mo)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        wt[mo][t] = ld_wt(hl, t, mo);
                sched_fence();
            }
            vfloat4 Gn[NG][MTW];
            PINN_UNROLL for (int q = 0; q < NG; ++q)
                PINN_UNROLL for (int t = 0; t < MTW; ++t) Gn[q][t] = vzero4();
            // OVL (H = 64): the staging of the dW operands (VALU + LDS stores) is issued between the MFMAs of the dA GEMM, and
            // the activation adjoint between those of the dW GEMM, instead of in phases of their own:
            //   publish dZ | barrier | dA(q) + stage(q) ... | barrier | dW(q) ... + act_adjoint | next layer
            constexpr bool OVL = WPRE && !S::CHUNKED;
            if (OVL) {
                STAMP(7)
                wg_barrier();                                               // dZ of every wave is in X0; the previous layer's dW reads are done
                STAMP(8)
                if (SPRE && hl - 1 >= 1) load_record(hl - 1);
                wave_prio(0);
                if (S::BFX_TR && (PINN_F2_SWP & 2)) gemm_swp(XZ, wtb, Gn, acc_mode(2));
                PINN_UNROLL for (int q = 0; q < ((S::BFX_TR && (PINN_F2_SWP & 2)) ? 0 : NG); ++q) {
                    if (S::BFX) {
                        constexpr bool ACC2 = (PINN_F2_SPLIT_ACC2 & 2) != 0;
                        vfloat4 Gs[MTW];
                        PINN_UNROLL for (int t = 0; t < MTW; ++t) Gs[t] = vzero4();
                        PINN_UNROLL for (int kb = 0; kb < S::KB; ++kb) {
                            vbf8 bb[3];
                            PINN_UNROLL for (int sp = 0; sp < 3; ++sp) bb[sp] = ld_bfrag(XZ, (q * S::KB + kb) * 3 + sp);
                            PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                                if (ACC2) mfma_split2(wtb[kb][t], bb, Gn[q][t], Gs[t]);
                                else Gn[q][t] = mfma_split(wtb[kb][t], bb, Gn[q][t]);
                            }
                        }
                        if (ACC2)
                            PINN_UNROLL for (int t = 0; t < MTW; ++t)
                                PINN_UNROLL for (int e = 0; e < 4; ++e) Gn[q][t][e] += Gs[t][e];
                    } else {
                        PINN_UNROLL for (int mo = 0; mo < MT; ++mo) {
                            vfloat4 b4 = lds_load4(X0, vint(((q * MT + mo) * 64) * 4) + (lane << 2));
                            PINN_UNROLL for (int t = 0; t < MTW; ++t)
                                PINN_UNROLL for (int rr = 0; rr < 4; ++rr) Gn[q][t] = mfma16(wt[mo][t][rr], b4[rr], Gn[q][t]);
                        }
                    }
                    if (S::BFX_TR) continue;                                // (nothing to stage: dW reads X0 / X1 as they are)
                    stage_q(q, ZT + q * (MTW * 256), X1 + q * 16 * HP);
                }
                if (!S::BFX && F2_GEMM_AHEAD > 0) sched_gemm_prefetch<MT * NG, MTW * 4, F2_GEMM_AHEAD>();
                if (!S::BFX_TR) {
                    wave_prio(1);
                    STAMP(10)
                    wg_barrier();                                           // staged operands complete; X0 free again
                    STAMP(11)
                    wave_prio(0);
                } else {
                    STAMP(10)
                }
                if (S::BFX_TR) {
                    if (ADJ_IL)
                        PINN_UNROLL for (int q = 0; q < NG; ++q)
                            PINN_UNROLL for (int t = 0; t < MTW; ++t) G[q][t] = Gn[q][t];
                    PINN_UNROLL for (int qp = 0; qp < (NG + 1) / 2; ++qp) dw_pair_tr(qp);
                    dw_merge();
                } else {
                    PINN_UNROLL for (int q = 0; q < NG; ++q) dw_q(ZT + q * (MTW * 256), X1 + q * 16 * HP);
                    if (F2_GEMM_AHEAD > 0 && MTW == 1 && MT == 4)
                        sched_gemm_prefetch<4 * NG, 4, F2_GEMM_AHEAD, 2>();
                }
                wave_prio(1);
                if (!ADJ_IL) {
                    PINN_UNROLL for (int q = 0; q < NG; ++q)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t) G[q][t] = Gn[q][t];
                    act_adjoint(G, Sr);
                }
                if (!S::WBAR_REG)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int ti = 0; ti < MT; ++ti) {
                            const vint off = vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2);
                            vfloat4 cur = gload4(slab + S::O_WBAR, off);
                            PINN_UNROLL for (int e = 0; e < 4; ++e) cur[e] += wacc[t][ti][e];
                            gstore4(slab + S::O_WBAR, off, cur);
                        }
                STAMP(9)
                continue;
            }
            // split-operand kernels of this path (H = 128): dA runs FIRST, on the W^T fragments requested above (their 48 registers are
            // dead again before the dW accumulators come alive)
            auto da_split = [&]() {
                if (PINN_F2_SWP & 2) { gemm_swp(XZ, wtb, Gn, acc_mode(2)); return; }
                constexpr bool ACC2 = (PINN_F2_SPLIT_ACC2 & 2) != 0;
                vfloat4 Gs[ACC2 ? NG : 1][ACC2 ? MTW : 1];
                if (ACC2)
                    PINN_UNROLL for (int q = 0; q < NG; ++q)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t) Gs[q][t] = vzero4();
                PINN_UNROLL for (int kb = 0; kb < S::KB; ++kb)
                    PINN_UNROLL for (int q = 0; q < NG; ++q) {
                        vbf8 bb[3];
                        PINN_UNROLL for (int sp = 0; sp < 3; ++sp) bb[sp] = ld_bfrag(XZ, (q * S::KB + kb) * 3 + sp);
                        PINN_UNROLL for (int t = 0; t < MTW; ++t) {
                            if (ACC2) mfma_split2(wtb[kb][t], bb, Gn[q][t], Gs[ACC2 ? q : 0][ACC2 ? t : 0]);
                            else Gn[q][t] = mfma_split(wtb[kb][t], bb, Gn[q][t]);
                        }
                    }
                if (ACC2)
                    PINN_UNROLL for (int q = 0; q < NG; ++q)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            PINN_UNROLL for (int e = 0; e < 4; ++e) Gn[q][t][e] += Gs[q][t][e];
            };
            if (S::CHUNKED) {
                PINN_UNROLL for (int q = 0; q < NG; ++q) {
                    float* cb = X1 + (q & 1) * S::CHSZ;
                    stage_q(q, cb + S::CH_AT + w * S::CH_ZT, cb);
                    wg_barrier();                                           // chunk q complete; chunk q-1's buffer is free again
                    dw_q(cb + S::CH_AT + w * S::CH_ZT, cb);
                }
            } else if (S::BFX_TR) {
                STAMP(7)
                wg_barrier();                                               // dZ (X0) and the a-jets (X1) of every wave are published
                STAMP(8)
                da_split();
                if (WACC_PRELOAD)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                            wacc[t][ti] = gload4(slab + S::O_WBAR, vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2));
                if (ADJ_IL)
                    PINN_UNROLL for (int q = 0; q < NG; ++q)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t) G[q][t] = Gn[q][t];
                PINN_UNROLL for (int qp = 0; qp < (NG + 1) / 2; ++qp) dw_pair_tr(qp);
                    dw_merge();
                STAMP(9)
            } else {
                PINN_UNROLL for (int q = 0; q < NG; ++q) stage_q(q, ZT + q * (MTW * 256), X1 + q * 16 * HP);
                STAMP(7)
                wg_barrier();
                STAMP(8)
                if (!S::BFX && WACC_PRELOAD)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                            wacc[t][ti] = gload4(slab + S::O_WBAR, vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2));
                if (S::BFX) {
                    // slab-resident dW: the running sums of this wave's tiles become the INITIAL accumulators of the dW GEMM — requested
                    // as the W^T fragments' registers fall free, they arrive under the tail of the dA GEMM, and the read-add-write chain after the GEMM shrinks to plain stores
                    da_split();
                    if (WACC_PRELOAD)
                        PINN_UNROLL for (int t = 0; t < MTW; ++t)
                            PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                                wacc[t][ti] = gload4(slab + S::O_WBAR, vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2));
                }
                PINN_UNROLL for (int q = 0; q < NG; ++q) dw_q(ZT + q * (MTW * 256), X1 + q * 16 * HP);
                STAMP(9)
            }
            if (!S::WBAR_REG)
                PINN_UNROLL for (int t = 0; t < MTW; ++t)
                    PINN_UNROLL for (int ti = 0; ti < MT; ++ti) {
                        const vint off = vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2);
                        if (WACC_PRELOAD) { gstore4(slab + S::O_WBAR, off, wacc[t][ti]); continue; }
                        vfloat4 cur = gload4(slab + S::O_WBAR, off);
                        PINN_UNROLL for (int e = 0; e < 4; ++e) cur[e] += wacc[t][ti][e];
                        gstore4(slab + S::O_WBAR, off, cur);
                    }
            if (SPRE && hl - 1 >= 1) load_record(hl - 1);                    // next iteration's record: latency hides under the dA GEMM
            // ---- dA (own input tiles) = W^T dZ ----
            PINN_UNROLL for (int mo = 0; mo < (S::BFX ? 0 : MT); ++mo) {
                if (!WPRE)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        wt[0][t] = ld_wt(hl, t, mo);
                PINN_UNROLL for (int q = 0; q < NG; ++q) {
                    vfloat4 b4 = lds_load4(X0, vint(((q * MT + mo) * 64) * 4) + (lane << 2));
                    PINN_UNROLL for (int t = 0; t < MTW; ++t)
                        PINN_UNROLL for (int rr = 0; rr < 4; ++rr) Gn[q][t] = mfma16(wt[WPRE ? mo : 0][t][rr], b4[rr], Gn[q][t]);
                }
            }
            STAMP(10)
            wg_barrier();                                                   // X0 / X1 free again
            STAMP(11)
            if (!ADJ_IL) {
                PINN_UNROLL for (int q = 0; q < NG; ++q)
                    PINN_UNROLL for (int t = 0; t < MTW; ++t) G[q][t] = Gn[q][t];
                act_adjoint(G, Sr);
            }
            STAMP(12)
        }
        // hidden layer 0: db0, dW1 in the D layout (per-lane partial sums over this lane's column)
        PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
            PINN_UNROLL for (int t = 0; t < MTW; ++t)
                PINN_UNROLL for (int r = 0; r < 4; ++r) {
                    const vfloat gz = G[pg * C][t][r];
                    bbar[0][t][r] += gz;
                    PINN_UNROLL for (int i = 0; i < D; ++i) w1bar[i][t][r] = vfma(gz, x[pg][i], w1bar[i][t][r]);
                    PINN_UNROLL for (int kf = 0; kf < NFIRST; ++kf)
                        PINN_UNROLL for (int i = 0; i < D; ++i)
                            if (i == S::first_axis(kf)) w1bar[i][t][r] += G[pg * C + 1 + kf][t][r];
                }
        if (NHH == 0) wg_barrier();                                         // UP reuse across tiles when there is no layer barrier
        STAMP(13)
    }  // tiles
}

// ---- epilogue: the wave's accumulators into this workgroup's gradient slab (every entry has exactly one writer) ----
template <class S, int PG = S::PG /* tape waves that hold PDE-parameter partials: the largest PG of the launch's members */>
DEV void acc2_store(Acc2<typename S::Shape>& ac, const GroupArgs& ga, int blk, int w, float* lds) {
    constexpr int HP = S::HP, MT = S::MT, MTW = S::MTW, NHH = S::NHH, LH = S::LH, D = S::D, NG = S::NG;
    const vint lane = lane_id();
    const vint g = lane >> 4;
    const vint c = lane & vint(15);
    float* slab = ga.slabs + (size_t)blk * S::SLAB;
    float* UP = lds + S::OFF_UP;
    if (S::WBAR_REG)
        PINN_UNROLL for (int hl = 0; hl < NHH; ++hl)
            PINN_UNROLL for (int t = 0; t < MTW; ++t)
                PINN_UNROLL for (int ti = 0; ti < MT; ++ti)
                    gstore4(slab + S::O_WBAR, vint((((hl * MT + w * MTW + t) * MT + ti) * 64) * 4) + (lane << 2), ac.wbar[hl][t][ti]);
    const vbool c0 = veq(c, 0);
    auto reduce_cols = [&](vfloat v) -> vfloat {       // sum over the 16 column lanes of a row group
        v = row_allsum16(v);
        return v;
    };
    PINN_UNROLL for (int t = 0; t < MTW; ++t)
        PINN_UNROLL for (int r = 0; r < 4; ++r) {
            const vint n = vint(16 * (w * MTW + t) + r) + (g << 2);          // natural neuron index
            PINN_UNROLL for (int l = 0; l < LH; ++l) gstore_masked(slab + S::O_BH + l * HP, n, reduce_cols(ac.bbar[l][t][r]), c0);
            PINN_UNROLL for (int i = 0; i < D; ++i) gstore_masked(slab + S::O_W1 + i * HP, n, reduce_cols(ac.w1bar[i][t][r]), c0);
            gstore_masked(slab + S::O_WL, n, reduce_cols(ac.wLbar[t][r]), c0);
        }
    // PDE-parameter gradients: the tape waves' partial sums meet in wave 0 (fixed order)
    static_assert(4 * MAX_PARAMS <= 16, "UB holds the per-wave parameter partials");
    const vbool all = vlt(lane, 64);
    float* UBp = UP + S::NW * NG * 16;
    if (w > 0 && w < PG)
        PINN_UNROLL for (int j = 0; j < MAX_PARAMS; ++j)
            lds_store(UBp, vint(w * MAX_PARAMS + j) + (lane & vint(0)), vfloat((float)wave_sum_d(ac.pbar[j], all)));
    if (PG > 1) wg_barrier();
    if (w == 0) {
        float s = (float)wave_sum_d(ac.bLbar, all);
        gstore_masked(slab + S::O_BL, vint(0), vfloat(s), veq(lane, 0));
        PINN_UNROLL for (int j = 0; j < MAX_PARAMS; ++j) {
            vfloat sp = vfloat((float)wave_sum_d(ac.pbar[j], all));
            PINN_UNROLL for (int ws = 1; ws < PG; ++ws) sp = sp + lds_load(UBp, vint(ws * MAX_PARAMS + j) + (lane & vint(0)));
            gstore_masked(slab + S::O_P, vint(j), sp, veq(lane, 0));
        }
    }
#if defined(PINN_STAMP) && !defined(PINN_EMU)
    STAMP(14)
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 24; ++i) reinterpret_cast<unsigned*>(slab + S::SLAB - 128)[w * 24 + i] = ac.st_acc[i];
#endif
}

// ---- one launch group: accumulators, every tile of every term of the group, epilogue ----
template <class S, int MODE, int ACTK>
DEV void wave_main2(const GroupArgs& ga, int blk, int nblocks, int w, float* lds) {
    constexpr bool BWD = (MODE == MODE_FUSED || MODE == MODE_GRADIN || MODE == MODE_GRADREC);
    constexpr bool SUMS = (MODE == MODE_FUSED || MODE == MODE_LOSS);
    const int wave = blk * S::NW + w;
    Acc2<typename S::Shape> ac;
    acc2_init<S>(ac, ga, blk, w, BWD);
    if (SUMS)
        for (int j = 0; j < ga.nterms; ++j) ga.losspart[(size_t)wave * ga.nterms_total + ga.terms[j].term_id] = 0.0;
    wave_tiles2<S, MODE, ACTK>(ga, blk, nblocks, w, lds, ac, 0, ga.nterms, 0, ga.ntiles);
    if (SUMS && ac.cur_term >= 0)
        ga.losspart[(size_t)wave * ga.nterms_total + ga.terms[ac.cur_term].term_id] = wave_sum_dd(ac.lsum, veq(lane_id() >> 4, 0));
    if (!BWD) return;
    acc2_store<S>(ac, ga, blk, w, lds);
}

// ---- MERGED launch: the tiles of TWO kernel-family members of one network (e.g. the interior term's jet set and the value-only set
// of the boundary terms) in one persistent launch.  Both members share Shape2, hence the accumulators: a wave's dW / db sums stay in
// registers across both tile lists and are written once — no slab round trip between two chained launches, one ramp, one epilogue.
// Terms [0, sub_terms0) / tiles [0, sub_tiles0) belong to S0, the rest to S1; a workgroup walks its S0 tiles, then its S1 tiles
// (round-robin over the concatenated list, so the members' tile counts need not divide the grid). ----
template <class S0, class S1, int ACTK, int MODE = MODE_FUSED>
DEV void wave_main2m(const GroupArgs& ga, int blk, int nblocks, int w, float* lds) {
    static_assert(same_type<typename S0::Shape, typename S1::Shape>::value, "merged launches need members of one network shape and neuron split");
    static_assert(S0::SLAB == S1::SLAB && S0::PACKED == S1::PACKED, "merged launches share the slab and the packed weight image");
    static_assert(S0::DW_NATURAL == S1::DW_NATURAL && S0::BFX_TR == S1::BFX_TR, "merged launches: one order of the dW tiles in the slab, one exchange-image layout");
    static_assert(MODE == MODE_FUSED || MODE == MODE_LOSS, "merged launches: the fused evaluation and the loss-only evaluation");
    const int wave = blk * S0::NW + w;
    Acc2<typename S0::Shape> ac;
    acc2_init<S0>(ac, ga, blk, w, MODE == MODE_FUSED);
    for (int j = 0; j < ga.nterms; ++j) ga.losspart[(size_t)wave * ga.nterms_total + ga.terms[j].term_id] = 0.0;
    wave_tiles2<S0, MODE, ACTK>(ga, blk, nblocks, w, lds, ac, 0, ga.sub_terms0, 0, ga.sub_tiles0);
    wg_barrier();                                                           // the members lay out the workgroup's LDS differently
    wave_tiles2<S1, MODE, ACTK>(ga, blk, nblocks, w, lds, ac, ga.sub_terms0, ga.nterms, ga.sub_tiles0, ga.ntiles);
    if (ac.cur_term >= 0)
        ga.losspart[(size_t)wave * ga.nterms_total + ga.terms[ac.cur_term].term_id] = wave_sum_dd(ac.lsum, veq(lane_id() >> 4, 0));
    if (MODE != MODE_FUSED) return;
    wg_barrier();
    acc2_store<S0, (S0::PG > S1::PG ? S0::PG : S1::PG)>(ac, ga, blk, w, lds);
}

}  // namespace pk
