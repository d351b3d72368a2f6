// kernel_entry.hpp — what a translation unit that INSTANTIATES kernels needs and nothing else: the description of a compiled family member
// (SpecInfo, filled from the kernel templates' compile-time constants) and the __global__ entry points.  Free of host-only headers, so the
// same text compiles three ways: hipcc (ahead-of-time table inst_*.hip and the hipcc back end of jit.cpp), g++ -DPINN_EMU (tests), and
// hiprtc IN PROCESS (the compiler-free back end of jit.cpp: no standard headers, no host code; SpecInfo comes back from a one-thread kernel).
#pragma once
#include "pinn_kernels.hpp"
#include "pinn_kernels2.hpp"
#include "pinn_kernels3.hpp"
#include "pinn_train.hpp"
#ifdef __HIPCC_RTC__
typedef struct ihipStream_t* plat_stream;       // (= hipStream_t; only the type of SpecInfo::launch's last parameter matters here)
#else
#include "plat.hpp"
#endif

namespace pk {

constexpr int MAX_GEN_CHANNELS = 24;

struct SpecInfo {
    int family;                      // 1: one wave per tile (pinn_kernels.hpp); 2: neuron-split workgroups (pinn_kernels2.hpp)
    int WG_PER_CU;
    int WG_FWD;                      // resident workgroups per CU of the forward-only launches (family 2: up to four waves per SIMD)
    int NW;                          // waves per workgroup (family 1: 4 independent waves; family 2: 4, or 8 at H = 128)
    int HP, NHH, D;
    unsigned D1MASK;
    unsigned long long PAIRS;
    unsigned HI;                     // bits 0..23: nibble per axis = highest pure derivative order carried (0, 3 or 4); bits 24..31: LAP
    unsigned LAP;                    // axis mask of the forward-Laplacian channel (0: none)
    int NPAIR, PG, C, NG, TP, MT, LH, NFIRST;
    int PACKED, SLAB, SCR, LDS_WG, COOP, SH, PW;
    int has_sin;                     // extra kernel variants compiled for this spec: bit 0 = sin activation, bit 1 = per-layer tanh / sigmoid (ACT_MIXED)
    int REC;                         // floats per tile of the HBM record store (MODE_FWDREC / MODE_GRADREC); 0: not supported
    int jit;                         // 1: specialised at run time (jit.cpp), 0: from the ahead-of-time table
    int act1, act2, dgm_rows;        // family 3 (DGM): gate / output-gate activation kinds, scratch rows per point
    int r_s, r_rec1, r_sr, r_dp1, r_dp, r_dpo, first_ch[8];      // family 3: scratch row bases and first-derivative channels for k_dgm_dw
    int ngen;                        // > 0: general multi-index channel set (jit.cpp gen_*): gen[i] = multi-index of channel i (nibble 0 = order, nibbles 1.. = axes)
    unsigned gen[MAX_GEN_CHANNELS];
    int OFF_W1, OFF_B, OFF_WL, OFF_BL, OFF_WPK, OFF_WTPK;
    int BFX_DW;                      // family 2: dW accumulators in the natural tile order of the transpose-read dW GEMM (column c of tile ti = input 16 ti + c)
    int BFX, OFF_WB, OFF_WTB;        // family 2, split-operand GEMMs (Spec2::BFIMG: the net's weight image carries them): bf16 piece images [NHH][tile][k-block][piece][64][8 bf16], forward / transposed
    int gemm;                        // family 2: GEMM arithmetic of this kernel (pk::GEMM_SPLIT / pk::GEMM_FP32); families 1, 3: GEMM_FP32
    int twin;                        // family 2: 1 when the shape is compiled in both GEMM modes (64- and 128-wide kernels), so a handle's mode selects
    int O_WBAR, O_BFRH, O_BFR0, O_W1, O_WL, O_BL, O_P;
    void (*launch)(const GroupArgs&, int mode, int blocks, plat_stream);
    // family 1: K optimiser iterations in one launch (pinn_train.hpp: k_train, tanh / sigmoid variants); nullptr: not compiled for this spec
    void (*train)(const GroupArgs&, const TrainArgs&, int blocks, plat_stream);
};

template <class S>
HD SpecInfo make_info(void (*launch)(const GroupArgs&, int, int, plat_stream), int has_sin = 0,
                      void (*train)(const GroupArgs&, const TrainArgs&, int, plat_stream) = nullptr) {
    SpecInfo s = {};
    s.train = train;
    s.has_sin = has_sin;
    s.jit = 0;
    s.act1 = s.act2 = s.dgm_rows = 0;
    s.ngen = S::J::GEN ? S::C : 0;
    for (int i = 0; i < MAX_GEN_CHANNELS; ++i) s.gen[i] = (S::J::GEN && i < S::C) ? S::J::gen_channel(i) : 0u;
    s.family = 1; s.WG_PER_CU = 1; s.WG_FWD = 1; s.NW = 4; s.gemm = GEMM_FP32; s.twin = 0; s.BFX = 0; s.BFX_DW = 0; s.OFF_WB = s.OFF_WTB = 0;
    s.HP = S::HP; s.NHH = S::NHH; s.D = S::D; s.D1MASK = S::D1MASK; s.PAIRS = S::PAIRS; s.NPAIR = S::NPAIR; s.HI = S::HI & 0xFFFFFFu; s.LAP = S::J::LAP;
    s.PG = S::PG; s.C = S::C; s.NG = S::NG; s.TP = S::TP; s.MT = S::MT; s.LH = S::LH; s.NFIRST = S::NFIRST;
    s.PACKED = S::PACKED; s.SLAB = S::SLAB; s.SCR = S::SCR; s.LDS_WG = S::LDS_WG; s.COOP = S::COOP ? 1 : 0; s.SH = S::SH; s.PW = S::PW;
    s.REC = 0;
    s.OFF_W1 = S::OFF_W1; s.OFF_B = S::OFF_B; s.OFF_WL = S::OFF_WL; s.OFF_BL = S::OFF_BL;
    s.OFF_WPK = S::OFF_WPK; s.OFF_WTPK = S::OFF_WTPK;
    s.O_WBAR = S::O_WBAR; s.O_BFRH = S::O_BFRH; s.O_BFR0 = S::O_BFR0; s.O_W1 = S::O_W1; s.O_WL = S::O_WL; s.O_BL = S::O_BL; s.O_P = S::O_P;
    s.launch = launch;
    return s;
}

template <class S>
HD SpecInfo make_info2(void (*launch)(const GroupArgs&, int, int, plat_stream), int has_sin = 0) {
    SpecInfo s = {};
    s.has_sin = has_sin;
    s.jit = 0;
    s.act1 = s.act2 = s.dgm_rows = 0;
    s.ngen = S::J::GEN ? S::C : 0;
    for (int i = 0; i < MAX_GEN_CHANNELS; ++i) s.gen[i] = (S::J::GEN && i < S::C) ? S::J::gen_channel(i) : 0u;
    s.family = 2; s.WG_PER_CU = S::WG_PER_CU; s.WG_FWD = S::WG_FWD; s.NW = S::NW; s.gemm = S::BFIMG ? GEMM_SPLIT : GEMM_FP32; s.twin = S::HAS_SPLIT ? 1 : 0;
    s.BFX = S::BFIMG ? 1 : 0; s.OFF_WB = S::OFF_WB; s.OFF_WTB = S::OFF_WTB; s.BFX_DW = S::DW_NATURAL ? 1 : 0;
    s.HP = S::HP; s.NHH = S::NHH; s.D = S::D; s.D1MASK = S::D1MASK; s.PAIRS = S::PAIRS; s.NPAIR = S::NPAIR; s.HI = S::HI & 0xFFFFFFu; s.LAP = S::J::LAP;
    s.PG = S::PG; s.C = S::C; s.NG = S::NG; s.TP = S::TP; s.MT = S::MT; s.LH = S::LH; s.NFIRST = S::NFIRST;
    s.PACKED = S::PACKED; s.SLAB = S::SLAB; s.SCR = S::SCR; s.LDS_WG = S::LDS_WG; s.COOP = 1; s.SH = S::SLAB; s.PW = 0;
    s.REC = S::REC;
    s.OFF_W1 = S::OFF_W1; s.OFF_B = S::OFF_B; s.OFF_WL = S::OFF_WL; s.OFF_BL = S::OFF_BL;
    s.OFF_WPK = S::OFF_WPK; s.OFF_WTPK = S::OFF_WTPK;
    s.O_WBAR = S::O_WBAR; s.O_BFRH = S::O_BH; s.O_BFR0 = S::O_BH; s.O_W1 = S::O_W1; s.O_WL = S::O_WL; s.O_BL = S::O_BL; s.O_P = S::O_P;
    s.launch = launch;
    return s;
}

template <class S>
HD SpecInfo make_info3(void (*launch)(const GroupArgs&, int, int, plat_stream)) {
    SpecInfo s = {};
    s.family = 3; s.WG_PER_CU = 8; s.WG_FWD = 8; s.NW = 1;
    s.HP = S::MP; s.NHH = S::L; s.D = S::D; s.D1MASK = S::D1MASK; s.PAIRS = S::PAIRS; s.NPAIR = S::NPAIR; s.HI = S::HI & 0xFFFFFFu; s.LAP = 0;
    s.PG = 4; s.C = S::C; s.NG = S::C; s.TP = 64; s.MT = 0; s.LH = S::L; s.NFIRST = S::NFIRST;
    s.COOP = 1;
    s.act1 = S::ACT1; s.act2 = S::ACT2; s.dgm_rows = S::ROWS;
    s.r_s = S::R_S; s.r_rec1 = S::R_REC1; s.r_sr = S::R_SR; s.r_dp1 = S::R_DP1; s.r_dp = S::R_DP; s.r_dpo = S::R_DPO;
    for (int i = 0; i < 8; ++i) {
        s.first_ch[i] = -1;
        for (int k = 0; k < S::NFIRST; ++k) if (S::J::first_axis(k) == i) s.first_ch[i] = S::J::CH_FIRST + k;
    }
    s.ngen = S::J::GEN ? S::C : 0;
    for (int i = 0; i < MAX_GEN_CHANNELS; ++i) s.gen[i] = (S::J::GEN && i < S::C) ? S::J::gen_channel(i) : 0u;
    s.launch = launch;
    return s;
}
#ifndef PINN_EMU
// One workgroup = 4 independent waves (one per SIMD); persistent grid of <= #CU workgroups.
// __launch_bounds__(256, 1): one wave per SIMD => the full 512-entry unified VGPR/AGPR file per lane
// is available for the persistent dW accumulators (MI355X_MICROARCH.md "Register files").
template <class S, int MODE, int ACTK>
__global__ void __launch_bounds__(256, 1) k_wave(const GroupArgs ga) {
    __shared__ __attribute__((aligned(16))) float lds_all[S::LDS_WG];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    wave_main<S, MODE, ACTK>(ga, (int)blockIdx.x, (int)gridDim.x, w, lds_all);
}
// K optimiser iterations of a small problem in one launch (pinn_train.hpp); every workgroup must be resident: blocks <= #CU
template <class S, int ACTK>
__global__ void __launch_bounds__(256, 1) k_train(const GroupArgs ga, const TrainArgs ta) {
    __shared__ __attribute__((aligned(16))) float lds_all[S::LDS_WG];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    wave_train<S, ACTK>(ga, ta, (int)blockIdx.x, (int)gridDim.x, w, lds_all);
}
// family 2: two waves per SIMD => at most 256 VGPR+AGPR per lane: two 4-wave workgroups per CU (H = 64), or one 8-wave workgroup
// (H = 128: LDS 100-150 KB per workgroup)
template <class S, int MODE, int ACTK>
__global__ void __launch_bounds__(64 * S::NW, (mode_is_forward_only(MODE) ? S::OCC_FWD : S::OCC)) k_wave2(const GroupArgs ga) {
    __shared__ __attribute__((aligned(16))) float lds_all[mode_is_forward_only(MODE) ? S::LDS_FWD : S::LDS_WG];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    wave_main2<S, MODE, ACTK>(ga, (int)blockIdx.x, (int)gridDim.x, w, lds_all);
}
// merged launch of two family members (wave_main2m): LDS = the larger tile, resident workgroups = the smaller count
template <class S0, class S1> struct Pair2 {
    static constexpr int WG_PER_CU = S0::WG_PER_CU < S1::WG_PER_CU ? S0::WG_PER_CU : S1::WG_PER_CU;
    static constexpr int LDS_WG = S0::LDS_WG > S1::LDS_WG ? S0::LDS_WG : S1::LDS_WG;
    static constexpr int SCR = S0::SCR > S1::SCR ? S0::SCR : S1::SCR;
    static constexpr int OCC = WG_PER_CU * S0::NW / 4;
    static constexpr int WG_FWD = S0::WG_FWD < S1::WG_FWD ? S0::WG_FWD : S1::WG_FWD;
    static constexpr int LDS_FWD = S0::LDS_FWD > S1::LDS_FWD ? S0::LDS_FWD : S1::LDS_FWD;
    static constexpr int OCC_FWD = WG_FWD * S0::NW / 4;
};
template <class S0, class S1, int ACTK, int MODE>
__global__ void __launch_bounds__(64 * S0::NW, (mode_is_forward_only(MODE) ? Pair2<S0, S1>::OCC_FWD : Pair2<S0, S1>::OCC)) k_wave2m(const GroupArgs ga) {
    __shared__ __attribute__((aligned(16))) float lds_all[mode_is_forward_only(MODE) ? Pair2<S0, S1>::LDS_FWD : Pair2<S0, S1>::LDS_WG];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    wave_main2m<S0, S1, ACTK, MODE>(ga, (int)blockIdx.x, (int)gridDim.x, w, lds_all);
}
template <class S, int MODE>
__global__ void __launch_bounds__(64) k_dgm(const GroupArgs ga) { wave_dgm<S, MODE>(ga, (int)blockIdx.x, (int)gridDim.x); }
#endif

}  // namespace pk
