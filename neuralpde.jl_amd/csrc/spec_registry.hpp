// spec_registry.hpp — table of compiled kernel-family members (one per translation unit inst_*.hip).
#pragma once
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "plat.hpp"
#include "kernel_entry.hpp"

namespace pk {

std::deque<SpecInfo>& registry();      // a deque: runtime-specialised kernels (jit.cpp) are appended while engines hold SpecInfo pointers

// A MERGED launch (family 2, wave_main2m): two members of the kernel family with the same network shape — `a` for the first tile list,
// `b` for the second — compiled into one persistent kernel.  `a` / `b` identify the members (compared field by field with the
// launch groups' SpecInfo); `launch` runs MODE_FUSED only.
struct PairInfo {
    SpecInfo a, b;
    int WG_PER_CU, NW, LDS_WG, SCR;      // of the merged kernel: min / common / max / max of the members
    int WG_FWD;                          // resident workgroups per CU of the loss-only variant (min of the members)
    void (*launch)(const GroupArgs&, int mode /* MODE_FUSED | MODE_LOSS */, int blocks, plat_stream);
};
std::deque<PairInfo>& pair_registry();

#ifdef PINN_EMU
// The emulation runs the four waves of a workgroup as four host threads joined by a barrier, so that the COOP dW phase
// (workgroup barriers + shared LDS) executes with real concurrency.
struct EmuBarrier {
    std::mutex m;
    std::condition_variable cv;
    int count = 0, gen = 0, nwaves = 4;
    static void wait(void* p) {
        EmuBarrier* b = (EmuBarrier*)p;
        std::unique_lock<std::mutex> lk(b->m);
        const int g = b->gen;
        if (++b->count == b->nwaves) { b->count = 0; ++b->gen; b->cv.notify_all(); }
        else b->cv.wait(lk, [&] { return b->gen != g; });
    }
};
template <class S, int MODE, int ACTK>
void run_emu(const GroupArgs& ga, int blocks) {
    std::vector<float> lds((size_t)S::LDS_WG);
    for (int b = 0; b < blocks; ++b) {
        std::fill(lds.begin(), lds.end(), 0.f);
        EmuBarrier bar;
        std::thread th[4];
        for (int w = 0; w < 4; ++w)
            th[w] = std::thread([&, w] {
                wv::emu_barrier_hook = &EmuBarrier::wait;
                wv::emu_barrier_ctx = &bar;
                wave_main<S, MODE, ACTK>(ga, b, blocks, w, lds.data());
            });
        for (int w = 0; w < 4; ++w) th[w].join();
    }
}
// the training kernel's grid barrier needs EVERY wave of the launch alive at once: blocks x 4 host threads, one barrier per workgroup
// (wg_barrier) and one over all of them (grid_barrier)
template <class S, int ACTK>
void run_emu_train(const GroupArgs& ga, const TrainArgs& ta, int blocks) {
    std::vector<std::vector<float>> lds((size_t)blocks, std::vector<float>((size_t)S::LDS_WG, 0.f));
    std::vector<EmuBarrier> bars((size_t)blocks);
    EmuBarrier grid;
    grid.nwaves = 4 * blocks;
    std::vector<std::thread> th;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 4; ++w)
            th.emplace_back([&, b, w] {
                wv::emu_barrier_hook = &EmuBarrier::wait;
                wv::emu_barrier_ctx = &bars[(size_t)b];
                wv::emu_grid_hook = &EmuBarrier::wait;
                wv::emu_grid_ctx = &grid;
                wave_train<S, ACTK>(ga, ta, b, blocks, w, lds[(size_t)b].data());
            });
    for (auto& t : th) t.join();
}
template <class S, int MODE, int ACTK>
void run_emu2(const GroupArgs& ga, int blocks) {
    std::vector<float> lds((size_t)S::LDS_WG);
    for (int b = 0; b < blocks; ++b) {
        for (auto& v : lds) v = std::nanf("");       // poison: nothing may be read before it is written
        EmuBarrier bar;
        bar.nwaves = S::NW;
        std::thread th[S::NW];
        for (int w = 0; w < S::NW; ++w)
            th[w] = std::thread([&, w] {
                wv::emu_barrier_hook = &EmuBarrier::wait;
                wv::emu_barrier_ctx = &bar;
                wave_main2<S, MODE, ACTK>(ga, b, blocks, w, lds.data());
            });
        for (int w = 0; w < S::NW; ++w) th[w].join();
    }
}
template <class S0, class S1, int ACTK, int MODE>
void run_emu2m(const GroupArgs& ga, int blocks) {
    std::vector<float> lds((size_t)(S0::LDS_WG > S1::LDS_WG ? S0::LDS_WG : S1::LDS_WG));
    for (int b = 0; b < blocks; ++b) {
        for (auto& v : lds) v = std::nanf("");
        EmuBarrier bar;
        bar.nwaves = S0::NW;
        std::thread th[S0::NW];
        for (int w = 0; w < S0::NW; ++w)
            th[w] = std::thread([&, w] {
                wv::emu_barrier_hook = &EmuBarrier::wait;
                wv::emu_barrier_ctx = &bar;
                wave_main2m<S0, S1, ACTK, MODE>(ga, b, blocks, w, lds.data());
            });
        for (int w = 0; w < S0::NW; ++w) th[w].join();
    }
}
#define PINN_LAUNCH2(S, MODE, ACTK, ga, blocks, st) run_emu2<S, MODE, ACTK>(ga, blocks)
#define PINN_LAUNCH2M(S0, S1, ACTK, MODE, ga, blocks, st) run_emu2m<S0, S1, ACTK, MODE>(ga, blocks)
#define PINN_LAUNCH1(S, MODE, ACTK, ga, blocks, st) run_emu<S, MODE, ACTK>(ga, blocks)
#define PINN_LAUNCH_TRAIN(S, ACTK, ga, ta, blocks, st) run_emu_train<S, ACTK>(ga, ta, blocks)
#else
#define PINN_LAUNCH2(S, MODE, ACTK, ga, blocks, st) hipLaunchKernelGGL((k_wave2<S, MODE, ACTK>), dim3(blocks), dim3(64 * S::NW), 0, st, ga)
#define PINN_LAUNCH2M(S0, S1, ACTK, MODE, ga, blocks, st) hipLaunchKernelGGL((k_wave2m<S0, S1, ACTK, MODE>), dim3(blocks), dim3(64 * S0::NW), 0, st, ga)
#define PINN_LAUNCH1(S, MODE, ACTK, ga, blocks, st) hipLaunchKernelGGL((k_wave<S, MODE, ACTK>), dim3(blocks), dim3(256), 0, st, ga)
#define PINN_LAUNCH_TRAIN(S, ACTK, ga, ta, blocks, st) hipLaunchKernelGGL((k_train<S, ACTK>), dim3(blocks), dim3(256), 0, st, ga, ta)
#endif

// The activation kind is a template parameter of the kernels (ACTK): tanh and sigmoid variants for every spec; sin variants (sincos in
// every jet rule) only for the specs registered with PINN_INSTANTIATE*_SIN.  The launch picks the variant from GroupArgs::act.
template <class S, int ACTK>
void launch_modes2(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    (void)st;
    if (mode == MODE_FUSED) PINN_LAUNCH2(S, MODE_FUSED, ACTK, ga, blocks, st);
    else if (mode == MODE_RESID) PINN_LAUNCH2(S, MODE_RESID, ACTK, ga, blocks, st);
    else if (mode == MODE_GRADIN) PINN_LAUNCH2(S, MODE_GRADIN, ACTK, ga, blocks, st);
    else if (mode == MODE_FWDREC) PINN_LAUNCH2(S, MODE_FWDREC, ACTK, ga, blocks, st);
    else if (mode == MODE_GRADREC) PINN_LAUNCH2(S, MODE_GRADREC, ACTK, ga, blocks, st);
    else if (mode == MODE_LOSS) PINN_LAUNCH2(S, MODE_LOSS, ACTK, ga, blocks, st);
    else PINN_LAUNCH2(S, MODE_FWD, ACTK, ga, blocks, st);
}
template <class S, int ACTK>
void launch_modes1(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    (void)st;
    if (mode == MODE_FUSED) PINN_LAUNCH1(S, MODE_FUSED, ACTK, ga, blocks, st);
    else if (mode == MODE_RESID) PINN_LAUNCH1(S, MODE_RESID, ACTK, ga, blocks, st);
    else if (mode == MODE_GRADIN) PINN_LAUNCH1(S, MODE_GRADIN, ACTK, ga, blocks, st);
    else if (mode == MODE_LOSS) PINN_LAUNCH1(S, MODE_LOSS, ACTK, ga, blocks, st);
    else PINN_LAUNCH1(S, MODE_FWD, ACTK, ga, blocks, st);
}
template <class S> void launch_spec2(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    if (ga.act == ACT_TANH) launch_modes2<S, ACT_TANH>(ga, mode, blocks, st); else launch_modes2<S, ACT_SIGMOID>(ga, mode, blocks, st);
}
template <class S> void launch_spec(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    if (ga.act == ACT_TANH) launch_modes1<S, ACT_TANH>(ga, mode, blocks, st); else launch_modes1<S, ACT_SIGMOID>(ga, mode, blocks, st);
}
// the training kernel of a family-1 spec (tanh / sigmoid; the engine keeps the stand-alone loop for the other activations)
template <class S> void launch_train(const GroupArgs& ga, const TrainArgs& ta, int blocks, plat_stream st) {
    (void)st;
    static_assert(sizeof(GroupArgs) + sizeof(TrainArgs) <= 4096, "kernel arguments of k_train");
    if (ga.act == ACT_TANH) PINN_LAUNCH_TRAIN(S, ACT_TANH, ga, ta, blocks, st); else PINN_LAUNCH_TRAIN(S, ACT_SIGMOID, ga, ta, blocks, st);
}
template <class S0, class S1> void launch_pair2(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    (void)st;
    if (mode == MODE_LOSS) {
        if (ga.act == ACT_TANH) PINN_LAUNCH2M(S0, S1, ACT_TANH, MODE_LOSS, ga, blocks, st); else PINN_LAUNCH2M(S0, S1, ACT_SIGMOID, MODE_LOSS, ga, blocks, st);
    } else {
        if (ga.act == ACT_TANH) PINN_LAUNCH2M(S0, S1, ACT_TANH, MODE_FUSED, ga, blocks, st); else PINN_LAUNCH2M(S0, S1, ACT_SIGMOID, MODE_FUSED, ga, blocks, st);
    }
}
template <class S> void launch_spec2_sin(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    if (ga.act == ACT_SIN) launch_modes2<S, ACT_SIN>(ga, mode, blocks, st); else launch_spec2<S>(ga, mode, blocks, st);
}
template <class S> void launch_spec_sin(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    if (ga.act == ACT_SIN) launch_modes1<S, ACT_SIN>(ga, mode, blocks, st); else launch_spec<S>(ga, mode, blocks, st);
}
template <class S> void launch_spec_mix(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    if (ga.act == ACT_MIXED) launch_modes1<S, ACT_MIXED>(ga, mode, blocks, st); else launch_spec<S>(ga, mode, blocks, st);
}


// ---- family 3 (DGM): one 64-lane wave per block, a lane per point; the weight-gradient contraction as a second kernel ----
inline DgmDwArgs dgm_dw_args(const GroupArgs& ga, const SpecInfo& s, int blocks) {
    DgmDwArgs a;
    std::memset(&a, 0, sizeof a);
    a.scratch = ga.scratch; a.slabs = ga.slabs; a.npad = ga.dgm_npad; a.slab = ga.dgm_slab; a.nblocks = blocks; a.ntiles = ga.ntiles;
    a.M = ga.dgm_modes; a.MPad = s.HP; a.d = s.D; a.L = s.NHH; a.C = s.C; a.nparams = ga.dgm_nparams;
    a.r_s = s.r_s; a.r_rec1 = s.r_rec1; a.r_sr = s.r_sr; a.r_dp1 = s.r_dp1; a.r_dp = s.r_dp; a.r_dpo = s.r_dpo;
    for (int i = 0; i < 8; ++i) a.first_ch[i] = s.first_ch[i];
    a.nterms = ga.nterms;
    for (int j = 0; j < MAX_GROUP_TERMS; ++j) a.terms[j] = ga.terms[j];
    return a;
}
#ifdef PINN_EMU
template <class S, int MODE> void run_emu3(const GroupArgs& ga, int blocks) {
    for (int b = 0; b < blocks; ++b) wave_dgm<S, MODE>(ga, b, blocks);
}
inline void run_dgm_dw(const DgmDwArgs& a, plat_stream) {
    for (int b = 0; b < a.nblocks; ++b)
        for (int e = 0; e < a.nparams; ++e) dgm_dw_entry(e, b, a);
}
#define PINN_LAUNCH3(S, MODE, ga, blocks, st) run_emu3<S, MODE>(ga, blocks)
#else
template <int UNUSED>       // (a template only for its linkage: this header is included by every kernel translation unit)
__global__ void __launch_bounds__(256) k_dgm_dw(const DgmDwArgs a) {
    const int e = (int)(blockIdx.x * 256 + threadIdx.x);
    if (e < a.nparams) dgm_dw_entry(e, (int)blockIdx.y, a);
}
inline void run_dgm_dw(const DgmDwArgs& a, plat_stream st) {
    hipLaunchKernelGGL((k_dgm_dw<0>), dim3((a.nparams + 255) / 256, a.nblocks), dim3(256), 0, st, a);
}
#define PINN_LAUNCH3(S, MODE, ga, blocks, st) hipLaunchKernelGGL((k_dgm<S, MODE>), dim3(blocks), dim3(64), 0, st, ga)
#endif
template <class S> SpecInfo& info3_of();          // the registered SpecInfo of S (defined by PINN_INSTANTIATE_DGM)
template <class S> void launch_spec3(const GroupArgs& ga, int mode, int blocks, plat_stream st) {
    (void)st;
    if (mode == MODE_FUSED) {
        PINN_LAUNCH3(S, MODE_FUSED, ga, blocks, st);
        run_dgm_dw(dgm_dw_args(ga, info3_of<S>(), blocks), st);
    } else if (mode == MODE_RESID) PINN_LAUNCH3(S, MODE_RESID, ga, blocks, st);
    else if (mode == MODE_LOSS) PINN_LAUNCH3(S, MODE_LOSS, ga, blocks, st);
    else PINN_LAUNCH3(S, MODE_FWD, ga, blocks, st);
}

struct Registrar {
    explicit Registrar(const SpecInfo& s) { registry().push_back(s); }
};
template <class S0, class S1>
PairInfo make_pair_info() {
    PairInfo p;
    p.a = make_info2<S0>(nullptr);
    p.b = make_info2<S1>(nullptr);
    p.WG_PER_CU = S0::WG_PER_CU < S1::WG_PER_CU ? S0::WG_PER_CU : S1::WG_PER_CU;
    p.NW = S0::NW;
    p.LDS_WG = S0::LDS_WG > S1::LDS_WG ? S0::LDS_WG : S1::LDS_WG;
    p.SCR = S0::SCR > S1::SCR ? S0::SCR : S1::SCR;
    p.WG_FWD = S0::WG_FWD < S1::WG_FWD ? S0::WG_FWD : S1::WG_FWD;
    p.launch = &launch_pair2<S0, S1>;
    return p;
}

// PAIRS encoding: pair p occupies byte p: low nibble = axis a, high nibble = axis b (a <= b)
#define PINN_PAIR(p, a, b) (((unsigned long long)((a) | ((b) << 4))) << (8 * (p)))

// HI: nibble per axis = highest pure derivative order carried along that axis (0, 3 or 4), e.g. PINN_HI(1, 4) for d4/dx_1^4
#define PINN_HI(axis, order) ((unsigned)(order) << (4 * (axis)))
// forward-Laplacian channel over the axes in `mask` (ORed into the HI argument)
#define PINN_LAP(mask) ((unsigned)(mask) << 24)
// family 2: every spec is registered in the GEMM modes PINN_F2_MODES asks for (bit 0: split-operand bf16 products, bit 1: fp32 MFMAs).
// A shape without split-operand kernels (widths other than 64 / 128) is the same fp32 kernel in both forms and is registered once.
#if PINN_F2_MODES & 1
#define PINN_F2_IF_SPLIT(...) __VA_ARGS__
#else
#define PINN_F2_IF_SPLIT(...)
#endif
#if PINN_F2_MODES & 2
#define PINN_F2_IF_FP32(...) __VA_ARGS__
#else
#define PINN_F2_IF_FP32(...)
#endif
// PINN_F1_TRAIN=0 leaves the training kernels out of a translation unit (A/B builds)
#ifndef PINN_F1_TRAIN
#define PINN_F1_TRAIN 1
#endif
#if PINN_F1_TRAIN
#define PINN_TRAIN_OF(S) (&pk::launch_train<S>)
#else
#define PINN_TRAIN_OF(S) nullptr
#endif
template <class S> struct Launch2Plain { static void fn(const GroupArgs& ga, int mode, int blocks, plat_stream st) { launch_spec2<S>(ga, mode, blocks, st); } };
template <class S> struct Launch2Sin { static void fn(const GroupArgs& ga, int mode, int blocks, plat_stream st) { launch_spec2_sin<S>(ga, mode, blocks, st); } };
template <class S, template <class> class L, bool ENABLE> struct Registrar2 {
    explicit Registrar2(int has_sin) { registry().push_back(make_info2<S>(&L<S>::fn, has_sin)); }
};
template <class S, template <class> class L> struct Registrar2<S, L, false> { explicit Registrar2(int) {} };      // (no kernel is instantiated)
template <class S0, class S1, bool ENABLE> struct PairRegistrar2 { PairRegistrar2() { pair_registry().push_back(make_pair_info<S0, S1>()); } };
template <class S0, class S1> struct PairRegistrar2<S0, S1, false> { PairRegistrar2() {} };
// the fp32 form of a spec is a kernel of its own only where the split form exists — or when this unit compiles the fp32 form alone
#define PINN_F2_FP32_ENABLED(S) (S::HAS_SPLIT || !(PINN_F2_MODES & 1))
#define PINN_INSTANTIATE2_ANY(NAME, LAUNCH, VARIANT, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI)                                   \
    namespace {                                                                                                                  \
    PINN_F2_IF_SPLIT(using NAME##_spec2 = pk::Spec2<HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI, pk::GEMM_SPLIT>;                    \
                     pk::Registrar2<NAME##_spec2, pk::LAUNCH, true> NAME##_reg2(VARIANT);)                                       \
    PINN_F2_IF_FP32(using NAME##_spec2f = pk::Spec2<HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI, pk::GEMM_FP32>;                     \
                    pk::Registrar2<NAME##_spec2f, pk::LAUNCH, PINN_F2_FP32_ENABLED(NAME##_spec2f)> NAME##_reg2f(VARIANT);)       \
    }
#define PINN_INSTANTIATE2_HI(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI) PINN_INSTANTIATE2_ANY(NAME, Launch2Plain, 0, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI)
#define PINN_INSTANTIATE_HI(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI)                  \
    namespace {                                                                              \
    using NAME##_spec = pk::Spec<HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI>;                  \
    pk::Registrar NAME##_reg(pk::make_info<NAME##_spec>(&pk::launch_spec<NAME##_spec>, 0, PINN_TRAIN_OF(NAME##_spec))); \
    }
// the same with the sin-activation kernels compiled in as well
#define PINN_INSTANTIATE2_HI_SIN(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI) PINN_INSTANTIATE2_ANY(NAME, Launch2Sin, 1, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI)
#define PINN_INSTANTIATE_HI_SIN(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI)              \
    namespace {                                                                              \
    using NAME##_spec = pk::Spec<HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI>;                  \
    pk::Registrar NAME##_reg(pk::make_info<NAME##_spec>(&pk::launch_spec_sin<NAME##_spec>, 1, PINN_TRAIN_OF(NAME##_spec))); \
    }
// family 1 spec that also carries the per-layer tanh / sigmoid variant (small nets such as the reference's Dense(1, 8, tanh), Dense(8, 8, sigma))
#define PINN_INSTANTIATE_HI_MIX(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI)                  \
    namespace {                                                                              \
    using NAME##_spec = pk::Spec<HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, HI>;                  \
    pk::Registrar NAME##_reg(pk::make_info<NAME##_spec>(&pk::launch_spec_mix<NAME##_spec>, 2, PINN_TRAIN_OF(NAME##_spec))); \
    }
// DGM network (family 3): modes padded to MP, L gated layers, D inputs, jet set, gate activation ACT1, output-gate activation ACT2
#define PINN_INSTANTIATE_DGM(NAME, MP, L, D, D1MASK, PAIRS, NPAIR, HI, ACT1, ACT2)           \
    namespace pk {                                                                           \
    using NAME##_spec3 = Spec3<MP, L, D, D1MASK, PAIRS, NPAIR, HI, ACT1, ACT2>;              \
    template <> SpecInfo& info3_of<NAME##_spec3>() { static SpecInfo s = make_info3<NAME##_spec3>(&launch_spec3<NAME##_spec3>); return s; } \
    }                                                                                        \
    namespace {                                                                              \
    pk::Registrar NAME##_reg3(pk::info3_of<pk::NAME##_spec3>());                             \
    }
// merged launch of two family-2 members of one network shape (tanh / sigmoid): member A = (D1MASK, PAIRS, NPAIR, PG, HI) of the first
// tile list (normally the interior term's jet set), member B of the second (normally the value-only set of the boundary terms)
#define PINN_INSTANTIATE2_PAIR(NAME, HP, NHH, D, D1MASK_A, PAIRS_A, NPAIR_A, PG_A, HI_A, D1MASK_B, PAIRS_B, NPAIR_B, PG_B, HI_B) \
    namespace {                                                                              \
    PINN_F2_IF_SPLIT(using NAME##_pa = pk::Spec2<HP, NHH, D, D1MASK_A, PAIRS_A, NPAIR_A, PG_A, HI_A, pk::GEMM_SPLIT>;  \
                     using NAME##_pb = pk::Spec2<HP, NHH, D, D1MASK_B, PAIRS_B, NPAIR_B, PG_B, HI_B, pk::GEMM_SPLIT>;  \
                     pk::PairRegistrar2<NAME##_pa, NAME##_pb, true> NAME##_regp;)            \
    PINN_F2_IF_FP32(using NAME##_paf = pk::Spec2<HP, NHH, D, D1MASK_A, PAIRS_A, NPAIR_A, PG_A, HI_A, pk::GEMM_FP32>;   \
                    using NAME##_pbf = pk::Spec2<HP, NHH, D, D1MASK_B, PAIRS_B, NPAIR_B, PG_B, HI_B, pk::GEMM_FP32>;   \
                    pk::PairRegistrar2<NAME##_paf, NAME##_pbf, PINN_F2_FP32_ENABLED(NAME##_paf)> NAME##_regpf;)        \
    }
#define PINN_INSTANTIATE2(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG) PINN_INSTANTIATE2_HI(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, 0u)
#define PINN_INSTANTIATE(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG) PINN_INSTANTIATE_HI(NAME, HP, NHH, D, D1MASK, PAIRS, NPAIR, PG, 0u)

}  // namespace pk
