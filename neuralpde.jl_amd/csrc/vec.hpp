// vec.hpp — lane-vector vocabulary for the wave-level kernels.
//
// Device build (hipcc, gfx950): vfloat/vint/vbool are plain per-lane scalars, every helper is a
// one-instruction inline wrapper (MFMA builtin, ds_*/global_* accesses, DPP/bpermute shuffles).
//
// PINN_EMU build (g++, tests only): the same names are 64-wide value arrays executed in lock-step
// on the host, with v_mfma_f32_16x16x4_f32 lane semantics restated from
// /opt/skills/guides/cdna_hip_programming.md §3.  The emulation exists so that tests/ can run the
// *actual kernel source* on a GPU-less box against the oracle; it is never linked into
// libpinn_hip.so and is not a product fallback.
#pragma once
#ifndef __HIPCC_RTC__          // (hiprtc, the in-process back end of jit.cpp, has no standard headers and needs none of them here)
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#endif

#ifdef PINN_EMU
// ------------------------------------------------------------------------------------------
// host lock-step emulation of one 64-lane wavefront
// ------------------------------------------------------------------------------------------
#define DEV inline
#define HD inline
#define PINN_UNROLL
namespace wv {
constexpr int W = 64;
struct vbool { bool v[W]; };
struct vint {
    int v[W];
    vint() {}
    vint(int s) { for (int l = 0; l < W; ++l) v[l] = s; }
};
struct vfloat {
    float v[W];
    vfloat() {}
    vfloat(float s) { for (int l = 0; l < W; ++l) v[l] = s; }
};
#define VOP2(T, op) \
    inline T operator op(const T& a, const T& b) { T r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] op b.v[l]; return r; }
VOP2(vfloat, +) VOP2(vfloat, -) VOP2(vfloat, *) VOP2(vfloat, /)
VOP2(vint, +) VOP2(vint, -) VOP2(vint, *) VOP2(vint, &) VOP2(vint, ^) VOP2(vint, |)
#undef VOP2
inline vint operator>>(const vint& a, int s) { vint r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] >> s; return r; }
inline vint operator<<(const vint& a, int s) { vint r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] << s; return r; }
inline vfloat operator-(const vfloat& a) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = -a.v[l]; return r; }
inline vfloat& operator+=(vfloat& a, const vfloat& b) { for (int l = 0; l < W; ++l) a.v[l] += b.v[l]; return a; }
inline vfloat& operator*=(vfloat& a, const vfloat& b) { for (int l = 0; l < W; ++l) a.v[l] *= b.v[l]; return a; }
struct vfloat4 {
    vfloat x[4];
    vfloat& operator[](int i) { return x[i]; }
    const vfloat& operator[](int i) const { return x[i]; }
};
inline vint lane_id() { vint r; for (int l = 0; l < W; ++l) r.v[l] = l; return r; }
inline vfloat vfma(const vfloat& a, const vfloat& b, const vfloat& c) {
    vfloat r; for (int l = 0; l < W; ++l) r.v[l] = std::fmaf(a.v[l], b.v[l], c.v[l]); return r;
}
#define VFN1(name, expr) \
    inline vfloat name(const vfloat& a) { vfloat r; for (int l = 0; l < W; ++l) { float x = a.v[l]; r.v[l] = (expr); } return r; }
VFN1(vtanh, std::tanh(x)) VFN1(vsin, std::sin(x)) VFN1(vcos, std::cos(x)) VFN1(vexp, std::exp(x))
VFN1(vlog, std::log(x)) VFN1(vsqrt, std::sqrt(x)) VFN1(vabs, std::fabs(x)) VFN1(vsinh, std::sinh(x))
VFN1(vcosh, std::cosh(x)) VFN1(vtan, std::tan(x)) VFN1(vrcp, 1.0f / x)
#ifndef PINN_ACT_TANH
#define PINN_ACT_TANH 4
#endif
#ifndef PINN_EMU_LIBM_ACT
// the DEVICE's arithmetic restated (v_exp_f32 / v_rcp_f32 are 1-ulp instructions; exp2f and the division here are at most as far off), so the
// CPU suite sees the cancellation structure of the activation the hardware evaluates; -DPINN_EMU_LIBM_ACT=1: libm (A/B of the activation's share)
inline float emu_tanh_dev(float x) {
#if PINN_ACT_TANH == 0
    return std::fmaf(-2.0f, 1.0f / (exp2f(x * 2.8853900817779268f) + 1.0f), 1.0f);
#else
#if PINN_ACT_TANH >= 3
    const float ax = std::fabs(x);
    const float e = exp2f(std::fmaf(ax, -2.885390043258667f, ax * -3.851926067000022e-08f));
#else
    const float e = exp2f(std::fabs(x) * -2.8853900817779268f);
#endif
    const float s = 1.0f + e;
    float r = 1.0f / s;
#if PINN_ACT_TANH == 2 || PINN_ACT_TANH == 3
    r = std::fmaf(std::fmaf(-s, r, 1.0f), r, r);
#endif
    return std::copysign((1.0f - e) * r, x);
#endif
}
VFN1(vtanh_fast, emu_tanh_dev(x))
VFN1(vsigmoid_fast, 1.0f / (1.0f + exp2f(x * -1.4426950408889634f)))
#else
VFN1(vtanh_fast, std::tanh(x)) VFN1(vsigmoid_fast, 1.0f / (1.0f + std::exp(-x)))
#endif
inline vfloat4 vtanh_fast4(const vfloat4& x) { vfloat4 t; for (int r = 0; r < 4; ++r) t[r] = vtanh_fast(x[r]); return t; }
VFN1(vsign, (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f))
VFN1(vsinpi, std::sin(3.14159265358979323846f * x)) VFN1(vcospi, std::cos(3.14159265358979323846f * x))
#undef VFN1
inline vfloat vpow(const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = std::pow(a.v[l], b.v[l]); return r; }
inline vfloat vmax(const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] > b.v[l] ? a.v[l] : b.v[l]; return r; }
inline vfloat vmin(const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] < b.v[l] ? a.v[l] : b.v[l]; return r; }
inline vbool vlt(const vint& a, int b) { vbool r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] < b; return r; }
inline vbool veq(const vint& a, int b) { vbool r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] == b; return r; }
inline vbool vgt(const vfloat& a, const vfloat& b) { vbool r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] > b.v[l]; return r; }
inline vbool vand(const vbool& a, const vbool& b) { vbool r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l] && b.v[l]; return r; }
inline vfloat vselect(const vbool& m, const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = m.v[l] ? a.v[l] : b.v[l]; return r; }
inline vint vselect(const vbool& m, const vint& a, const vint& b) { vint r; for (int l = 0; l < W; ++l) r.v[l] = m.v[l] ? a.v[l] : b.v[l]; return r; }
// global memory
inline vfloat gload(const float* p, const vint& i) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = p[i.v[l]]; return r; }
inline vfloat gload_masked(const float* p, const vint& i, const vbool& m) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = m.v[l] ? p[i.v[l]] : 0.f; return r; }
inline void gstore(float* p, const vint& i, const vfloat& x) { for (int l = 0; l < W; ++l) p[i.v[l]] = x.v[l]; }
inline void gstore_masked(float* p, const vint& i, const vfloat& x, const vbool& m) { for (int l = 0; l < W; ++l) if (m.v[l]) p[i.v[l]] = x.v[l]; }
inline vfloat4 gload4(const float* p, const vint& i) { vfloat4 r; for (int k = 0; k < 4; ++k) for (int l = 0; l < W; ++l) r.x[k].v[l] = p[i.v[l] + k]; return r; }
inline void gstore4(float* p, const vint& i, const vfloat4& x) { for (int k = 0; k < 4; ++k) for (int l = 0; l < W; ++l) p[i.v[l] + k] = x.x[k].v[l]; }
// 32-row register tape with a wave-uniform dynamic row index (device: VGPR-index mode, s_set_gpr_idx_on)
struct vtape { vfloat r[32]; };
inline vfloat tape_get(const vtape& t, int i) { return t.r[i]; }
inline void tape_set(vtape& t, int i, const vfloat& x) { t.r[i] = x; }
inline void tape_zero(vtape& t) { for (int i = 0; i < 32; ++i) t.r[i] = vfloat(0.f); }
// 16-byte record at a wave-uniform address (device: scalar-cache load)
struct urec16 { int x, y, z, w; };
inline void sched_fence() {}
inline int opaque_lane(int x) { return x; }
template <int N> inline void lds_wait() {}
inline void chain_fence() {}
inline void store_pad() {}
inline void keep_alive(const vfloat4&) {}
inline void wave_prio(int) {}
template <int NGROUPS, int MFMA_PER, int AHEAD, int DS_PER = 1> inline void sched_gemm_prefetch() {}
inline void vsincos(const vfloat& x, vfloat& s, vfloat& c) { for (int l = 0; l < W; ++l) { s.v[l] = std::sin(x.v[l]); c.v[l] = std::cos(x.v[l]); } }
inline urec16 uload16(const void* p) { urec16 r; std::memcpy(&r, p, 16); return r; }
struct urec32 { int v[8]; };
inline urec32 uload32(const void* p) { urec32 r; std::memcpy(&r, p, 32); return r; }
// uniform-base buffer view (device: buffer descriptor in SGPRs + scalar offset + per-lane voffset)
struct ubuf { float* p; };
inline ubuf ub_make(const float* p, size_t) { return ubuf{const_cast<float*>(p)}; }
inline vfloat4 ub_load4(const ubuf& b, int soff, const vint& voff) { vfloat4 r; for (int k = 0; k < 4; ++k) for (int l = 0; l < W; ++l) r.x[k].v[l] = b.p[soff + voff.v[l] + k]; return r; }
inline vfloat4 ub_load4_sc1(const ubuf& b, int soff, const vint& voff) { return ub_load4(b, soff, voff); }
inline void ub_store4(const ubuf& b, int soff, const vint& voff, const vfloat4& x) { for (int k = 0; k < 4; ++k) for (int l = 0; l < W; ++l) b.p[soff + voff.v[l] + k] = x.x[k].v[l]; }
inline vfloat ub_load(const ubuf& b, int soff, const vint& voff) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = b.p[soff + voff.v[l]]; return r; }
inline vfloat ub_load_sc1(const ubuf& b, int soff, const vint& voff) { return ub_load(b, soff, voff); }
// agent-scope write-through stores / loads of data another workgroup reads inside the same launch (device: sc1)
inline void ub_store4_wt(const ubuf& b, const vint& voff, const vfloat4& x) { for (int k = 0; k < 4; ++k) for (int l = 0; l < W; ++l) b.p[voff.v[l] + k] = x.x[k].v[l]; }
inline void gstore_masked_wt(float* p, const vint& i, const vfloat& x, const vbool& m) { for (int l = 0; l < W; ++l) if (m.v[l]) p[i.v[l]] = x.v[l]; }
inline void ustore_wt(double* p, double x) { *p = x; }
inline void ustore_wt(float* p, float x) { *p = x; }
inline float uload_wt(const float* p) { return *p; }
inline double uload_wt(const double* p) { return *p; }
// LDS (per-wave private region in the emulation == a plain array)
inline vfloat lds_load(const float* p, const vint& i) { return gload(p, i); }
inline void lds_store(float* p, const vint& i, const vfloat& x) { gstore(p, i, x); }
inline vfloat4 lds_load4(const float* p, const vint& i) { return gload4(p, i); }
inline void lds_store4(float* p, const vint& i, const vfloat4& x) { gstore4(p, i, x); }
inline void wave_fence() {}
// workgroup barrier: the emulation runs the 4 waves of a workgroup as host threads (spec_registry.hpp)
extern thread_local void (*emu_barrier_hook)(void*);
extern thread_local void* emu_barrier_ctx;
inline void wg_barrier() { if (emu_barrier_hook) emu_barrier_hook(emu_barrier_ctx); }
// grid barrier of the persistent training kernel (pinn_train.hpp): the emulation runs EVERY wave of the launch as a host thread
extern thread_local void (*emu_grid_hook)(void*);
extern thread_local void* emu_grid_ctx;
inline void grid_barrier(unsigned*, unsigned) { if (emu_grid_hook) emu_grid_hook(emu_grid_ctx); }
inline void grid_barrier_wt(unsigned*, unsigned) { if (emu_grid_hook) emu_grid_hook(emu_grid_ctx); }
// cross-lane
inline vfloat shfl_xor(const vfloat& a, int m) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = a.v[l ^ m]; return r; }
// all-reduce sums without LDS traffic (device: DPP row rotations / gfx950 permlane swaps)
inline vfloat row_allsum16(const vfloat& a) {            // over the 16 lanes of each row (lanes sharing l >> 4)
    vfloat v = a;
    for (int n : {8, 4, 2, 1}) { vfloat r; for (int l = 0; l < W; ++l) r.v[l] = v.v[l] + v.v[(l & ~15) | ((l - n) & 15)]; v = r; }
    return v;
}
inline vfloat xrow_allsum(const vfloat& a) {             // over the 4 rows (lanes sharing l & 15)
    vfloat s, t;
    for (int l = 0; l < W; ++l) s.v[l] = a.v[l & ~16] + a.v[l | 16];
    for (int l = 0; l < W; ++l) t.v[l] = s.v[l & ~32] + s.v[l | 32];
    return t;
}
inline float lane0(const vfloat& a) { return a.v[0]; }
inline double wave_sum_d(const vfloat& a, const vbool& m) { double s = 0; for (int l = 0; l < W; ++l) if (m.v[l]) s += (double)a.v[l]; return s; }
// per-lane DOUBLE accumulator of products of floats (the sums of squared residuals): exact products, so a lane's sum does not depend on
// how many tiles it saw — the per-term sums of a sharded evaluation equal the single-device ones to double rounding
struct vdacc { double v[W]; };
inline vdacc vdacc_zero() { vdacc r; for (int l = 0; l < W; ++l) r.v[l] = 0.0; return r; }
inline vdacc vdacc_fma(const vfloat& a, const vfloat& b, const vdacc& c) { vdacc r; for (int l = 0; l < W; ++l) r.v[l] = std::fma((double)a.v[l], (double)b.v[l], c.v[l]); return r; }
inline double wave_sum_dd(const vdacc& a, const vbool& m) { double s = 0; for (int l = 0; l < W; ++l) if (m.v[l]) s += a.v[l]; return s; }
// v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15]; lane l holds D[row=(l>>4)*4+r][col=l&15], r=0..3. k-ordered fmaf chain.
inline vfloat4 mfma16(const vfloat& a, const vfloat& b, const vfloat4& c) {
    vfloat4 d;
    for (int l = 0; l < W; ++l)
        for (int r = 0; r < 4; ++r) {
            int row = (l >> 4) * 4 + r, col = l & 15;
            float acc = c.x[r].v[l];
            for (int k = 0; k < 4; ++k) acc = std::fmaf(a.v[k * 16 + row], b.v[k * 16 + col], acc);
            d.x[r].v[l] = acc;
        }
    return d;
}
inline vfloat4 vzero4() { vfloat4 z; for (int k = 0; k < 4; ++k) z.x[k] = vfloat(0.f); return z; }
// ---- bf16 operands of v_mfma_f32_16x16x32_bf16 (split-operand GEMMs, pinn_kernels2.hpp GEMM_SPLIT) ----
// bf16 = the upper 16 bits of an fp32, conversion rounds to nearest even (v_cvt_pk_bf16_f32).
inline uint16_t bf16_bits(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_float(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float x; std::memcpy(&x, &u, 4); return x; }
struct vbf4 { uint16_t v[4][W]; };      // 4 bf16 per lane (8 bytes)
struct vbf8 { uint16_t v[8][W]; };      // 8 bf16 per lane (16 bytes): one A or B operand of the 16x16x32 MFMA
// x = hi + mid + lo to 24 significant bits: three bf16 pieces of every element (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid))
inline void split3_bf16(const vfloat4& x, vbf4& h, vbf4& m, vbf4& l) {
    for (int k = 0; k < 4; ++k)
        for (int q = 0; q < W; ++q) {
            const float v = x.x[k].v[q];
            const uint16_t hb = bf16_bits(v);
            const float r = v - bf16_float(hb);
            const uint16_t mb = bf16_bits(r);
            const float r2 = r - bf16_float(mb);
            h.v[k][q] = hb; m.v[k][q] = mb; l.v[k][q] = bf16_bits(r2);
        }
}
// 8-byte LDS store / 16-byte LDS and buffer loads of bf16 operands; indices in FLOATS like every other accessor here
inline void split1_bf16(const vfloat4& x, vbf4& h) { vbf4 m, l; split3_bf16(x, h, m, l); }
inline void lds_store_bf4(float* p, const vint& i, const vbf4& x) {
    for (int q = 0; q < W; ++q) { uint16_t t[4] = {x.v[0][q], x.v[1][q], x.v[2][q], x.v[3][q]}; std::memcpy(p + i.v[q], t, 8); }
}
inline vbf8 lds_load_bf8(const float* p, const vint& i) {
    vbf8 r;
    for (int q = 0; q < W; ++q) { uint16_t t[8]; std::memcpy(t, p + i.v[q], 16); for (int k = 0; k < 8; ++k) r.v[k][q] = t[k]; }
    return r;
}
inline vbf8 ub_load_bf8(const ubuf& b, int soff, const vint& voff) { vint i; for (int q = 0; q < W; ++q) i.v[q] = soff + voff.v[q]; return lds_load_bf8(b.p, i); }
inline vbf4 lds_load_bf4(const float* p, const vint& i) {
    vbf4 r;
    for (int q = 0; q < W; ++q) { uint16_t t[4]; std::memcpy(t, p + i.v[q], 8); for (int k = 0; k < 4; ++k) r.v[k][q] = t[k]; }
    return r;
}
// ds_read_b64_tr_b16 (gfx950 LDS transpose read; semantics pinned on hardware by tools/micro/tr_probe.hip): every lane names 8 bytes
// (4 halfwords); a 16-lane group's 16 x 4 halfwords form a [4][16] block M[k][n] (lane 4 k + n / 4 of the group supplies M[k][4 (n / 4) .. + 3])
// and lane n of the group receives the column M[0..3][n]
inline vbf4 lds_load_tr_bf4(const float* p, const vint& i) {
    vbf4 r;
    for (int q = 0; q < W; ++q) {
        const int G = q >> 4, n = q & 15;
        for (int k = 0; k < 4; ++k) {
            uint16_t t[4];
            std::memcpy(t, p + i.v[16 * G + 4 * k + (n >> 2)], 8);
            r.v[k][q] = t[n & 3];
        }
    }
    return r;
}
inline vbf8 cat_bf8(const vbf4& lo, const vbf4& hi) {
    vbf8 r;
    for (int k = 0; k < 4; ++k) for (int q = 0; q < W; ++q) { r.v[k][q] = lo.v[k][q]; r.v[4 + k][q] = hi.v[k][q]; }
    return r;
}
// lanes where m is false get +0.0 in every element
inline vbf8 bf8_select(const vbool& m, const vbf8& x) {
    vbf8 r;
    for (int k = 0; k < 8; ++k) for (int q = 0; q < W; ++q) r.v[k][q] = m.v[q] ? x.v[k][q] : (uint16_t)0;
    return r;
}
// v_mfma_f32_16x16x32_bf16: D[i][j] += sum_{k < 32} A[i][k] B[k][j]; lane l supplies A[i = l & 15][k = 8 (l >> 4) + e] and
// B[k = 8 (l >> 4) + e][j = l & 15], e = 0..7; D as for the 16x16x4 form.  fp32 accumulation, k-ordered here (the hardware's internal
// order differs in the last bits).
inline vfloat4 mfma16x32bf(const vbf8& a, const vbf8& b, const vfloat4& c) {
    vfloat4 d;
    for (int l = 0; l < W; ++l)
        for (int r = 0; r < 4; ++r) {
            const int row = (l >> 4) * 4 + r, col = l & 15;
            float acc = c.x[r].v[l];
            for (int k = 0; k < 32; ++k) acc = std::fmaf(bf16_float(a.v[k & 7][(k >> 3) * 16 + row]), bf16_float(b.v[k & 7][(k >> 3) * 16 + col]), acc);
            d.x[r].v[l] = acc;
        }
    return d;
}
}  // namespace wv
#else
// ------------------------------------------------------------------------------------------
// gfx950 device build
// ------------------------------------------------------------------------------------------
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
// Floating-point contraction is OFF for everything below: every multiply-add that is meant to be fused is written as vfma / fmaf.
// With the default (-ffp-contract=fast) the compiler fuses a*b+c depending on how often a*b is used, so the SAME source expression
// rounded differently in different instantiations of one template (MODE_LOSS vs MODE_FUSED: per-point residuals one ulp apart, found
// on hardware in round 3) and differently from the g++ emulation.  Explicit fusion => a point's residual does not depend on which
// kernel variant evaluated it.  -DPINN_FP_CONTRACT_FAST=1 restores the compiler default (A/B measurements).
#ifndef PINN_FP_CONTRACT_FAST
#define PINN_FP_CONTRACT_FAST 0
#endif
#if !PINN_FP_CONTRACT_FAST
#pragma clang fp contract(off)
#endif
#define DEV __device__ __forceinline__
#define HD __host__ __device__ __forceinline__
#define PINN_UNROLL _Pragma("unroll")
namespace wv {
using vfloat = float;
using vint = int;
using vbool = bool;
typedef float vfloat4 __attribute__((ext_vector_type(4)));
DEV vint lane_id() { return (int)(threadIdx.x & 63); }
DEV vfloat vfma(vfloat a, vfloat b, vfloat c) { return __builtin_fmaf(a, b, c); }
DEV vfloat vtanh(vfloat x) { return tanhf(x); }
DEV vfloat vsin(vfloat x) { return sinf(x); }
DEV vfloat vcos(vfloat x) { return cosf(x); }
DEV vfloat vtan(vfloat x) { return tanf(x); }
DEV vfloat vexp(vfloat x) { return expf(x); }
DEV vfloat vlog(vfloat x) { return logf(x); }
DEV vfloat vsqrt(vfloat x) { return sqrtf(x); }
DEV vfloat vabs(vfloat x) { return fabsf(x); }
DEV vfloat vsinh(vfloat x) { return sinhf(x); }
DEV vfloat vcosh(vfloat x) { return coshf(x); }
DEV vfloat vrcp(vfloat x) { return __builtin_amdgcn_rcpf(x); }
// tanh(x) = 1 - 2/(exp(2x)+1) on v_exp_f32 / v_rcp_f32: absolute error ~1e-7 (the quantity that matters for a
// bounded activation feeding a linear layer); saturates correctly to +-1 for |x| large.
// sin and cos together in ~25 instructions (the library sinf/cosf inline a Payne-Hanek reduction each): two-constant Cody-Waite
// reduction to [-pi/4, pi/4] (exact enough for |x| < ~1e4; network pre-activations are O(1..10)) + the cephes minimax polynomials.
DEV void vsincos(vfloat x, vfloat& s, vfloat& c) {
    const float k = __builtin_rintf(x * 0.63661977236758134f);
    float r = __builtin_fmaf(k, -1.5707963109016418f, x);
    r = __builtin_fmaf(k, -1.5893254712295857e-8f, r);
    const float r2 = r * r;
    const float sp = __builtin_fmaf(r * r2, __builtin_fmaf(r2, __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = __builtin_fmaf(r2, __builtin_fmaf(r2, __builtin_fmaf(r2, __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), -0.5f), 1.0f);
    const int q = (int)k & 3;
    const float sv = (q & 1) ? cp : sp, cv = (q & 1) ? sp : cp;
    s = (q & 2) ? -sv : sv;
    c = ((q + 1) & 2) ? -cv : cv;
}
// exp(2x) and exp(-x) as ONE multiply + v_exp_f32 (2^(x * 2 log2 e)); __expf(2.0f * x) compiled to add, multiply, v_exp
#ifndef PINN_ACT_EXP2
#define PINN_ACT_EXP2 1
#endif
// tanh.  PINN_ACT_TANH = 0 (r01-r04): 1 - 2 / (e^{2x} + 1) with e^{2x} = exp2(x * float32(2 log2 e)) — five instructions, and two flaws that only a
// TRAINED network shows (profiles/r05_theta_variants_ab.txt): (a) float32(2 log2 e) is 1.33e-8 (relative) too small, which scales EVERY
// pre-activation of the network coherently — at a trained theta that is amplified like a perturbation of all weights in one direction (the
// engine's gradient error was 2.4-2.8 x a plain float32 evaluation's); (b) the sum e^{2x} + 1 and the reciprocal are rounded at magnitude ~1
// and the final subtraction turns that into ABSOLUTE errors of up to 2.2e-7 around x = 0.  1: the odd form sign(x) (1 - e) / (1 + e),
// e = e^{-2|x|} in (0, 1]: 1 - e is exact for e >= 1/2 and every later rounding is relative to the result.  2: 1 + one Newton step on the
// 1-ulp v_rcp_f32.  3: 2 + a TWO-CONSTANT exponent fma(|x|, -c_hi, |x| * -c_lo), c_hi + c_lo = 2 log2 e to 48 bits.  4 (the product): 3
// without the Newton step — the reciprocal's rounding is incoherent and measured irrelevant (0.3-1.2 x torch-f32 with or without), eight
// instructions.  Accuracy against double tanh on the device: tools/micro/tanh_probe.hip, profiles/r05_tanh_probe.txt.
#ifndef PINN_ACT_TANH
#define PINN_ACT_TANH 4
#endif
DEV vfloat vtanh_fast(vfloat x) {
#if PINN_ACT_TANH == 0
    const float e = PINN_ACT_EXP2 ? __builtin_amdgcn_exp2f(x * 2.8853900817779268f) : __expf(2.0f * x);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
#else
#if PINN_ACT_TANH >= 3
    const float ax = __builtin_fabsf(x);
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(ax, -2.885390043258667f, ax * -3.851926067000022e-08f));
#else
    const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(x) * -2.8853900817779268f);
#endif
    const float s = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(s);
#if PINN_ACT_TANH == 2 || PINN_ACT_TANH == 3
    r = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
#endif
    return __builtin_copysignf((1.0f - e) * r, x);
#endif
}
// the four activations of one accumulator fragment (r06).  PINN_ACT_TANH_PK = 1: the SAME arithmetic as vtanh_fast (variant 4), bit for bit, written on
// register pairs so that the exponent argument (v_pk_mul_f32 + v_pk_fma_f32 on x itself, |.| and the sign moved into v_exp_f32's source modifiers:
// fma(|x|, -c_hi, |x| * -c_lo) = -|fma(x, c_hi, x * c_lo)| exactly), 1 + e, 1 - e and the final product take one instruction per PAIR: 5.5 VALU
// instructions per activation instead of 7
#ifndef PINN_ACT_TANH_PK
#define PINN_ACT_TANH_PK 1
#endif
typedef float vfloat2_ __attribute__((ext_vector_type(2)));
DEV vfloat4 vtanh_fast4(vfloat4 x) {
#if PINN_ACT_TANH == 4 && PINN_ACT_TANH_PK
    vfloat4 t;
    PINN_UNROLL for (int h = 0; h < 2; ++h) {
        const vfloat2_ xx = {x[2 * h], x[2 * h + 1]};
        const vfloat2_ y = __builtin_elementwise_fma(xx, vfloat2_{2.885390043258667f, 2.885390043258667f}, xx * 3.851926067000022e-08f);
        const vfloat2_ e = {__builtin_amdgcn_exp2f(-__builtin_fabsf(y[0])), __builtin_amdgcn_exp2f(-__builtin_fabsf(y[1]))};
        const vfloat2_ s = e + 1.0f;
        const vfloat2_ r = {__builtin_amdgcn_rcpf(s[0]), __builtin_amdgcn_rcpf(s[1])};
        const vfloat2_ q = (1.0f - e) * r;
        t[2 * h] = __builtin_copysignf(q[0], xx[0]);
        t[2 * h + 1] = __builtin_copysignf(q[1], xx[1]);
    }
    return t;
#else
    vfloat4 t;
    PINN_UNROLL for (int r = 0; r < 4; ++r) t[r] = vtanh_fast(x[r]);
    return t;
#endif
}
DEV vfloat vsigmoid_fast(vfloat x) {
    const float e = PINN_ACT_EXP2 ? __builtin_amdgcn_exp2f(x * -1.4426950408889634f) : __expf(-x);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
DEV vfloat vsign(vfloat x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }
DEV vfloat vsinpi(vfloat x) { return sinpif(x); }
DEV vfloat vcospi(vfloat x) { return cospif(x); }
DEV vfloat vpow(vfloat a, vfloat b) { return powf(a, b); }
DEV vfloat vmax(vfloat a, vfloat b) { return fmaxf(a, b); }
DEV vfloat vmin(vfloat a, vfloat b) { return fminf(a, b); }
DEV vbool vlt(vint a, int b) { return a < b; }
DEV vbool veq(vint a, int b) { return a == b; }
DEV vbool vgt(vfloat a, vfloat b) { return a > b; }
DEV vbool vand(vbool a, vbool b) { return a && b; }
DEV vfloat vselect(vbool m, vfloat a, vfloat b) { return m ? a : b; }
DEV vint vselect(vbool m, vint a, vint b) { return m ? a : b; }
DEV vfloat gload(const float* p, vint i) { return p[i]; }
DEV vfloat gload_masked(const float* p, vint i, vbool m) { return m ? p[i] : 0.f; }
DEV void gstore(float* p, vint i, vfloat x) { p[i] = x; }
DEV void gstore_masked(float* p, vint i, vfloat x, vbool m) { if (m) p[i] = x; }
DEV vfloat4 gload4(const float* p, vint i) { return *reinterpret_cast<const vfloat4*>(p + i); }
DEV void gstore4(float* p, vint i, vfloat4 x) { *reinterpret_cast<vfloat4*>(p + i) = x; }
// Uniform-base buffer view: descriptor in SGPRs + scalar byte offset + ONE per-lane voffset VGPR.  Used for every access
// of the form base[const + f(lane)] inside the tile loop: with flat `global_*` the compiler materialises one 64-bit VGPR
// address per distinct constant (>4 KB apart), hoists hundreds of them out of the loop and spills them
// (cdna_hip_programming.md T8/T20).  `p` must be wave-uniform.
// 32-row tape in vector registers; a wave-uniform runtime row index compiles to VGPR-index mode
// (s_set_gpr_idx_on), not to scratch or LDS.
#ifndef PINN_TAPE_ROWS
#define PINN_TAPE_ROWS 32
#endif
typedef float vtape __attribute__((ext_vector_type(PINN_TAPE_ROWS)));
DEV vfloat tape_get(const vtape& t, int i) { return t[i]; }
DEV void tape_set(vtape& t, int i, vfloat x) { t[i] = x; }
DEV void tape_zero(vtape& t) { PINN_UNROLL for (int i = 0; i < PINN_TAPE_ROWS; ++i) t[i] = 0.f; }
// 16-byte record at a wave-uniform, read-only address, fetched through the scalar cache (s_load_dwordx4) into SGPRs.
// A plain load is not scalarised because the kernel also stores to global memory: it becomes global_load + vmcnt(0)
// (which also drains every outstanding record store) and its fields need waterfall loops to be used as register indices.
// the instruction scheduler may not move anything across this point (keeps a block of prefetch loads where it was written)
DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// a value the compiler must treat as per-lane (divergent): conditions on it become selects (v_cndmask) instead of uniform branches — around a load a
// uniform branch costs the load its place in the batch of loads in flight (pinn_kernels6.hpp)
DEV int opaque_lane(int x) { asm volatile("" : "+v"(x)); return x; }
// Wait until at most N of this wave's LDS operations are outstanding (s_waitcnt lgkmcnt(N); vmcnt / expcnt untouched).  Placed in FRONT of a
// chain of MFMAs that accumulate into one register tuple: left alone, the compiler waits for every operand at its first use, i.e. puts
// s_waitcnt instructions BETWEEN the dependent MFMAs of the chain — and one extra issue state between two MFMAs on the same accumulator costs
// ~43 cycles (MI355X_MICROARCH.md, instruction constants: "a cliff, not a slope"); with the whole group's operands waited for up front the
// chain issues back to back.  The compiler's own wait-count pass accounts for this instruction and drops the waits it makes redundant.
#ifndef PINN_LDS_WAIT_HOIST
#define PINN_LDS_WAIT_HOIST 1
#endif
// closes a chain of dependent MFMAs: the scheduler may not move later instructions (the next group's operand reads, interleaved
// element-wise work) up between the MFMAs of the chain — same cliff as above
#ifndef PINN_CHAIN_FENCE
#define PINN_CHAIN_FENCE 1
#endif
DEV void chain_fence() { if (PINN_CHAIN_FENCE) __builtin_amdgcn_sched_barrier(0); }
template <int N> DEV void lds_wait() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    if (PINN_LDS_WAIT_HOIST) {
        __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
        __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks the wait behind the first MFMA of the chain)
    }
}
// The data registers of a 128-bit buffer store must not be rewritten while the memory pipeline is still reading them.  The compiler
// pads that hazard only for stores WITHOUT an SGPR offset; measured on gfx950 with an SGPR offset (8-wave workgroups, two back-to-back
// record stores, v_pk_mul_f32 rewriting the stored registers 0-2 instructions later) the record in memory came out partly overwritten
// (gradient errors of 1e-3, varying from run to run; -fno-slp-vectorize, -amdgpu-waitcnt-forcezero or a scheduling barrier after the
// stores each made it exact).  store_pad + keep_alive: nothing is scheduled across, and the stored registers stay live, for
// PINN_STORE_PAD wait states after a group of stores.
#ifndef PINN_STORE_PAD
#define PINN_STORE_PAD 8
#endif
DEV void store_pad() {
    if (PINN_STORE_PAD > 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (PINN_STORE_PAD >= 16) asm volatile("s_nop 15");
        else asm volatile("s_nop %0" ::"n"(PINN_STORE_PAD - 1));
    }
}
DEV void keep_alive(const vfloat4& x) {
    if (PINN_STORE_PAD > 0) asm volatile("" ::"v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
}
// Instruction-order request for a GEMM region of NGROUPS x [1 LDS fragment read -> MFMA_PER MFMAs]: the reads run AHEAD groups in front of
// the MFMAs that consume them, so a wave's MFMAs issue back to back instead of waiting one LDS round trip per group (left alone, the
// compiler sinks every ds_read next to its first use to save registers: ds_read, s_waitcnt, 4 x v_mfma, ds_read, s_waitcnt, ...).
template <int NGROUPS, int MFMA_PER, int AHEAD, int DS_PER = 1> DEV void sched_gemm_prefetch() {
    __builtin_amdgcn_sched_group_barrier(0x100, AHEAD * DS_PER, 0);
    PINN_UNROLL for (int i = 0; i < NGROUPS - AHEAD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, MFMA_PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, DS_PER, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, MFMA_PER * AHEAD, 0);
}
// issue priority of this wave against the other wave resident on its SIMD (s_setprio): 0 inside the MFMA clusters, 1 in the element-wise
// phases, 3 for the tape waves the rest of the workgroup waits for (asymmetric priorities between the two waves of a SIMD and no
// priorities at all were measured in rounds 2-3: no gain)
template <int P> DEV void wave_prio_t() { __builtin_amdgcn_s_setprio(P); }
#define wave_prio(P) wave_prio_t<P>()
struct urec16 { int x, y, z, w; };
DEV urec16 uload16(const void* p) {
    typedef int i4 __attribute__((ext_vector_type(4)));
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);          // (the builtin returns int: no sign extension)
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    const unsigned long long u = (unsigned long long)lo | ((unsigned long long)hi << 32);
    const i4 v = *(const __attribute__((address_space(4))) i4*)u;
    return {v.x, v.y, v.z, v.w};
}
struct urec32 { int v[8]; };
DEV urec32 uload32(const void* p) {                          // s_load_dwordx8
    typedef int i8 __attribute__((ext_vector_type(8)));
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    const unsigned long long u = (unsigned long long)lo | ((unsigned long long)hi << 32);
    const i8 v = *(const __attribute__((address_space(4))) i8*)u;
    return {{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]}};
}
struct ubuf { __amdgpu_buffer_rsrc_t r; };
typedef unsigned vuint4 __attribute__((ext_vector_type(4)));
DEV ubuf ub_make(const float* p, size_t nfloats) {
    return ubuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(nfloats * 4), 0x00020000)};
}
DEV vfloat4 ub_load4(ubuf b, int soff, vint voff) {
    return __builtin_bit_cast(vfloat4, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff * 4, soff * 4, 0));
}
// L1-bypassing (sc1) variant for data another wave / an earlier phase may have rewritten
DEV vfloat4 ub_load4_sc1(ubuf b, int soff, vint voff) {
    return __builtin_bit_cast(vfloat4, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff * 4, soff * 4, 16));
}
DEV void ub_store4(ubuf b, int soff, vint voff, vfloat4 x) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vuint4, x), b.r, voff * 4, soff * 4, 0);
}
DEV vfloat ub_load(ubuf b, int soff, vint voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff * 4, soff * 4, 0));
}
// one double per lane through a buffer view over doubles (offsets in doubles); AUX 16 = sc1 (agent-scope load)
template <int AUX> DEV double ub_loadd(ubuf b, int soff, vint voff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(b.r, voff * 8, soff * 8, AUX));
}
template <int AUX> DEV float ub_loadf(ubuf b, int soff, vint voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff * 4, soff * 4, AUX));
}
// values the optimiser must treat as unknown from here on: loop-invariant address arithmetic built on them stays INSIDE the loop instead of
// being hoisted in front of it as hundreds of precomputed (and then spilled) registers
DEV void opaque_v(int& x) { asm volatile("" : "+v"(x)); }
DEV void opaque_s(int& x) { asm volatile("" : "+s"(x)); }
DEV vfloat ub_load_sc1(ubuf b, int soff, vint voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff * 4, soff * 4, 16));
}
// Agent-scope WRITE-THROUGH stores (sc1) and agent-scope loads of data that another workgroup reads / wrote inside the same launch (the
// training kernel's gradient slabs, loss partials and weight image): such data needs no release / acquire fence around the grid barrier
// (cdna_hip_programming.md Guideline 16, forms R1 / R2) — every storing wave drains its stores (s_waitcnt vmcnt(0)) before the arrival.
// 16-byte write-through store through a buffer view (one store: four 4-byte sc1 stores would be the slow "narrow re-issue" form of a bulk
// publish); everything in the per-lane offset, no SGPR offset: the form whose store-data hazard the compiler pads (see store_pad above)
DEV void ub_store4_wt(ubuf b, vint voff, vfloat4 x) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vuint4, x), b.r, voff * 4, 0, 16);
}
DEV void gstore_masked_wt(float* p, vint i, vfloat x, vbool m) { if (m) __hip_atomic_store(p + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void ustore_wt(double* p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void ustore_wt(float* p, float x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV float uload_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV double uload_wt(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV vfloat lds_load(const float* p, vint i) { return p[i]; }
DEV void lds_store(float* p, vint i, vfloat x) { p[i] = x; }
DEV vfloat4 lds_load4(const float* p, vint i) { return *reinterpret_cast<const vfloat4*>(p + i); }
DEV void lds_store4(float* p, vint i, vfloat4 x) { *reinterpret_cast<vfloat4*>(p + i) = x; }
// Orders this wave's LDS traffic as seen by its OTHER lanes: DS ops of one wave execute in issue
// order, so only the compiler must be stopped from reordering a lane's read above another lane's
// write (it cannot see the cross-lane dependence).
DEV void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifndef PINN_PROBE
#define PINN_PROBE 0
#endif
DEV void wg_barrier() { if (!(PINN_PROBE & 4)) __syncthreads(); }
// Grid barrier of a launch whose workgroups are ALL resident (pinn_train.hpp: at most one workgroup per CU, far fewer workgroups than
// CUs).  bar[0]: monotonic arrival counter, zeroed by the host ONCE and running on from launch to launch; `target` = arrivals expected so
// far (the launch's starting count + workgroups x barriers passed; compared as a signed difference, so the counter may wrap); bar[1]:
// time-out flag.  Protocol of cdna_hip_programming.md Guideline 16 in its counter form: every wave drains its
// stores, the workgroup meets, ONE lane releases at agent scope (L2 write-back: the other XCDs' L2s are not coherent with this one),
// arrives, polls the one word with relaxed agent-scope loads + s_sleep, acquires ONCE (drops this CU's stale L1 / non-local L2 lines), and
// the workgroup meets again.  Every spin is bounded: a launch that is not fully resident sets bar[1] and runs to its end with wrong numbers
// instead of hanging the device (the host checks the flag and fails the call).
DEV void grid_barrier(unsigned* bar, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (restates the wait behind the L2 write-back where the compiler cannot drop it: Guideline 16, pitfall 12)
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}
// the same barrier for data exchanged with write-through stores and agent-scope loads only (gstore*_wt / ustore_wt / uload_wt / ub_load*_sc1):
// no release, no acquire — the arrival follows the drain of every wave's stores, the loads behind it bypass this CU's L1
DEV void grid_barrier_wt(unsigned* bar, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    __syncthreads();
}
DEV vfloat shfl_xor(vfloat a, int m) { return __shfl_xor(a, m, 64); }
// all-reduce sums without LDS traffic: v_add_f32_dpp row_ror within the 16-lane rows, v_permlane16/32_swap (gfx950) across rows
template <int CTRL> DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
DEV vfloat row_allsum16(vfloat v) {
    v += dpp_mov<0x128>(v); v += dpp_mov<0x124>(v); v += dpp_mov<0x122>(v); v += dpp_mov<0x121>(v);
    return v;
}
DEV vfloat xrow_allsum(vfloat v) {
    // inline asm: with this compiler the second result of __builtin_amdgcn_permlane{16,32}_swap is mis-selected (the sum
    // came out as r[0] + r[0]).  swap16: odd rows of a <-> even rows of b; swap32: rows 2,3 of a <-> rows 0,1 of b.
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float s = a + b, t = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s), "+v"(t));
    return s + t;
}
DEV float lane0(vfloat a) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a))); }      // (the builtin is int -> int: a float argument would be truncated)
DEV double wave_sum_d(vfloat a, vbool m) {
    double s = m ? (double)a : 0.0;
    PINN_UNROLL for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}
// per-lane DOUBLE accumulator of products of floats (see the emulation section)
using vdacc = double;
DEV vdacc vdacc_zero() { return 0.0; }
DEV vdacc vdacc_fma(vfloat a, vfloat b, vdacc c) { return __builtin_fma((double)a, (double)b, c); }
DEV double wave_sum_dd(vdacc a, vbool m) {
    double s = m ? a : 0.0;
    PINN_UNROLL for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}
DEV vfloat4 mfma16(vfloat a, vfloat b, vfloat4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
DEV vfloat4 vzero4() { return vfloat4{0.f, 0.f, 0.f, 0.f}; }
// ---- bf16 operands of v_mfma_f32_16x16x32_bf16 (see the emulation section) ----
typedef __bf16 vbf4 __attribute__((ext_vector_type(4)));
typedef __bf16 vbf8 __attribute__((ext_vector_type(8)));
// PINN_SPLIT_PK = 1 (r06): the same conversions and subtractions written on register PAIRS — one v_cvt_pk_bf16_f32 per two elements and piece, the
// remainders as v_pk_add_f32 (element by element the compiler paired only part of them: 184 of its 568 conversions in the bench kernel converted ONE
// element and were repeated pairwise for the packed store)
#ifndef PINN_SPLIT_PK
#define PINN_SPLIT_PK 1
#endif
typedef __bf16 vbf2_ __attribute__((ext_vector_type(2)));
DEV void split3_bf16(vfloat4 x, vbf4& h, vbf4& m, vbf4& l) {
#if PINN_SPLIT_PK
    PINN_UNROLL for (int q = 0; q < 2; ++q) {
        const vfloat2_ xx = {x[2 * q], x[2 * q + 1]};
        const vbf2_ hb = __builtin_convertvector(xx, vbf2_);                 // v_cvt_pk_bf16_f32: round to nearest even
        const vfloat2_ r = xx - __builtin_convertvector(hb, vfloat2_);
        const vbf2_ mb = __builtin_convertvector(r, vbf2_);
        const vfloat2_ r2 = r - __builtin_convertvector(mb, vfloat2_);
        const vbf2_ lb = __builtin_convertvector(r2, vbf2_);
        h[2 * q] = hb[0]; h[2 * q + 1] = hb[1];
        m[2 * q] = mb[0]; m[2 * q + 1] = mb[1];
        l[2 * q] = lb[0]; l[2 * q + 1] = lb[1];
    }
#else
    PINN_UNROLL for (int k = 0; k < 4; ++k) {
        const __bf16 hb = (__bf16)x[k];                       // v_cvt_pk_bf16_f32: round to nearest even
        const float r = x[k] - (float)hb;
        const __bf16 mb = (__bf16)r;
        const float r2 = r - (float)mb;
        h[k] = hb; m[k] = mb; l[k] = (__bf16)r2;
    }
#endif
}
DEV void split1_bf16(vfloat4 x, vbf4& h) { PINN_UNROLL for (int k = 0; k < 4; ++k) h[k] = (__bf16)x[k]; }      // (timing probes only)
DEV void lds_store_bf4(float* p, vint i, vbf4 x) { *reinterpret_cast<vbf4*>(p + i) = x; }
DEV vbf8 lds_load_bf8(const float* p, vint i) { return *reinterpret_cast<const vbf8*>(p + i); }
// 8-byte LDS read of an operand half.  PINN_F2_LDS_NOMERGE (default): a volatile access in the LDS address space, which the compiler's
// load/store optimiser leaves alone — it would otherwise pair two such reads into one ds_read2st64_b64, which the LDS serves at HALF the
// rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS table: 8 array cycles per wave-instruction against 2 + 2).
#ifndef PINN_F2_LDS_NOMERGE
#define PINN_F2_LDS_NOMERGE 1
#endif
DEV vbf4 lds_load_bf4(const float* p, vint i) {
#if PINN_F2_LDS_NOMERGE
    return *(const volatile __attribute__((address_space(3))) vbf4*)(p + i);
#else
    return *reinterpret_cast<const vbf4*>(p + i);
#endif
}
DEV vbf4 lds_load_tr_bf4(const float* p, vint i) {             // ds_read_b64_tr_b16 (see the emulation section)
    typedef short s4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(vbf4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p + i)));
}
DEV vbf8 cat_bf8(vbf4 lo, vbf4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
DEV vbf8 bf8_select(vbool m, vbf8 x) {
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4 v = __builtin_bit_cast(i4, x);
    return __builtin_bit_cast(vbf8, i4{m ? v[0] : 0, m ? v[1] : 0, m ? v[2] : 0, m ? v[3] : 0});
}
DEV vbf8 ub_load_bf8(ubuf b, int soff, vint voff) {
    return __builtin_bit_cast(vbf8, __builtin_amdgcn_raw_buffer_load_b128(b.r, voff * 4, soff * 4, 0));
}
DEV vfloat4 mfma16x32bf(vbf8 a, vbf8 b, vfloat4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
}  // namespace wv
#endif

// ------------------------------------------------------------------------------------------
// plain DOUBLE overloads of the vocabulary (both builds): the per-point float64 kernels (pinn_kernels4.hpp) instantiate the activation /
// jet rules (pinn_kernels.hpp) and the residual tape (rprog.hpp) with V = double — one lane, one point, IEEE double, library functions
// ------------------------------------------------------------------------------------------
namespace wv {
HD double vfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
HD double vtanh(double x) { return tanh(x); }
// e^y for -80 <= y <= 0: y = n ln 2 + r, degree-13 Taylor polynomial of e^r on |r| <= ln 2 / 2 (remainder 4e-18), scaled by 2^n
HD double vexp_nonpos(double y) {
    const double n = __builtin_rint(y * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, y);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = __builtin_fma(p, r, 2.08767569878681e-09);
    p = __builtin_fma(p, r, 2.505210838544172e-08);
    p = __builtin_fma(p, r, 2.755731922398589e-07);
    p = __builtin_fma(p, r, 2.7557319223985893e-06);
    p = __builtin_fma(p, r, 2.48015873015873e-05);
    p = __builtin_fma(p, r, 1.984126984126984e-04);
    p = __builtin_fma(p, r, 1.388888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}
// the float64 kernels' tanh: sign(x) (1 - e) / (1 + e), e = e^{-2|x|} — absolute error <= 1.7e-16 (ocml / libm: 0.6e-16) at a quarter of the
// device's tanh(double) cost (776 -> 193 cycles per evaluation, tools/micro/tanh64_probe.hip, profiles/r05_tanh64_probe.txt); the relative
// error of tiny results (2.7e-10 at |x| ~ 1e-7) is immaterial here: activations enter sums of O(1) terms
// n / s for 1 <= s <= 2, |n| <= 1: the hardware reciprocal, two Newton steps and one residual correction (8 instructions, <= 1 ulp) instead of the
// compiler's IEEE division sequence (scaling, fix-up of denormal / infinite operands: 12-13 instructions that these operands never need)
#ifndef PINN_F64_DIV_RCP
#define PINN_F64_DIV_RCP 1
#endif
HD double vdiv_1to2(double n, double s) {
#if PINN_F64_DIV_RCP
#ifdef PINN_EMU
    double r = 1.0 / s;                                  // (v_rcp_f64's result differs from this in its last bits; the Newton steps below take both to the same place)
#else
    double r = __builtin_amdgcn_rcp(s);
#endif
    r = __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-s, r, 1.0), r, r);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-s, q, n), r, q);
#else
    return n / s;
#endif
}
// (the clamps are selects, not fmin: fmin(NaN, 40) = 40 would turn a NaN pre-activation into +-1 and mask a diverged run — ADVICE r05)
HD double vtanh_fast(double x) {
    const double a0 = __builtin_fabs(x);
    const double ax = (a0 > 40.0) ? 40.0 : a0;
    const double e = vexp_nonpos(-2.0 * ax);
    return __builtin_copysign(vdiv_1to2(1.0 - e, 1.0 + e), x);
}
// sigma(x) = 1 / (1 + e^-|x|) for x >= 0, e^-|x| / (1 + e^-|x|) for x < 0: the same exponential and division (ocml's exp alone is ~4x the cost)
HD double vsigmoid_fast(double x) {
    const double a0 = __builtin_fabs(x);
    const double e = vexp_nonpos(-((a0 > 80.0) ? 80.0 : a0));
    return vdiv_1to2(x >= 0.0 ? 1.0 : e, 1.0 + e);
}
HD void vsincos(double x, double& s, double& c) { s = sin(x); c = cos(x); }
HD double vsin(double x) { return sin(x); }
HD double vcos(double x) { return cos(x); }
HD double vtan(double x) { return tan(x); }
HD double vexp(double x) { return exp(x); }
HD double vlog(double x) { return log(x); }
HD double vsqrt(double x) { return sqrt(x); }
HD double vabs(double x) { return fabs(x); }
HD double vsinh(double x) { return sinh(x); }
HD double vcosh(double x) { return cosh(x); }
HD double vrcp(double x) { return 1.0 / x; }
HD double vsign(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); }
HD double vsinpi(double x) { return sin(3.14159265358979323846 * x); }
HD double vcospi(double x) { return cos(3.14159265358979323846 * x); }
HD double vpow(double a, double b) { return pow(a, b); }
HD double vmax(double a, double b) { return a > b ? a : b; }
HD double vmin(double a, double b) { return a < b ? a : b; }
HD bool vgt(double a, double b) { return a > b; }
HD double vselect(bool m, double a, double b) { return m ? a : b; }
}  // namespace wv
