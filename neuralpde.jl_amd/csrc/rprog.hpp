// rprog.hpp — the residual program: a straight-line SSA tape evaluated per collocation point.
//
// This is the device-side stand-in for the Julia function that the reference generates per
// equation with RuntimeGeneratedFunctions (src/discretize.jl:163-175) out of
// parse_equation/_transform_expression (src/symbolic_utilities.jl:132-331, 360-370):
//   rows [0, d)                coordinates of the point (cord[[i],:], discretize.jl:126-131)
//   rows [d, d+np)             PDE parameters p / theta.p (discretize.jl:83-109)
//   rows [d+np, d+np+C)        Taylor-jet channels of the trial function(s): u, du/dx_i, d2u/dx_i dx_j
//                              (the `u(cord,theta,phi)` / `derivative(phi,u,cord,eps,order,theta)` calls,
//                              symbolic_utilities.jl:145-202)
//   rows after that            one per op
// Closed op set = SURVEY.md App. B (every function appearing in the reference's PDE tests/docs).
#pragma once
#include "vec.hpp"

namespace rp {

enum Op : int {
    OP_CONST = 0, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_ADDC, OP_MULC, OP_POWI, OP_POW, OP_POWC,
    OP_SIN, OP_COS, OP_TAN, OP_EXP, OP_LOG, OP_SQRT, OP_ABS, OP_TANH, OP_SINH, OP_COSH, OP_SECH,
    OP_SINPI, OP_COSPI, OP_MAX, OP_MIN,
    OP_DATA,        // nullary: channel (int)imm of the term's user-supplied per-point data (pinn_set_point_data): observations for data-misfit
                    // terms.  Coordinate-only by construction, so it only ever runs in the per-point-set source pass (k_src) or in k_expr.
    OP_COUNT
};

// 32 bytes, pre-decoded by the engine (finalize): the arithmetic ops CONST/ADD/SUB/MUL/NEG/ADDC/MULC are all instances of
//     out = a * (k3 * b + k1) + (k2 * b + k0),     d out/da = k3 * b + k1,     d out/db = k3 * a + k2
// so the fused kernels evaluate them and their adjoints without any dispatch on the op code.
struct Instr {
    int code;
    int a;
    int b;
    float imm;
    float k0, k1, k2, k3;
};
static_assert(sizeof(Instr) == 32, "Instr is fetched as one 32-byte scalar load");

// instruction q of a program whose address is uniform across the wave (fused kernels): scalar fetch
DEV Instr fetch_uniform(const Instr* prog, int q) {
    const wv::urec32 r = wv::uload32(prog + q);
    Instr ins;
    __builtin_memcpy(&ins, &r, 32);
    return ins;
}

#ifdef PINN_TAPE_ROWS
constexpr int MAX_ROWS_FUSED = PINN_TAPE_ROWS;
#else
constexpr int MAX_ROWS_FUSED = 32;
#endif   // LDS tape rows available to the fused kernel (values + adjoints)
constexpr int MAX_ROWS = 256;        // hard limit of the IR

#ifdef PINN_EMU
}  // namespace rp
namespace wv {
// scalar overloads so the same templates serve float (per-thread kernels) and vfloat (wave kernels)
inline float vfma(float a, float b, float c) { return std::fmaf(a, b, c); }
inline float vtanh(float x) { return std::tanh(x); }
inline float vsin(float x) { return std::sin(x); }
inline float vcos(float x) { return std::cos(x); }
inline float vtan(float x) { return std::tan(x); }
inline float vexp(float x) { return std::exp(x); }
inline float vlog(float x) { return std::log(x); }
inline float vsqrt(float x) { return std::sqrt(x); }
inline float vabs(float x) { return std::fabs(x); }
inline float vsinh(float x) { return std::sinh(x); }
inline float vcosh(float x) { return std::cosh(x); }
inline float vrcp(float x) { return 1.0f / x; }
inline float vsign(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }
inline float vsinpi(float x) { return std::sin(3.14159265358979323846f * x); }
inline float vcospi(float x) { return std::cos(3.14159265358979323846f * x); }
inline float vpow(float a, float b) { return std::pow(a, b); }
inline float vmax(float a, float b) { return a > b ? a : b; }
inline float vmin(float a, float b) { return a < b ? a : b; }
inline bool vgt(float a, float b) { return a > b; }
inline float vselect(bool m, float a, float b) { return m ? a : b; }
}  // namespace wv
namespace rp {
#endif

using namespace wv;

template <class T>
DEV T powi(T x, int n) {
    bool neg = n < 0;
    if (neg) n = -n;
    T r = T(1.0f), b = x;
    while (n) {
        if (n & 1) r = r * b;
        b = b * b;
        n >>= 1;
    }
    return neg ? vrcp(r) : r;
}

// forward value of one op (T: lane-value type — vfloat, float, or double in the float64 kernels, whose immediates I are doubles)
template <class T, class I>
DEV T apply(int code, T va, T vb, I imm) {
    switch (code) {
        case OP_CONST: return T(imm);
        case OP_ADD: return va + vb;
        case OP_SUB: return va - vb;
        case OP_MUL: return va * vb;
        case OP_DIV: return va / vb;
        case OP_NEG: return T(0.0f) - va;
        case OP_ADDC: return va + T(imm);
        case OP_MULC: return va * T(imm);
        case OP_POWI: return powi(va, (int)imm);
        case OP_POW: return vpow(va, vb);
        case OP_POWC: return vpow(va, T(imm));
        case OP_SIN: return vsin(va);
        case OP_COS: return vcos(va);
        case OP_TAN: return vtan(va);
        case OP_EXP: return vexp(va);
        case OP_LOG: return vlog(va);
        case OP_SQRT: return vsqrt(va);
        case OP_ABS: return vabs(va);
        case OP_TANH: return vtanh(va);
        case OP_SINH: return vsinh(va);
        case OP_COSH: return vcosh(va);
        case OP_SECH: return vrcp(vcosh(va));
        case OP_SINPI: return vsinpi(va);
        case OP_COSPI: return vcospi(va);
        case OP_MAX: return vmax(va, vb);
        case OP_MIN: return vmin(va, vb);
        default: return T(0.0f);
    }
}

// reverse: given inputs, output value vo and output adjoint g, return (d/da, d/db) contributions
template <class T, class I>
DEV void adjoint(int code, T va, T vb, T vo, I imm, T g, T& da, T& db) {
    const T PI = T(I(3.14159265358979323846));
    da = T(0.0f);
    db = T(0.0f);
    switch (code) {
        case OP_CONST: break;
        case OP_ADD: da = g; db = g; break;
        case OP_SUB: da = g; db = T(0.0f) - g; break;
        case OP_MUL: da = g * vb; db = g * va; break;
        case OP_DIV: { T inv = vrcp(vb); da = g * inv; db = T(0.0f) - g * vo * inv; } break;
        case OP_NEG: da = T(0.0f) - g; break;
        case OP_ADDC: da = g; break;
        case OP_MULC: da = g * T(imm); break;
        case OP_POWI: { int n = (int)imm; da = (n == 0) ? T(0.0f) : g * T((float)n) * powi(va, n - 1); } break;
        case OP_POW: da = g * vb * vpow(va, vb - T(1.0f)); db = g * vo * vlog(va); break;
        case OP_POWC: da = g * T(imm) * vpow(va, T(imm - I(1))); break;
        case OP_SIN: da = g * vcos(va); break;
        case OP_COS: da = T(0.0f) - g * vsin(va); break;
        case OP_TAN: da = g * (T(1.0f) + vo * vo); break;
        case OP_EXP: da = g * vo; break;
        case OP_LOG: da = g * vrcp(va); break;
        case OP_SQRT: da = g * T(0.5f) * vrcp(vo); break;
        case OP_ABS: da = g * vsign(va); break;
        case OP_TANH: da = g * (T(1.0f) - vo * vo); break;
        case OP_SINH: da = g * vcosh(va); break;
        case OP_COSH: da = g * vsinh(va); break;
        case OP_SECH: da = T(0.0f) - g * vo * vtanh(va); break;
        case OP_SINPI: da = g * PI * vcospi(va); break;
        case OP_COSPI: da = T(0.0f) - g * PI * vsinpi(va); break;
        case OP_MAX: { auto m = vgt(va, vb); da = vselect(m, g, T(0.0f)); db = vselect(m, T(0.0f), g); } break;
        case OP_MIN: { auto m = vgt(vb, va); da = vselect(m, g, T(0.0f)); db = vselect(m, T(0.0f), g); } break;
        default: break;
    }
}

HD bool is_binary(int code) {
    return code == OP_ADD || code == OP_SUB || code == OP_MUL || code == OP_DIV || code == OP_POW ||
           code == OP_MAX || code == OP_MIN;
}
HD bool is_nullary(int code) { return code == OP_CONST || code == OP_DATA; }
HD bool is_bilinear(int code) { return code <= OP_MULC && code != OP_DIV; }
// fill the pre-decoded coefficients (host side, once per program)
inline void finalize(Instr& I) {
    I.k0 = I.k1 = I.k2 = I.k3 = 0.f;
    switch (I.code) {
        case OP_CONST: I.k0 = I.imm; break;
        case OP_ADD: I.k1 = 1.f; I.k2 = 1.f; break;
        case OP_SUB: I.k1 = 1.f; I.k2 = -1.f; break;
        case OP_MUL: I.k3 = 1.f; break;
        case OP_NEG: I.k1 = -1.f; break;
        case OP_ADDC: I.k1 = 1.f; I.k0 = I.imm; break;
        case OP_MULC: I.k1 = I.imm; break;
        default: break;
    }
}
template <class T>
DEV T apply_bilinear(const Instr& I, T va, T vb) { return vfma(va, vfma(vb, T(I.k3), T(I.k1)), vfma(vb, T(I.k2), T(I.k0))); }
template <class T>
DEV void adjoint_bilinear(const Instr& I, T va, T vb, T g, T& da, T& db) {
    da = g * vfma(vb, T(I.k3), T(I.k1));
    db = g * vfma(va, T(I.k3), T(I.k2));
}

}  // namespace rp
