// pinn_kernels5.hpp — "family 4m": the FLOAT64 evaluation on the matrix pipe (v_mfma_f64_16x16x4_f64), r05.
//
// Family 4 (pinn_kernels4.hpp) evaluates one POINT per lane and pays for every multiply-add with scratch-row loads; the reference's default
// eltype (Float64, src/discretize.jl:432-449) deserves the matrix cores.  Same mathematics, same scratch rows, same slab layout and the same
// small-entry / reduction kernels as family 4 — only the two heavy kernels are replaced:
//   * k_f64m_tile: one WAVE per tile of 16 * PG points.  The f64 MFMA's C/D layout (lane (q, j) = (lane >> 4, lane & 15) holds rows q + 4 r,
//     column j, cdna_hip_programming.md "f64 MFMA does NOT use these maps") is SELF-FEEDING with neuron = row, point = column: D register r of
//     output tile t is exactly the B operand (k = lane >> 4, j = lane & 15) of k-block 4 t + r of the next GEMM — activations, jets and dZ stay
//     in the lane that produced them from the first layer to the last and back: no LDS exchange, no barriers, wave-private tiles.  A operands
//     (W for the forward GEMM, W^T for dA) come straight from theta (double, ComponentArrays order; L2-resident), masked to the layer's
//     true width.  All C jet channels of PG point groups travel as NCG = PG * C column groups of 16 columns that share every weight fragment.
//     Per element the activation / jet rules are the templates of pinn_kernels.hpp with V = double, the tape is family 4's.
//   * k_f64m_dwt: the hidden-to-hidden weight gradients dW = dZ A^T (and those layers' bias gradients) as MFMAs over the scratch rows
//     (k = 16 consecutive points of one channel), one workgroup of four waves per (512-point block, layer), each wave a quarter of the block's
//     points and every output x input tile of the layer in registers, combined through LDS in wave order, written into the block's slab.
//     One more workgroup row of the same launch adds the tile kernel's per-tile sums (below) into the slab.
// Records / post-activation jets / dZ of the hidden-to-hidden GEMMs go through scratch rows numbered as family 4's, laid out point-block-major
// (f64m_six).  What family 4's k_f64_dw reads back from rows — first and last layer, output bias, PDE parameters, the sum of squares — the tile
// kernel sums over its own points while the operands are in registers (F64Args::tpart: one row of partial sums per tile): the first layer's dZ
// rows, the last hidden layer's activation rows and the seed / parameter / residual rows are never written.
// Eligibility (f64.cpp): tanh / sigmoid networks with at least one hidden layer, hidden widths <= 16 * HT of an instantiated (jet set, HT)
// pair; everything else keeps family 4.  PINN_F64_NO_MFMA=1 keeps family 4 everywhere (A/B, tests).
#pragma once
#include "pinn_kernels4.hpp"
#ifndef PINN_F64M_DWT_SINGLE
#define PINN_F64M_DWT_SINGLE 1          // the dW kernel with ONE operand buffer: 256 registers, two workgroups per CU — 2.93 against 3.92 ms for the bench
                                        // workload's float64 evaluation with the second buffer (334 registers, one workgroup per CU); 0 restores it (A/B)
#endif
#ifndef PINN_F64M_DEFER
#define PINN_F64M_DEFER 1               // a layer's scratch stores ride inside the NEXT GEMM's MFMA stream (f64m_tile); 0: issued by the activation loop (A/B)
#endif
#ifndef PINN_F64M_CAP
#define PINN_F64M_CAP 2                 // column groups per wave of the small-jet-set kernels (value-only terms: PG = CAP point groups per tile); A/B
#endif
#ifndef PINN_F64M_WAVES_SMALL
#define PINN_F64M_WAVES_SMALL 2         // waves per SIMD the kernels with HT x PG x C <= 8 are compiled for (A/B: 4 with CAP = 1)
#endif
#ifndef PINN_F64M_PROBE
#define PINN_F64M_PROBE 0               // timing probes (tools only, wrong numbers): 1 no rolling reload of the weight fragments, 2 no scratch stores, 4 no activation function, 8 no lane predicates on the element loops (full-width nets and full tiles only)
#endif

namespace pk {

// ---- lane arrays: N doubles per lane.  Device: registers of the executing lane; emulation: [N][64], phases written as PINN_LANES(l) { ... } ----
#ifdef PINN_EMU
template <int N> struct LVd {
    double v[N][64];
    double& operator()(int l, int i) { return v[i][l]; }
    const double& operator()(int l, int i) const { return v[i][l]; }
};
#define PINN_LANES(l) for (int l = 0; l < 64; ++l)
// D[i][j] += sum_k A[i][k] B[k][j]: lane (k, i) supplies A[i][k], lane (k, j) supplies B[k][j], lane (q, j) receives rows q + 4 r (r = 0..3) in
// registers c0 + r * cs
template <int NC, int NA, int NB>
inline void mfma_f64(LVd<NC>& C, int c0, int cs, const LVd<NA>& A, int ia, const LVd<NB>& B, int ib) {
    double out[4][64];
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int i = (l >> 4) + 4 * r, j = l & 15;
            double acc = C(l, c0 + r * cs);
            for (int k = 0; k < 4; ++k) acc = std::fma(A(16 * k + i, ia), B(16 * k + j, ib), acc);
            out[r][l] = acc;
        }
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) C(l, c0 + r * cs) = out[r][l];
}
// sum over the four lanes that share lane & 15, result in all four
template <int N> inline void lv_qsum(LVd<N>& X, int i) {
    for (int j = 0; j < 16; ++j) {
        const double s = (X(j, i) + X(j + 16, i)) + (X(j + 32, i) + X(j + 48, i));
        for (int q = 0; q < 4; ++q) X(j + 16 * q, i) = s;
    }
}
// sum over the sixteen lanes that share lane >> 4 (the 16 points of a tile row), in every one of them: the device's four DPP row rotations
template <int N> inline void lv_rowsum(LVd<N>& X, int i) {
    for (int q = 0; q < 4; ++q) {
        double t[16], u[16];
        for (int j = 0; j < 16; ++j) t[j] = X(16 * q + j, i);
        for (int o = 8; o >= 1; o >>= 1) {
            for (int j = 0; j < 16; ++j) u[j] = t[j] + t[(j - o) & 15];
            for (int j = 0; j < 16; ++j) t[j] = u[j];
        }
        for (int j = 0; j < 16; ++j) X(16 * q + j, i) = t[j];
    }
}
#else
template <int N> struct LVd {
    double v[N];
    DEV double& operator()(int, int i) { return v[i]; }
    DEV const double& operator()(int, int i) const { return v[i]; }
};
#define PINN_LANES(l) for (int l = (int)(threadIdx.x & 63), once_##l = 1; once_##l; once_##l = 0)
template <int NC, int NA, int NB>
DEV void mfma_f64(LVd<NC>& C, int c0, int cs, const LVd<NA>& A, int ia, const LVd<NB>& B, int ib) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 c = {C.v[c0], C.v[c0 + cs], C.v[c0 + 2 * cs], C.v[c0 + 3 * cs]};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A.v[ia], B.v[ib], c, 0, 0, 0);
    C.v[c0] = c[0]; C.v[c0 + cs] = c[1]; C.v[c0 + 2 * cs] = c[2]; C.v[c0 + 3 * cs] = c[3];
}
template <int N> DEV void lv_qsum(LVd<N>& X, int i) {
    const double x = X.v[i];
    const double y = x + __shfl_xor(x, 16, 64);          // (q ^ 1)
    X.v[i] = y + __shfl_xor(y, 32, 64);                  // same association as the emulation: (x0 + x1) + (x2 + x3)
}
template <int CTRL> DEV double dpp_mov_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int N> DEV void lv_rowsum(LVd<N>& X, int i) {  // row_ror 8, 4, 2, 1 (vec.hpp: row_allsum16): VALU only, no LDS crossbar
    double v = X.v[i];
    v += dpp_mov_d<0x128>(v); v += dpp_mov_d<0x124>(v); v += dpp_mov_d<0x122>(v); v += dpp_mov_d<0x121>(v);
    X.v[i] = v;
}
#endif

// scratch element (row, point) of the matrix-pipe kernels: POINT-BLOCK-MAJOR [point / 16][row][point % 16] — a wave's tile (16 points x every
// row) is one contiguous region (rows x 128 bytes), so its ~1,500 stores per layer land in a few DRAM pages instead of one 128-byte line in
// each of 1,500 rows that lie npad x 8 bytes apart (family 4's row-major layout: 1.6 of 4.1 ms of the bench workload's float64 evaluation were
// the tile kernels' stores, profiles/r05_f64_kernel_stats.txt), and the dW kernel's 16-point operand lines of one channel are neighbours
HD size_t f64m_six(const F64Args& a, size_t row, int p) { return (((size_t)(p >> 4) * (size_t)a.nrows + row) << 4) + (size_t)(p & 15); }
// four consecutive doubles at a 32-byte aligned address (device: two global_load_dwordx4)
#ifdef PINN_EMU
inline void ld4_f64(const double* p, double (&o)[4]) { for (int i = 0; i < 4; ++i) o[i] = p[i]; }
#else
DEV void ld4_f64(const double* p, double (&o)[4]) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 a = *reinterpret_cast<const d2*>(p), b = *reinterpret_cast<const d2*>(p + 2);
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
}
#endif

// ---- kernel A': forward jets of every network, residual tape, reverse sweep of one tile of 16 * PG points ----
template <class J, int HT, int PG, int ACTK>
DEV void f64m_tile(int tile, const F64Args& a) {
    constexpr int C = J::C, NCG = PG * C, NR = HT * 4;
    constexpr bool SIN = (ACTK == ACT_SIN);
    const int pbase = tile * (16 * PG);
    // this tile's term (F64Sub): its own point set, weights, data rows, tape, slots and seed factor; T_p0 + (launch point index) = index into its set
    int sidx = 0;
    for (int s_ = 1; s_ < a.nsub; ++s_) sidx += tile >= a.sub_tile0[s_] ? 1 : 0;
    const F64Sub& T = a.sub[sidx];
    const int T_p0 = a.nsub > 0 ? -a.sub_tile0[sidx] * (16 * PG) : a.p0;
    const int T_npts = a.nsub > 0 ? a.sub_tile0[sidx] * (16 * PG) + T.N : a.npts;          // launch point indices below this one are points of the term
    double* S = a.scratch;
    LVd<NR * NCG> X, Z;                                          // X: operand of the next GEMM (a jets / dZ); Z: its result (z jets / G)
    LVd<F64_MAX_NETS * NCG> U;                                   // every network's output jets (forward), then their seeds (reverse)
    // value-only terms of ONE tanh / sigmoid network: the last hidden layer's activations are still in X when the reverse sweep starts (the output
    // layer and the tape read X / U only) and the record of such an element is its activation: that layer's rows never go to memory
    const bool keep_last = a.nnets == 1 && C == 1 && a.post_alias != 0;
    double* TP = a.tpart + (size_t)tile * (size_t)a.ntp;         // this tile's row of partial sums (F64Args::tpart)
    // DEFERRED STORES (PINN_F64M_DEFER).  vmcnt counts loads and stores alike and retires them in order: a burst of ~100 row stores behind an
    // activation loop stands between the wave and the first weight fragment of the next GEMM, with one wave (or two) per SIMD nothing else to
    // run meanwhile (timing probe without stores: -0.5 of 2.4 ms, profiles/r05_f64_kernel_stats.txt section 6).  The values stay where they are
    // — activations / dZ in X (the next GEMM's B operand), the record's derivative channels in Z until the next GEMM's tile row overwrites
    // them — and go out a row at a time between that GEMM's MFMAs.
    constexpr bool DEFER = PINN_F64M_DEFER != 0;
    static_assert(!DEFER || !SIN, "deferred stores take the record's value channel from X: tanh / sigmoid kernels only");
    const bool stores_on = a.mode == 0 && !(PINN_F64M_PROBE & 2);
    // row tr of X (neurons 4 tr + q of a layer `width` wide) -> scratch rows base + neuron * C + c, channels [c0, C)
    auto store_x_row = [&](int tr, int base, int width, int c0) {
        PINN_LANES(l) {
            const int m = 4 * tr + (l >> 4);
            if (m < width) {
                PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                    const int p = pbase + 16 * pg + (l & 15);
                    PINN_UNROLL for (int c = 0; c < C; ++c) if (c >= c0) S[f64m_six(a, (size_t)base + (size_t)m * C + c, p)] = X(l, tr * NCG + pg * C + c);
                }
            }
        }
    };
    // the derivative channels (c >= 1) of row tr of Z -> the record rows
    auto store_z_row = [&](int tr, int base, int width) {
        PINN_LANES(l) {
            const int m = 4 * tr + (l >> 4);
            if (m < width) {
                PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                    const int p = pbase + 16 * pg + (l & 15);
                    PINN_UNROLL for (int c = 1; c < C; ++c) S[f64m_six(a, (size_t)base + (size_t)m * C + c, p)] = Z(l, tr * NCG + pg * C + c);
                }
            }
        }
    };
    // =========================== forward ===========================
    for (int ni = 0; ni < a.nnets; ++ni) {
        const F64Net& n = a.net[ni];
        const int L = n.nl - 1;
        for (int lyr = 0; lyr < L; ++lyr) {
            const int n_in = n.sizes[lyr], n_out = n.sizes[lyr + 1];
            const double* W = a.theta + n.woff[lyr];
            const double* B = a.theta + n.boff[lyr];
            if (lyr == 0) {
                PINN_LANES(l) {
                    const int q = l >> 4, j = l & 15;
                    PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                        const int p = pbase + 16 * pg + j, pc = p < T_npts ? p : T_npts - 1;
                        double x[4] = {0.0, 0.0, 0.0, 0.0};
                        for (int i = 0; i < n.d; ++i) x[i] = T.pts[(size_t)(T_p0 + pc) * a.dt + n.imap[i]];
                        PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                            const int m = 16 * (tr >> 2) + 4 * (tr & 3) + q, mc = m < n_out ? m : n_out - 1;
                            const bool valid = m < n_out;
                            double z0 = B[mc];
                            for (int i = 0; i < n.d; ++i) z0 = vfma(W[mc + (size_t)i * n_out], x[i], z0);
                            PINN_UNROLL for (int c = 0; c < C; ++c) Z(l, tr * NCG + pg * C + c) = 0.0;
                            Z(l, tr * NCG + pg * C) = valid ? z0 : 0.0;
                            PINN_UNROLL for (int kf = 0; kf < J::NFIRST; ++kf)
                                Z(l, tr * NCG + pg * C + J::CH_FIRST + kf) = valid ? W[mc + (size_t)J::first_axis(kf) * n_out] : 0.0;
                        }
                    }
                }
            } else {
                if (!DEFER) { PINN_LANES(l) { PINN_UNROLL for (int e = 0; e < NR * NCG; ++e) Z(l, e) = 0.0; } }
                // the layer's biases first: requested in front of the GEMM (behind it — after the previous layer's scratch stores, which the
                // compiler must assume to alias theta — their L2 round trip would sit between the last MFMA and the activation)
                // (one-wave-per-SIMD kernels only: the two-wave kernels have no 32 registers to spare — measured 207 -> 244 us with the prefetch)
                constexpr bool BPRE = (HT * NCG > 8);
                LVd<BPRE ? NR : 1> Bv;
                if (BPRE) {
                    PINN_LANES(l) {
                        PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                            const int m = 16 * (tr >> 2) + 4 * (tr & 3) + (l >> 4);
                            Bv(l, BPRE ? tr : 0) = B[m < n_out ? m : n_out - 1];
                        }
                    }
                }
                // A fragments (W rows of output tile t, k-block kb) with a ROLLING prefetch: fragment (t + 1, kb) is requested right behind the
                // MFMAs that consumed fragment (t, kb), so an L2 round trip has a whole tile row of MFMAs (16 NCG x 64 cycles) to land.
                // r06: the loads are UNCONDITIONAL (a predicate per element made every load its own exec-masked branch in the MFMA stream,
                // profiles/r06_f64_kernel_stats.txt): what a narrow layer reads past its matrix (theta carries a zeroed margin, f64.cpp) meets
                // rows of X that are zero (k >= n_in) or lands in rows of Z that the element loops mask (m >= n_out)
                LVd<NR> Af;
                PINN_LANES(l) {
                    PINN_UNROLL for (int kb = 0; kb < NR; ++kb) {
                        const int m = (l & 15), k = 4 * kb + (l >> 4);
                        Af(l, kb) = W[m + k * n_out];
                    }
                }
                PINN_UNROLL for (int t = 0; t < HT; ++t) {
                    if (DEFER) {
                        // tile row t of Z still holds the PREVIOUS layer's pre-activation jets: their derivative channels are that layer's record
                        if (C > 1 && stores_on) { PINN_UNROLL for (int r = 0; r < 4; ++r) store_z_row(4 * t + r, n.r_rec[lyr - 1], n_in); }
                        PINN_LANES(l) { PINN_UNROLL for (int e = 0; e < 4 * NCG; ++e) Z(l, 4 * t * NCG + e) = 0.0; }
                    }
                    if (16 * t < n_out) {
                        PINN_UNROLL for (int kb = 0; kb < NR; ++kb) {
                            if (4 * kb >= n_in) break;
                            PINN_UNROLL for (int g = 0; g < NCG; ++g) mfma_f64(Z, (4 * t) * NCG + g, NCG, Af, kb, X, kb * NCG + g);
                            if (t + 1 < HT && !(PINN_F64M_PROBE & 1)) {
                                PINN_LANES(l) {
                                    const int m = 16 * (t + 1) + (l & 15), k = 4 * kb + (l >> 4);
                                    Af(l, kb) = W[m + k * n_out];
                                }
                            }
                            // the previous layer's activations (this GEMM's B operand, row kb of X) go out behind the MFMAs that read them first:
                            // value channel = the record's value channel (tanh / sigmoid), every channel = the post-activation jets the dW kernel reads
                            if (DEFER && t == 0 && stores_on) {
                                if (a.post_alias) store_x_row(kb, n.r_rec[lyr - 1], n_in, 0);
                                else {
                                    PINN_LANES(l) {
                                        const int m = 4 * kb + (l >> 4);
                                        if (m < n_in) {
                                            PINN_UNROLL for (int pg = 0; pg < PG; ++pg)
                                                S[f64m_six(a, (size_t)n.r_rec[lyr - 1] + (size_t)m * C, pbase + 16 * pg + (l & 15))] = X(l, kb * NCG + pg * C);
                                        }
                                    }
                                    store_x_row(kb, n.r_post[lyr - 1], n_in, 0);
                                }
                            }
                        }
                    }
                }
                PINN_LANES(l) {
                    const int q = l >> 4;
                    PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                        const int m = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                        const double b = m < n_out ? (BPRE ? Bv(l, BPRE ? tr : 0) : B[m]) : 0.0;
                        PINN_UNROLL for (int pg = 0; pg < PG; ++pg) Z(l, tr * NCG + pg * C) += b;
                    }
                }
            }
            // activation: record, jets (rows as family 4 writes them: rec / post of hidden layer lyr, row = base + neuron * C + channel)
            PINN_LANES(l) {
                const int q = l >> 4, j = l & 15;
                PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                    const int m = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                    const bool valid = (PINN_F64M_PROBE & 8) ? true : m < n_out;
                    PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                        const int p = pbase + 16 * pg + j;
                        const bool st = (PINN_F64M_PROBE & 8) ? true : valid && a.mode == 0 && !(PINN_F64M_PROBE & 2);      // (points past the chunk's end: into the rows' padding — npad is a multiple of the 512-point block, every reader masks by the point count)
                        double z[C];
                        PINN_UNROLL for (int c = 0; c < C; ++c) z[c] = Z(l, tr * NCG + pg * C + c);
                        const double a0 = (PINN_F64M_PROBE & 4) ? z[0] * 0.5 : act_value<SIN>(f64_act(n, lyr), z[0]);
                        z[0] = act_record<SIN>(z[0], a0);
                        // the LAST hidden layer's rows have one reader left, this wave's reverse sweep (its post-activation jets fed the output
                        // weights' gradient, which the reverse sweep now forms itself): no post rows, and no record either where X keeps it
                        // (the other layers' rows: deferred into the next GEMM, above)
                        if (st && (!DEFER || lyr == L - 1) && !(keep_last && lyr == L - 1)) { PINN_UNROLL for (int c = 0; c < C; ++c) S[f64m_six(a, (size_t)n.r_rec[lyr] + (size_t)m * C + c, p)] = z[c]; }
                        double dd[ND];
                        act_derivs_n<J::NORD - 1, SIN>(f64_act(n, lyr), z[0], dd);
                        jet_forward<J>(z, dd);
                        z[0] = a0;
                        if (st && !DEFER && !a.post_alias && lyr != L - 1) { PINN_UNROLL for (int c = 0; c < C; ++c) S[f64m_six(a, (size_t)n.r_post[lyr] + (size_t)m * C + c, p)] = z[c]; }
                        PINN_UNROLL for (int c = 0; c < C; ++c) X(l, tr * NCG + pg * C + c) = valid ? z[c] : 0.0;
                    }
                }
            }
        }
        // output layer: u[g] = sum_n w_n a_n[g] (+ b on the value channel): per-lane partial over its neurons, then over the four lane groups
        {
            const int n_in = n.sizes[L];
            const double* W = a.theta + n.woff[L];
            const double bo = a.theta[n.boff[L]];
            LVd<NCG> u;
            PINN_LANES(l) {
                const int q = l >> 4;
                PINN_UNROLL for (int g = 0; g < NCG; ++g) u(l, g) = 0.0;
                PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                    const int m = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                    const double w = m < n_in ? W[m] : 0.0;
                    PINN_UNROLL for (int g = 0; g < NCG; ++g) u(l, g) = vfma(w, X(l, tr * NCG + g), u(l, g));
                }
            }
            PINN_UNROLL for (int g = 0; g < NCG; ++g) lv_qsum(u, g);
            PINN_LANES(l) {
                PINN_UNROLL for (int g = 0; g < NCG; ++g) U(l, ni * NCG + g) = u(l, g) + ((g % C) == 0 ? bo : 0.0);
            }
        }
    }
    // =========================== residual tape (per point; the four lanes of a point run it redundantly, lane group 0 writes) ===========================
    LVd<1 + MAX_PARAMS> TS;                                      // lane group 0, lane j: squared weighted residual / PDE-parameter partials of its points
    PINN_LANES(l) { PINN_UNROLL for (int e = 0; e < 1 + MAX_PARAMS; ++e) TS(l, e) = 0.0; }
    PINN_LANES(l) {
        const int q = l >> 4, j = l & 15;
        const int R0 = a.dt + a.np + T.nslots;
        PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
            const int p = pbase + 16 * pg + j, pc = p < T_npts ? p : T_npts - 1, gp = T_p0 + pc;
            const bool live = p < T_npts, wr = live && q == 0;
            // AFFINE residuals (r06; F64Sub::lin, f64.cpp: f64_affine): r = S(point) + sum_s a_s u_s with constant a_s and a coordinate-only part S that the
            // host evaluated once per point set — no interpreter, no private v[] / g[] arrays (their scratch-memory round trips are a third of a small
            // problem's tile latency); everything else runs the tape
            const double* LIN = T.lin;
            double v[F64_MAX_ROWS], g[F64_MAX_ROWS];
            double r;
            if (LIN) {
                r = LIN[T.nslots + gp];
                for (int s = 0; s < T.nslots; ++s) {
                    double uu = 0.0;
                    for (int ni = 0; ni < a.nnets; ++ni)
                        PINN_UNROLL for (int c = 0; c < C; ++c) if (T.slot_net[s] == ni && T.slot_chan[s] == c) uu = U(l, ni * NCG + pg * C + c);
                    r = vfma(LIN[s], uu, r);
                }
            } else {
                for (int i = 0; i < a.dt; ++i) v[i] = T.pts[(size_t)gp * a.dt + i];
                for (int k = 0; k < a.np; ++k) v[a.dt + k] = k < a.ne ? a.theta[a.p_off + k] : a.pdef[k];
                for (int s = 0; s < T.nslots; ++s) {
                    double uu = 0.0;
                    for (int ni = 0; ni < a.nnets; ++ni)
                        PINN_UNROLL for (int c = 0; c < C; ++c) if (T.slot_net[s] == ni && T.slot_chan[s] == c) uu = U(l, ni * NCG + pg * C + c);
                    v[a.dt + a.np + s] = uu;
                }
                for (int o = 0; o < T.nops; ++o) {
                    const rp::Instr ins = T.prog[o];
                    const double va = rp::is_nullary(ins.code) ? 0.0 : v[ins.a];
                    const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
                    v[R0 + o] = (ins.code == rp::OP_DATA) ? T.data[(size_t)(int)T.imm[o] * (size_t)T.N + (size_t)gp] : rp::apply<double, double>(ins.code, va, vb, T.imm[o]);
                }
                r = v[T.out_row];
            }
            if (a.mode == 2) { if (wr) a.resid[gp] = r; continue; }
            const double sw = T.pw ? (double)T.pw[gp] : 1.0;
            const double rs = r * sw;
            if (wr) TS(l, 0) += rs * rs;
            if (a.mode == 1) continue;
            if (!LIN) {
                for (int o = 0; o < R0 + T.nops; ++o) g[o] = 0.0;
                g[T.out_row] = 1.0;
                for (int o = T.nops - 1; o >= 0; --o) {
                    const rp::Instr ins = T.prog[o];
                    if (rp::is_nullary(ins.code)) continue;
                    const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
                    double da, db;
                    rp::adjoint<double, double>(ins.code, v[ins.a], vb, v[R0 + o], T.imm[o], g[R0 + o], da, db);
                    g[ins.a] += da;
                    if (rp::is_binary(ins.code)) g[ins.b] += db;
                }
            }
            const double rbar = live ? rs * T.scale * sw : 0.0;
            PINN_UNROLL for (int k = 0; k < MAX_PARAMS; ++k) if (wr && k < a.ne && !LIN) TS(l, 1 + k) += rbar * g[a.dt + k];      // (an affine residual reads no estimated parameter)
            // the seeds of every network's output jets replace its outputs in U
            for (int ni = 0; ni < a.nnets; ++ni) {
                double ub[C];
                PINN_UNROLL for (int c = 0; c < C; ++c) ub[c] = 0.0;
                for (int s = 0; s < T.nslots; ++s) {
                    if (T.slot_net[s] != ni) continue;
                    const double gs = rbar * (LIN ? LIN[s] : g[a.dt + a.np + s]);
                    PINN_UNROLL for (int c = 0; c < C; ++c) if (T.slot_chan[s] == c) ub[c] += gs;
                }
                PINN_UNROLL for (int c = 0; c < C; ++c) U(l, ni * NCG + pg * C + c) = ub[c];
            }
        }
    }
    // the tile's sums of the per-point scalars: sum of squares, PDE-parameter partials, every network's output-bias gradient (= sum of its value seeds)
    if (a.mode != 2) {
        PINN_UNROLL for (int e = 0; e < 1 + MAX_PARAMS; ++e) { if (e > a.ne) break; lv_rowsum(TS, e); }
        PINN_LANES(l) {
            if (l == 0) {
                TP[a.tp_p + a.ne] = TS(l, 0);
                PINN_UNROLL for (int k = 0; k < MAX_PARAMS; ++k) if (k < a.ne && a.mode == 0) TP[a.tp_p + k] = TS(l, 1 + k);
            }
        }
    }
    if (a.mode != 0) return;
    for (int ni = 0; ni < a.nnets; ++ni) {
        const F64Net& n = a.net[ni];
        LVd<1> bl;
        PINN_LANES(l) {
            double t = 0.0;
            PINN_UNROLL for (int pg = 0; pg < PG; ++pg) t += U(l, ni * NCG + pg * C);
            bl(l, 0) = (l >> 4) == 0 ? t : 0.0;                   // (the four lane groups of a point hold the same seeds)
        }
        lv_rowsum(bl, 0);
        PINN_LANES(l) { if (l == 0) TP[n.tp0 + (n.d + 1) * n.sizes[1] + n.sizes[n.nl - 1]] = bl(l, 0); }
    }
    // =========================== reverse sweep, network by network ===========================
    for (int ni = 0; ni < a.nnets; ++ni) {
        const F64Net& n = a.net[ni];
        const int L = n.nl - 1;
        for (int lyr = L - 1; lyr >= 0; --lyr) {
            const int H = n.sizes[lyr + 1];
            const int n_next = n.sizes[lyr + 2];
            const double* Wn = a.theta + n.woff[lyr + 1];            // W_{lyr+1}[m + k * n_next]
            if (lyr == L - 1) {
                PINN_LANES(l) {
                    const int q = l >> 4;
                    PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                        const int k = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                        const double w = k < H ? Wn[k] : 0.0;
                        PINN_UNROLL for (int g = 0; g < NCG; ++g) Z(l, tr * NCG + g) = w * U(l, ni * NCG + g);
                    }
                }
            } else {
                PINN_LANES(l) { PINN_UNROLL for (int e = 0; e < NR * NCG; ++e) Z(l, e) = 0.0; }
                LVd<NR> Af;                                          // W^T: rows = this layer's neurons, k = the layer above's; rolling prefetch as above
                PINN_LANES(l) {
                    PINN_UNROLL for (int kb = 0; kb < NR; ++kb) {
                        const int k = (l & 15), m = 4 * kb + (l >> 4);
                        Af(l, kb) = Wn[m + k * n_next];
                    }
                }
                PINN_UNROLL for (int t = 0; t < HT; ++t) {
                    if (16 * t >= H) break;
                    PINN_UNROLL for (int kb = 0; kb < NR; ++kb) {
                        if (4 * kb >= n_next) break;
                        PINN_UNROLL for (int g = 0; g < NCG; ++g) mfma_f64(Z, (4 * t) * NCG + g, NCG, Af, kb, X, kb * NCG + g);
                        if (t + 1 < HT && !(PINN_F64M_PROBE & 1)) {
                            PINN_LANES(l) {
                                const int k = 16 * (t + 1) + (l & 15), m = 4 * kb + (l >> 4);
                                Af(l, kb) = Wn[m + k * n_next];
                            }
                        }
                        // dZ of the layer above (this GEMM's B operand, row kb of X): its rows for the dW kernel, behind the MFMAs that read it first
                        if (DEFER && t == 0 && stores_on) store_x_row(kb, n.r_dz[lyr + 1], n_next, 0);
                    }
                }
            }
            // this layer's records into X (dead after the GEMM above): every load in flight before the first use — behind the stores of the
            // adjoint loop below the compiler could not hoist them (same scratch pointer), and each would expose a memory round trip
            if (!(keep_last && lyr == L - 1)) {
                PINN_LANES(l) {
                    const int q = l >> 4, j = l & 15;
                    // (unclamped, affine addresses: rows beyond the layer's width / points beyond the chunk's end read the scratch margin or the rows'
                    // padding — whatever comes back is discarded by the selects of the adjoint loop below)
                    PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                        const int k = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                        PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                            const int p = pbase + 16 * pg + j;
                            PINN_UNROLL for (int c = 0; c < C; ++c) X(l, tr * NCG + pg * C + c) = S[f64m_six(a, (size_t)n.r_rec[lyr] + (size_t)k * C + c, p)];
                        }
                    }
                }
            }
            // first hidden layer: the network's inputs at this lane's points, for dW_0 = sum_p dZ_0 x^T
            LVd<PG * 4> XI;
            if (lyr == 0) {
                PINN_LANES(l) {
                    PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                        const int p = pbase + 16 * pg + (l & 15), pc = p < T_npts ? p : T_npts - 1;
                        PINN_UNROLL for (int i = 0; i < 4; ++i) XI(l, pg * 4 + i) = i < n.d ? T.pts[(size_t)(T_p0 + pc) * a.dt + n.imap[i]] : 0.0;
                    }
                }
            }
            // one tile row (four neurons x 16 points x PG) at a time: adjoint of the activation, then — first / last hidden layer — the row's
            // sums over the tile's points of what the input / output weights' gradients need (the rows never leave the wave: no dZ_0 rows, no
            // re-read of the last layer's activations by another kernel)
            PINN_UNROLL for (int tr = 0; tr < NR; ++tr) {
                LVd<6> T6;                                           // [0, d) dW_0[m][i], [4] db_0[m], [5] dW_L[m]
                PINN_LANES(l) {
                    const int q = l >> 4, j = l & 15;
                    const int k = 16 * (tr >> 2) + 4 * (tr & 3) + q;
                    const bool valid = (PINN_F64M_PROBE & 8) ? true : k < H;
                    double t6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                    PINN_UNROLL for (int pg = 0; pg < PG; ++pg) {
                        const int p = pbase + 16 * pg + j;
                        const bool st = (PINN_F64M_PROBE & 8) ? true : valid && p < T_npts;
                        const bool st2 = !DEFER && ((PINN_F64M_PROBE & 8) ? lyr != 0 : valid && lyr != 0 && !(PINN_F64M_PROBE & 2));   // (deferred: out of X inside the next dA GEMM)
                        double s[C], gq[C], dd[ND];
                        PINN_UNROLL for (int c = 0; c < C; ++c) s[c] = X(l, tr * NCG + pg * C + c);
                        PINN_UNROLL for (int c = 0; c < C; ++c) gq[c] = Z(l, tr * NCG + pg * C + c);
                        act_derivs_n<J::NORD, SIN>(f64_act(n, lyr), s[0], dd);
                        if (lyr == L - 1) {                          // dW_L[k] += sum_c ubar_c * (post-activation jet c of neuron k): the forward rule on the record
                            double pz[C];
                            PINN_UNROLL for (int c = 0; c < C; ++c) pz[c] = s[c];
                            jet_forward<J>(pz, dd);
                            pz[0] = SIN ? act_value<SIN>(f64_act(n, lyr), s[0]) : s[0];
                            double t2 = 0.0;
                            PINN_UNROLL for (int c = 0; c < C; ++c) t2 = vfma(U(l, ni * NCG + pg * C + c), pz[c], t2);
                            t6[5] += st ? t2 : 0.0;
                        }
                        jet_adjoint<J>(gq, s, dd);
                        if (st2) { PINN_UNROLL for (int c = 0; c < C; ++c) S[f64m_six(a, (size_t)n.r_dz[lyr] + (size_t)k * C + c, p)] = gq[c]; }
                        PINN_UNROLL for (int c = 0; c < C; ++c) X(l, tr * NCG + pg * C + c) = st ? gq[c] : 0.0;
                        if (lyr == 0) {
                            const double z0 = st ? gq[0] : 0.0;
                            t6[4] += z0;
                            PINN_UNROLL for (int i = 0; i < 4; ++i) {
                                if (i >= n.d) break;
                                double t2 = z0 * XI(l, pg * 4 + i);
                                const int ck = a.first_ch[i];
                                PINN_UNROLL for (int c = 1; c < C; ++c) if (c == ck) t2 += st ? gq[c] : 0.0;
                                t6[i] += t2;
                            }
                        }
                    }
                    PINN_UNROLL for (int i = 0; i < 6; ++i) T6(l, i) = t6[i];
                }
                if (lyr == 0) {
                    PINN_UNROLL for (int i = 0; i < 4; ++i) { if (i >= n.d) break; lv_rowsum(T6, i); }
                    lv_rowsum(T6, 4);
                }
                if (lyr == L - 1) lv_rowsum(T6, 5);
                PINN_LANES(l) {
                    const int k = 16 * (tr >> 2) + 4 * (tr & 3) + (l >> 4);
                    if ((l & 15) == 0 && k < H) {
                        const int n1 = n.sizes[1];
                        if (lyr == 0) {
                            PINN_UNROLL for (int i = 0; i < 4; ++i) if (i < n.d) TP[n.tp0 + i * n1 + k] = T6(l, i);
                            TP[n.tp0 + n.d * n1 + k] = T6(l, 4);
                        }
                        if (lyr == L - 1) TP[n.tp0 + (n.d + 1) * n1 + k] = T6(l, 5);
                    }
                }
            }
        }
    }
}

// ---- kernel B2': the hidden-to-hidden weight gradients dW = dZ A^T AND the bias gradients of those layers for one 512-point block: one
// workgroup of four waves per (layer, block).  Wave w takes the 16-point steps w, w + 4, ... of the block and accumulates EVERY output tile x input
// tile of the layer (HT^2 accumulators: each scratch datum is read exactly once, by one wave); k = 16 consecutive points of one channel per
// group of MFMAs — lane (q, i) loads points p + 4 q .. + 3 of its row (32 contiguous bytes; a row's 16 points are one 128-byte line) and MFMA
// step s contracts the points {s, 4 + s, 8 + s, 12 + s}: any assignment of points to k works as long as A and B agree.  The operands of step
// i + 1 are requested before the MFMAs of step i.  The four waves' partial sums are combined through LDS in wave order (deterministic). ----
HD int f64m_num_layers(const F64Args& a) {
    int t = 0;
    for (int ni = 0; ni < a.nnets; ++ni) t += (a.net[ni].nl - 2 > 0) ? a.net[ni].nl - 2 : 0;
    return t;
}
constexpr int F64M_DWT_WAVES = 4;
template <int HT> struct F64mDwtAcc { LVd<HT * HT * 4> acc; LVd<HT> bsum; };
// one wave's partial sums
template <int HT>
DEV void f64m_dwt_wave(int ni, int lyr, int b, int w, const F64Args& a, F64mDwtAcc<HT>& R) {
    const F64Net& n = a.net[ni];
    const int n_out = n.sizes[lyr + 1], n_in = n.sizes[lyr], C = a.C;
    const int BP = f64m_block(a);
    const int lo = b * BP, hi = (lo + BP < a.npts) ? lo + BP : a.npts;
    const double* S = a.scratch;
    PINN_LANES(l) {
        PINN_UNROLL for (int e = 0; e < HT * HT * 4; ++e) R.acc(l, e) = 0.0;
        PINN_UNROLL for (int t = 0; t < HT; ++t) R.bsum(l, t) = 0.0;
    }
    LVd<HT * 4> Af[2], Bf[2];
    // UNCONDITIONAL 32-byte loads (rows clamped to a valid one, the four points always inside the scratch row: npad is a multiple of the block
    // size), masked by selects afterwards: a predicate per element turns every load into its own exec-masked 8-byte access
    auto load_step = [&](LVd<HT * 4>& A_, LVd<HT * 4>& B_, int p, int c) {
        PINN_LANES(l) {
            const int pp = p + 4 * (l >> 4);
            PINN_UNROLL for (int t = 0; t < HT; ++t) {
                const int m = 16 * t + (l & 15);
                const double* rz = S + f64m_six(a, (size_t)n.r_dz[lyr] + (size_t)(m < n_out ? m : 0) * C + c, pp);
                const double* ri = S + f64m_six(a, (size_t)n.r_post[lyr - 1] + (size_t)(m < n_in ? m : 0) * C + c, pp);
                double za[4], ia[4];
                ld4_f64(rz, za);
                ld4_f64(ri, ia);
                PINN_UNROLL for (int s4 = 0; s4 < 4; ++s4) {
                    A_(l, 4 * t + s4) = (m < n_out && pp + s4 < hi) ? za[s4] : 0.0;
                    B_(l, 4 * t + s4) = (m < n_in && pp + s4 < hi) ? ia[s4] : 0.0;
                }
            }
        }
    };
    auto mma_step = [&](const LVd<HT * 4>& A_, const LVd<HT * 4>& B_, int c) {
        PINN_UNROLL for (int s4 = 0; s4 < 4; ++s4)
            PINN_UNROLL for (int to = 0; to < HT; ++to) {
                if (16 * to >= n_out) break;
                PINN_UNROLL for (int t = 0; t < HT; ++t) {
                    if (16 * t >= n_in) break;
                    mfma_f64(R.acc, (to * HT + t) * 4, 1, A_, 4 * to + s4, B_, 4 * t + s4);
                }
            }
        if (c == 0) {                                           // bias gradient = sum over the points of the value channel's dZ
            PINN_LANES(l) {
                PINN_UNROLL for (int to = 0; to < HT; ++to)
                    R.bsum(l, to) += (A_(l, 4 * to) + A_(l, 4 * to + 1)) + (A_(l, 4 * to + 2) + A_(l, 4 * to + 3));
            }
        }
    };
    const int CE = a.use_ceff ? n.ceff : C;                      // (behind a sliced tile kernel only the channel prefix of the network was written)
    const int nsteps = ((hi - lo + 15) / 16) * CE;               // step i: points lo + 16 (i / CE) .., channel i % CE; this wave: i = w, w + 4, ...
    int i = w;
#if PINN_F64M_DWT_SINGLE
    for (; i < nsteps; i += F64M_DWT_WAVES) { load_step(Af[0], Bf[0], lo + 16 * (i / CE), i % CE); mma_step(Af[0], Bf[0], i % CE); }
#endif
    if (i < nsteps) load_step(Af[0], Bf[0], lo + 16 * (i / CE), i % CE);
    for (; i < nsteps; i += 2 * F64M_DWT_WAVES) {
        const int i1 = i + F64M_DWT_WAVES, i2 = i + 2 * F64M_DWT_WAVES;
        if (i1 < nsteps) load_step(Af[1], Bf[1], lo + 16 * (i1 / CE), i1 % CE);
        mma_step(Af[0], Bf[0], i % CE);
        if (i2 < nsteps) load_step(Af[0], Bf[0], lo + 16 * (i2 / CE), i2 % CE);
        if (i1 < nsteps) mma_step(Af[1], Bf[1], i1 % CE);
    }
}
// wave w's turn of the fixed-order combine through `lds` ([HT * HT * 4 + HT][64] doubles): wave 0 stores, waves 1, 2 add, wave 3 adds and writes the slab
template <int HT>
DEV void f64m_dwt_combine(int ni, int lyr, int b, int w, const F64Args& a, F64mDwtAcc<HT>& R, double* lds) {
    const F64Net& n = a.net[ni];
    const int n_out = n.sizes[lyr + 1], n_in = n.sizes[lyr];
    constexpr int NE = HT * HT * 4;
    PINN_LANES(l) {
        PINN_UNROLL for (int e = 0; e < NE; ++e) {
            const double v = (w == 0) ? R.acc(l, e) : lds[e * 64 + l] + R.acc(l, e);
            if (w < F64M_DWT_WAVES - 1) lds[e * 64 + l] = v;
            else {
                const int to = e / (HT * 4), t = (e / 4) % HT, r = e % 4;
                const int m = 16 * to + (l >> 4) + 4 * r, k = 16 * t + (l & 15);
                if (m < n_out && k < n_in) a.slab[(size_t)b * a.nent + n.ent0 + (n.woff[lyr] - n.theta0) + m + (size_t)k * n_out] = v;
            }
        }
        PINN_UNROLL for (int to = 0; to < HT; ++to) {
            const double v = (w == 0) ? R.bsum(l, to) : lds[(NE + to) * 64 + l] + R.bsum(l, to);
            if (w < F64M_DWT_WAVES - 1) lds[(NE + to) * 64 + l] = v;
            else R.bsum(l, to) = v;
        }
    }
    if (w == F64M_DWT_WAVES - 1) {                               // the four lane groups hold the row's points 4 q .. 4 q + 3 of every step: sum them
        PINN_UNROLL for (int to = 0; to < HT; ++to) lv_qsum(R.bsum, to);
        PINN_LANES(l) {
            PINN_UNROLL for (int to = 0; to < HT; ++to) {
                const int m = 16 * to + (l & 15);
                if ((l >> 4) == 0 && m < n_out) a.slab[(size_t)b * a.nent + n.ent0 + (n.boff[lyr] - n.theta0) + m] = R.bsum(l, to);
            }
        }
    }
}
HD bool f64m_dwt_locate(int idx, const F64Args& a, int& ni, int& lyr) {
    for (int i = 0; i < a.nnets; ++i)
        for (int l = 1; l < a.net[i].nl - 1; ++l) {
            if (idx == 0) { ni = i; lyr = l; return true; }
            --idx;
        }
    return false;
}

// ---- the remaining slab entries (first / last layer, PDE parameters, the block's sum of squares): the tile kernel left them summed per TILE
// (F64Args::tpart); column e of block b = its tiles' values added in tile order -> the block's slab.  Runs as one more workgroup row of the dW
// kernel's launch (k_f64m_dwt, blockIdx.x == number of hidden-to-hidden layers). ----
HD void f64m_tsum_entry(int e, int b, const F64Args& a) {
    int ent = -1;
    bool grad_entry = true;
    if (e >= a.tp_p) {
        if (e - a.tp_p < a.ne) ent = a.ent_p + (e - a.tp_p);
        else { ent = a.nent - 1; grad_entry = false; }            // the sum of squares: needed in every mode
    } else {
        for (int ni = 0; ni < a.nnets; ++ni) {
            const F64Net& n = a.net[ni];
            const int L = n.nl - 1, n1 = n.sizes[1], nL = n.sizes[L];
            const int r = e - n.tp0;
            if (r < 0 || r >= (n.d + 1) * n1 + nL + 1) continue;
            if (r < n.d * n1) ent = n.ent0 + (n.woff[0] - n.theta0) + r;                            // W_0[m + i * n_1]
            else if (r < (n.d + 1) * n1) ent = n.ent0 + (n.boff[0] - n.theta0) + (r - n.d * n1);
            else if (r < (n.d + 1) * n1 + nL) ent = n.ent0 + (n.woff[L] - n.theta0) + (r - (n.d + 1) * n1);
            else ent = n.ent0 + (n.boff[L] - n.theta0);
            break;
        }
    }
    if (ent < 0 || (grad_entry && a.mode != 0)) return;
    const int tpb = f64m_block(a) / a.tile_pts, nt = (a.npts + a.tile_pts - 1) / a.tile_pts;
    const int t0 = b * tpb, t1 = (t0 + tpb < nt) ? t0 + tpb : nt;
    if (!grad_entry && a.nsub > 0) {                              // merged launch: one sum of squares per sub-term (a tile belongs to exactly one)
        for (int s = 0; s < a.nsub; ++s) {
            const int u0 = t0 > a.sub_tile0[s] ? t0 : a.sub_tile0[s], u1 = t1 < a.sub_tile0[s + 1] ? t1 : a.sub_tile0[s + 1];
            double sum = 0.0;
            for (int t = u0; t < u1; ++t) sum += a.tpart[(size_t)t * (size_t)a.ntp + e];
            a.slab[(size_t)b * a.nent + (a.nent - a.nsub + s)] = sum;
        }
        return;
    }
    double sum = 0.0;
    for (int t = t0; t < t1; ++t) sum += a.tpart[(size_t)t * (size_t)a.ntp + e];
    a.slab[(size_t)b * a.nent + ent] = sum;
}

// ---- the kernel table: (inputs, jet set, HT) -> launchers; matched against a term's float64 kernel in f64.cpp ----
struct F64MKernel {
    int sliced = 0;                     // 1: family 4s (pinn_kernels6.hpp): channel-sliced GEMM passes through the scratch rows — needs the scratch in every mode
    int D, NPAIR, HT, PG;
    unsigned D1MASK, HI;
    unsigned long long PAIRS;
    void (*launch_tile)(const F64Args&, plat_stream);
    void (*launch_dwt)(const F64Args&, plat_stream);
};
std::deque<F64MKernel>& f64m_registry();

#ifdef PINN_EMU
template <class J, int HT, int PG> void launch_f64m_tile(const F64Args& a, plat_stream) {
    const int nt = (a.npts + 16 * PG - 1) / (16 * PG);
    for (int t = 0; t < nt; ++t) f64m_tile<J, HT, PG, ACT_TANH>(t, a);
}
template <int HT> void launch_f64m_dwt(const F64Args& a, plat_stream) {
    const int nb = (a.npts + f64m_block(a) - 1) / f64m_block(a), nl = f64m_num_layers(a);
    for (int b = 0; b < nb; ++b)
        for (int e = 0; e < a.ntp; ++e) f64m_tsum_entry(e, b, a);
    if (a.mode != 0) return;
    std::vector<double> lds((size_t)(HT * HT * 4 + HT) * 64);
    for (int b = 0; b < nb; ++b)
        for (int li = 0; li < nl; ++li) {
            int ni = 0, lyr = 1;
            if (!f64m_dwt_locate(li, a, ni, lyr)) continue;
            for (int w = 0; w < F64M_DWT_WAVES; ++w) {          // (the combine is sequential in wave order: running the waves one after another IS the device's result)
                F64mDwtAcc<HT> R;
                f64m_dwt_wave<HT>(ni, lyr, b, w, a, R);
                f64m_dwt_combine<HT>(ni, lyr, b, w, a, R, lds.data());
            }
        }
}
#else
template <class J, int HT, int PG> __global__ void __launch_bounds__(64, (HT * PG * J::C <= 8) ? ((HT * PG * J::C <= 4) ? PINN_F64M_WAVES_SMALL : 2) : 1) k_f64m_tile(const F64Args a) {
    f64m_tile<J, HT, PG, ACT_TANH>((int)blockIdx.x, a);
}
template <int HT> __global__ void __launch_bounds__(64 * F64M_DWT_WAVES, PINN_F64M_DWT_SINGLE ? 2 : 1) k_f64m_dwt(const F64Args a) {
    __shared__ double lds[(HT * HT * 4 + HT) * 64];
    int ni = 0, lyr = 1;
    if (!f64m_dwt_locate((int)blockIdx.x, a, ni, lyr)) {          // the workgroup row behind the layers: this block's sums of the tile partials
        for (int e = (int)threadIdx.x; e < a.ntp; e += 64 * F64M_DWT_WAVES) f64m_tsum_entry(e, (int)blockIdx.y, a);
        return;
    }
    if (a.mode != 0) return;
    const int w = (int)(threadIdx.x >> 6), b = (int)blockIdx.y;
    F64mDwtAcc<HT> R;
    f64m_dwt_wave<HT>(ni, lyr, b, w, a, R);
    for (int turn = 0; turn < F64M_DWT_WAVES; ++turn) {
        if (w == turn) f64m_dwt_combine<HT>(ni, lyr, b, w, a, R, lds);
        __syncthreads();
    }
}
template <class J, int HT, int PG> void launch_f64m_tile(const F64Args& a, plat_stream st) {
    const int nt = (a.npts + 16 * PG - 1) / (16 * PG);
    hipLaunchKernelGGL((k_f64m_tile<J, HT, PG>), dim3(nt), dim3(64), 0, st, a);
}
template <int HT> void launch_f64m_dwt(const F64Args& a, plat_stream st) {
    const int nb = (a.npts + f64m_block(a) - 1) / f64m_block(a), nl = f64m_num_layers(a);
    hipLaunchKernelGGL((k_f64m_dwt<HT>), dim3(nl + 1, nb), dim3(64 * F64M_DWT_WAVES), 0, st, a);
}
#endif

// ---- float64 optimiser loop (f64.cpp: f64_adam_steps): points redrawn by the fp32 samplers -> double, Adam in double, the step's total loss ----
DEV void f64_cvt_elem(int i, const float* src, double* dst) { dst[i] = (double)src[i]; }
// the coordinate-only part S of an affine residual (F64Sub::lin), once per point set: the term's tape run at point p with the slots at zero,
// S = k + sum_j coef_j * row_j (rows that depend on coordinates, constants and default parameters only — f64.cpp: f64_affine)
constexpr int F64_LIN_MAX_TERMS = 8;
struct F64LinSrcArgs {
    const double* pts; int N, dt, np, nslots, nops;
    double pdef[MAX_PARAMS];
    const rp::Instr* prog; const double* imm;
    int nterms, row[F64_LIN_MAX_TERMS];
    double coef[F64_LIN_MAX_TERMS], k;
    double* out;                         // [N]
};
DEV void f64_lin_src_point(int p, const F64LinSrcArgs& a) {
    double v[F64_MAX_ROWS];
    for (int i = 0; i < a.dt; ++i) v[i] = a.pts[(size_t)p * a.dt + i];
    for (int k = 0; k < a.np; ++k) v[a.dt + k] = a.pdef[k];
    for (int s = 0; s < a.nslots; ++s) v[a.dt + a.np + s] = 0.0;
    const int R0 = a.dt + a.np + a.nslots;
    for (int o = 0; o < a.nops; ++o) {
        const rp::Instr ins = a.prog[o];
        const double va = rp::is_nullary(ins.code) ? 0.0 : v[ins.a];
        const double vb = rp::is_binary(ins.code) ? v[ins.b] : 0.0;
        v[R0 + o] = (ins.code == rp::OP_DATA) ? 0.0 : rp::apply<double, double>(ins.code, va, vb, a.imm[o]);
    }
    double s = a.k;
    for (int j = 0; j < a.nterms; ++j) s = vfma(a.coef[j], v[a.row[j]], s);
    a.out[p] = s;
}
// redrawn set of a term behind a periodic input embedding (f64.cpp): the float coordinates widened, feature row du + k = sin / cos (omega_k x_src_k) in double
struct F64EmbedArgs { double* pts; const float* upts; int n, d, du, ncols; int src[4], is_cos[4]; double omega[4]; };      // upts: the drawn coordinates [n][du]
HD void f64_embed_point(int p, const F64EmbedArgs& a) {
    for (int j = 0; j < a.du; ++j) a.pts[(size_t)p * a.d + j] = (double)a.upts[(size_t)p * a.du + j];
    for (int k = 0; k < a.ncols; ++k) {
        const double ph = a.omega[k] * a.pts[(size_t)p * a.d + a.src[k]];
        a.pts[(size_t)p * a.d + a.du + k] = a.is_cos[k] ? cos(ph) : sin(ph);
    }
}
DEV void f64_adam_elem(int i, double* theta, double* m, double* v, const double* g, double lr, double b1, double b2, double eps, double c1, double c2) {
    const double gi = g[i];
    const double mi = b1 * m[i] + (1.0 - b1) * gi, vi = b2 * v[i] + (1.0 - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    theta[i] -= lr * (mi * c1) / (sqrt(vi * c2) + eps);
}
#ifdef PINN_EMU
inline void launch_f64_cvt(const float* src, double* dst, int64_t n, plat_stream) { for (int64_t i = 0; i < n; ++i) dst[i] = (double)src[i]; }
inline void launch_f64_embed(const F64EmbedArgs& a, plat_stream) { for (int p = 0; p < a.n; ++p) f64_embed_point(p, a); }
inline void launch_f64_lin_src(const F64LinSrcArgs& a, plat_stream) { for (int p = 0; p < a.N; ++p) f64_lin_src_point(p, a); }
inline void launch_f64_adam(double* theta, double* m, double* v, const double* g, int P, double lr, double b1, double b2, double eps, double c1, double c2, plat_stream) {
    for (int i = 0; i < P; ++i) f64_adam_elem(i, theta, m, v, g, lr, b1, b2, eps, c1, c2);
}
inline void launch_f64_total(double* hist, int step, const double* sumsq, const double* w_over_n, int K, plat_stream) {
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += w_over_n[k] * sumsq[k];
    hist[step] = s;
}
// the resident loop's update: Adam on every element + (one thread) the step's total weighted loss — one launch instead of two
inline void launch_f64_adam_total(double* theta, double* m, double* v, const double* g, int P, double lr, double b1, double b2, double eps, double c1, double c2,
                                  double* hist, int step, const double* sumsq, const double* w_over_n, int K, plat_stream st) {
    launch_f64_total(hist, step, sumsq, w_over_n, K, st);
    launch_f64_adam(theta, m, v, g, P, lr, b1, b2, eps, c1, c2, st);
}
inline void launch_f64_narrow(const double* src, float* dst, int64_t n, plat_stream) { for (int64_t i = 0; i < n; ++i) dst[i] = (float)src[i]; }
#else
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_cvt(const float* src, double* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_adam(double* theta, double* m, double* v, const double* g, int P, double lr, double b1, double b2, double eps, double c1, double c2) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < P) f64_adam_elem(i, theta, m, v, g, lr, b1, b2, eps, c1, c2);
}
template <int UNUSED> __global__ void k_f64_total(double* hist, int step, const double* sumsq, const double* w_over_n, int K) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { double s = 0.0; for (int k = 0; k < K; ++k) s += w_over_n[k] * sumsq[k]; hist[step] = s; }
}
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_embed(const F64EmbedArgs a) {
    const int p = (int)(blockIdx.x * 256 + threadIdx.x);
    if (p < a.n) f64_embed_point(p, a);
}
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_lin_src(const F64LinSrcArgs a) {
    const int p = (int)(blockIdx.x * 256 + threadIdx.x);
    if (p < a.N) f64_lin_src_point(p, a);
}
inline void launch_f64_lin_src(const F64LinSrcArgs& a, plat_stream st) { hipLaunchKernelGGL((k_f64_lin_src<0>), dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st, a); }
inline void launch_f64_embed(const F64EmbedArgs& a, plat_stream st) { hipLaunchKernelGGL((k_f64_embed<0>), dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, st, a); }
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_adam_total(double* theta, double* m, double* v, const double* g, int P, double lr, double b1, double b2, double eps, double c1, double c2,
                                                                              double* hist, int step, const double* sumsq, const double* w_over_n, int K) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < P) f64_adam_elem(i, theta, m, v, g, lr, b1, b2, eps, c1, c2);
    if (i == P) { double s = 0.0; for (int k = 0; k < K; ++k) s += w_over_n[k] * sumsq[k]; hist[step] = s; }      // (same association as k_f64_total)
}
inline void launch_f64_adam_total(double* theta, double* m, double* v, const double* g, int P, double lr, double b1, double b2, double eps, double c1, double c2,
                                  double* hist, int step, const double* sumsq, const double* w_over_n, int K, plat_stream st) {
    hipLaunchKernelGGL((k_f64_adam_total<0>), dim3((P + 1 + 255) / 256), dim3(256), 0, st, theta, m, v, g, P, lr, b1, b2, eps, c1, c2, hist, step, sumsq, w_over_n, K);
}
inline void launch_f64_cvt(const float* src, double* dst, int64_t n, plat_stream st) { hipLaunchKernelGGL((k_f64_cvt<0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n); }
inline void launch_f64_adam(double* theta, double* m, double* v, const double* g, int P, double lr, double b1, double b2, double eps, double c1, double c2, plat_stream st) {
    hipLaunchKernelGGL((k_f64_adam<0>), dim3((P + 255) / 256), dim3(256), 0, st, theta, m, v, g, P, lr, b1, b2, eps, c1, c2);
}
inline void launch_f64_total(double* hist, int step, const double* sumsq, const double* w_over_n, int K, plat_stream st) {
    hipLaunchKernelGGL((k_f64_total<0>), dim3(1), dim3(64), 0, st, hist, step, sumsq, w_over_n, K);
}
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64_narrow(const double* src, float* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
inline void launch_f64_narrow(const double* src, float* dst, int64_t n, plat_stream st) { hipLaunchKernelGGL((k_f64_narrow<0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n); }
#endif

// ---- slab reduction of the matrix-pipe path: FOUR lanes per entry (blocks part, part + 4, ... each; (p0 + p1) + (p2 + p3)), four times the
// threads and a quarter of the dependent adds of family 4's one-thread-per-entry kernel (38 us per term on the bench workload) ----
constexpr int F64M_RED_LANES = 16;       // lanes per slab entry: lane i adds the blocks i, i + 16, ...; the lanes' sums meet in an xor butterfly (1, 2, 4, 8)
DEV double f64m_reduce_part(int e, int part, const F64ReduceArgs& a) {
    double s = 0.0;
    for (int b = part; b < a.nblocks; b += F64M_RED_LANES) s += a.slab[(size_t)b * a.nent + e];
    return s;
}
#ifdef PINN_EMU
inline void launch_f64m_reduce(const F64ReduceArgs& a, plat_stream) {
    for (int e = 0; e < a.nent; ++e) {
        if (e < a.nent - a.nsq && !a.with_grad) continue;
        double t[F64M_RED_LANES], u[F64M_RED_LANES];
        for (int i = 0; i < F64M_RED_LANES; ++i) t[i] = f64m_reduce_part(e, i, a);
        for (int o = 1; o < F64M_RED_LANES; o <<= 1) {
            for (int i = 0; i < F64M_RED_LANES; ++i) u[i] = t[i] + t[i ^ o];
            for (int i = 0; i < F64M_RED_LANES; ++i) t[i] = u[i];
        }
        f64_reduce_write(e, t[0], a);
    }
}
#else
template <int UNUSED> __global__ void __launch_bounds__(256) k_f64m_reduce(const F64ReduceArgs a) {
    const int gid = (int)(blockIdx.x * 256 + threadIdx.x), e = gid / F64M_RED_LANES, part = gid % F64M_RED_LANES;
    const bool live = e < a.nent && (e >= a.nent - a.nsq || a.with_grad);
    double s = live ? f64m_reduce_part(e, part, a) : 0.0;
    PINN_UNROLL for (int o = 1; o < F64M_RED_LANES; o <<= 1) s += __shfl_xor(s, o, 64);
    if (live && part == 0) f64_reduce_write(e, s, a);
}
inline void launch_f64m_reduce(const F64ReduceArgs& a, plat_stream st) {
    hipLaunchKernelGGL((k_f64m_reduce<0>), dim3((F64M_RED_LANES * a.nent + 255) / 256), dim3(256), 0, st, a);
}
#endif

template <int D, unsigned D1MASK, unsigned long long PAIRS, int NPAIR, unsigned HI, int HT> F64MKernel make_f64m_kernel() {
    using J = JetSet<D1MASK, PAIRS, NPAIR, HI>;
    static_assert(J::NLAP == 0, "the float64 kernels carry plain derivative channels (no forward-Laplacian channel)");
    // column groups per wave: as many point groups as keep the two operand arrays (2 x HT * 4 * NCG doubles per lane) inside the register file
    // ... and, where the channel count allows it, few enough (NCG <= 2: the operand arrays take <= 128 registers) that TWO waves fit a SIMD —
    // one wave's element-wise float64 work (the activation alone is ~45 f64 instructions per element) then runs under the other's MFMAs
    constexpr int CAP = PINN_F64M_CAP;
    constexpr int PG = (J::C >= CAP) ? 1 : CAP / J::C;
    static_assert(HT * PG * J::C <= 24, "operand arrays of this (jet set, width) pair exceed the register file: keep family 4");
    F64MKernel k;
    k.D = D; k.NPAIR = NPAIR; k.HT = HT; k.PG = PG; k.D1MASK = D1MASK; k.HI = HI; k.PAIRS = PAIRS;
    k.launch_tile = &launch_f64m_tile<J, HT, PG>;
    k.launch_dwt = &launch_f64m_dwt<HT>;
    return k;
}
struct F64MRegistrar { explicit F64MRegistrar(const F64MKernel& k) { f64m_registry().push_back(k); } };
#define PINN_INSTANTIATE_F64M(NAME, D, D1MASK, PAIRS, NPAIR, HI, HT) \
    namespace { pk::F64MRegistrar NAME##_regf64m(pk::make_f64m_kernel<D, D1MASK, PAIRS, NPAIR, HI, HT>()); }

}  // namespace pk
