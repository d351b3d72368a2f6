// family 4s: float64 evaluation on the matrix pipe, channel-sliced (pinn_kernels6.hpp; v_mfma_f64_16x16x4_f64) — the (jet set, width) pairs whose
// activations do not fit family 4m's registers: 128-wide nets (BASELINE configs 4 and 5) and the 3-D / 4-D jet sets.  f64.cpp prefers a family 4m
// kernel (inst_f64m.hip) where one exists and takes the smallest HT that covers a term's networks.
#include "spec_registry.hpp"
#include "pinn_kernels6.hpp"
// 2-D, 65..128-wide (BASELINE config 4: three 5 x 128 nets): value, gradient, {u, u_x, u_y, u_xx, u_yy}, {u, u_t, u_x, u_xx}, Hessian
PINN_INSTANTIATE_F64S(f64s_d2_v_8, 2, 0x0, 0ull, 0, 0u, 8)
PINN_INSTANTIATE_F64S(f64s_d2_g_8, 2, 0x3, 0ull, 0, 0u, 8)
PINN_INSTANTIATE_F64S(f64s_d2_p_8, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 0u, 8)
PINN_INSTANTIATE_F64S(f64s_d2_b_8, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 0u, 8)
PINN_INSTANTIATE_F64S(f64s_d2_h_8, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 0u, 8)
// 4-D (BASELINE config 5: 6 x 128, {u, u_t, u_x, u_y, u_z, u_tt .. u_zz}): value and the pure-second-derivative set, 64- and 128-wide
PINN_INSTANTIATE_F64S(f64s_d4_v_4, 4, 0x0, 0ull, 0, 0u, 4)
PINN_INSTANTIATE_F64S(f64s_d4_p_4, 4, 0xf, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1) | PINN_PAIR(2, 2, 2) | PINN_PAIR(3, 3, 3)), 4, 0u, 4)
PINN_INSTANTIATE_F64S(f64s_d4_v_8, 4, 0x0, 0ull, 0, 0u, 8)
PINN_INSTANTIATE_F64S(f64s_d4_p_8, 4, 0xf, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1) | PINN_PAIR(2, 2, 2) | PINN_PAIR(3, 3, 3)), 4, 0u, 8)
// 3-D: value and the Hessian set, 64- and 128-wide
PINN_INSTANTIATE_F64S(f64s_d3_v_4, 3, 0x0, 0ull, 0, 0u, 4)
PINN_INSTANTIATE_F64S(f64s_d3_h_4, 3, 0x7, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2)), 6, 0u, 4)
PINN_INSTANTIATE_F64S(f64s_d3_v_8, 3, 0x0, 0ull, 0, 0u, 8)
PINN_INSTANTIATE_F64S(f64s_d3_h_8, 3, 0x7, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2)), 6, 0u, 8)
// 1-D, 65..128-wide: value, {u, u', u''}
PINN_INSTANTIATE_F64S(f64s_d1_v_8, 1, 0x0, 0ull, 0, 0u, 8)
PINN_INSTANTIATE_F64S(f64s_d1_h_8, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 0u, 8)
