// family 4m: float64 evaluation on the matrix pipe (pinn_kernels5.hpp; v_mfma_f64_16x16x4_f64).  One (jet set, HT) pair per line, HT = hidden
// width / 16 rounded up; f64.cpp picks the smallest HT that covers a term's networks and keeps family 4 (inst_f64.hip) for everything else.
#include "spec_registry.hpp"
#include "pinn_kernels5.hpp"
// 1-D: value, {u, u', u''} — the reference's own regime (12-32 wide nets) and BASELINE config 1 (3 x 32)
PINN_INSTANTIATE_F64M(f64m_d1_v_2, 1, 0x0, 0ull, 0, 0u, 2)
PINN_INSTANTIATE_F64M(f64m_d1_h_2, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 0u, 2)
// 2-D, 64-wide (BASELINE configs 2 and 3): value (boundary terms), gradient, pure second derivatives (Poisson), {u, u_t, u_x, u_xx} (Burgers), Hessian
PINN_INSTANTIATE_F64M(f64m_d2_v_4, 2, 0x0, 0ull, 0, 0u, 4)
PINN_INSTANTIATE_F64M(f64m_d2_g_4, 2, 0x3, 0ull, 0, 0u, 4)
PINN_INSTANTIATE_F64M(f64m_d2_p_4, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 0u, 4)
PINN_INSTANTIATE_F64M(f64m_d2_b_4, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 0u, 4)
PINN_INSTANTIATE_F64M(f64m_d2_h_4, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 0u, 4)
