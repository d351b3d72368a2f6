// family 4m for NARROW networks (r06): the reference's own regime — 12- to 32-wide nets on a few hundred points, Float64 by default
// (src/discretize.jl:432-449) — padded to HT = 4 row tiles ran sixteen times the MFMAs and weight-fragment loads of a 16-wide layer; a lone
// wave's tile latency is the whole cost of such an evaluation (tools/r06/time_small_f64.py, profiles/r06_small_f64.txt).  HT = 1 (<= 16 wide) and
// HT = 2 (<= 32 wide) of the 1-D, 2-D and 3-D sets; f64.cpp picks the smallest HT that covers a term's networks.
#include "spec_registry.hpp"
#include "pinn_kernels5.hpp"
PINN_INSTANTIATE_F64M(f64m_d1_v_1, 1, 0x0, 0ull, 0, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d1_h_1, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d2_v_1, 2, 0x0, 0ull, 0, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d2_g_1, 2, 0x3, 0ull, 0, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d2_p_1, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d2_b_1, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d2_h_1, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d2_v_2, 2, 0x0, 0ull, 0, 0u, 2)
PINN_INSTANTIATE_F64M(f64m_d2_g_2, 2, 0x3, 0ull, 0, 0u, 2)
PINN_INSTANTIATE_F64M(f64m_d2_p_2, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 0u, 2)
PINN_INSTANTIATE_F64M(f64m_d2_b_2, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 0u, 2)
PINN_INSTANTIATE_F64M(f64m_d2_h_2, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 0u, 2)
PINN_INSTANTIATE_F64M(f64m_d3_v_1, 3, 0x0, 0ull, 0, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d3_h_1, 3, 0x7, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2)), 6, 0u, 1)
PINN_INSTANTIATE_F64M(f64m_d3_v_2, 3, 0x0, 0ull, 0, 0u, 2)
PINN_INSTANTIATE_F64M(f64m_d3_h_2, 3, 0x7, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2)), 6, 0u, 2)
