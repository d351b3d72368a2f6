// 3-input nets of width <= 16 (1-3 hidden layers): value-only and full-Hessian jet sets (u, u_i, u_ij) — covers the reference's 3-D
// tests (e.g. u_t = u_xx + u_yy on (t, x, y)) at the small widths its test-suite uses
#include "spec_registry.hpp"
#define HESS3 (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2))
PINN_INSTANTIATE(h16n0d3_val, 16, 0, 3, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h16n0d3_hess, 16, 0, 3, 0x7, HESS3, 6, 1)
PINN_INSTANTIATE(h16n1d3_hess, 16, 1, 3, 0x7, HESS3, 6, 1)
PINN_INSTANTIATE(h16n2d3_val, 16, 2, 3, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h16n2d3_hess, 16, 2, 3, 0x7, HESS3, 6, 1)
