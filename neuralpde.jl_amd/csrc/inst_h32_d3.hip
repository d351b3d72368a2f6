// widths 17..32, 3 inputs (e.g. test/NNPDE1's 3-D PDE with inner = 25): value-only and full-Hessian jet sets
#include "spec_registry.hpp"
#define HESS3 (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2))
PINN_INSTANTIATE(h32n0d3_val, 32, 0, 3, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n0d3_hess, 32, 0, 3, 0x7, HESS3, 6, 1)
PINN_INSTANTIATE(h32n1d3_val, 32, 1, 3, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n1d3_hess, 32, 1, 3, 0x7, HESS3, 6, 1)
PINN_INSTANTIATE(h32n2d3_val, 32, 2, 3, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n2d3_hess, 32, 2, 3, 0x7, HESS3, 6, 1)
