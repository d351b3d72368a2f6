// widths 33..64, 3 inputs, 2 to 4 hidden layers: value-only and full-Hessian jet sets
#include "spec_registry.hpp"
#define HESS3 (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2))
PINN_INSTANTIATE2(f2_h64n1d3_val, 64, 1, 3, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n1d3_hess, 64, 1, 3, 0x7, HESS3, 6, 1)
PINN_INSTANTIATE2(f2_h64n2d3_val, 64, 2, 3, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n2d3_hess, 64, 2, 3, 0x7, HESS3, 6, 1)
PINN_INSTANTIATE2(f2_h64n3d3_val, 64, 3, 3, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n3d3_hess, 64, 3, 3, 0x7, HESS3, 6, 1)
