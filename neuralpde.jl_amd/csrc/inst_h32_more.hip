// widths 17..32 (the reference's tests use 18, 20, 25, 32): 1 to 3 hidden layers, 1-3 inputs, value-only and full-Hessian jet sets
// (+ the forward-Laplacian set in 2-D).  inst_h32_d1.hip holds the BASELINE config 1 kernels (3 x 32, d = 1).
#include "spec_registry.hpp"
#define HESS2 (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1))
#define HESS3 (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2))
PINN_INSTANTIATE(h32n0d1_val, 32, 0, 1, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n0d1_lap, 32, 0, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
PINN_INSTANTIATE(h32n1d1_val, 32, 1, 1, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n1d1_lap, 32, 1, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
PINN_INSTANTIATE(h32n0d2_val, 32, 0, 2, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n0d2_hess, 32, 0, 2, 0x3, HESS2, 3, 1)
PINN_INSTANTIATE(h32n1d2_val, 32, 1, 2, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n1d2_hess, 32, 1, 2, 0x3, HESS2, 3, 1)
PINN_INSTANTIATE_HI(h32n1d2_lapc, 32, 1, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE(h32n2d2_val, 32, 2, 2, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n2d2_hess, 32, 2, 2, 0x3, HESS2, 3, 1)
PINN_INSTANTIATE_HI(h32n2d2_lapc, 32, 2, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
