// widths 33..64 with 2 or 3 hidden layers, 2 inputs (e.g. docs/src/tutorials: Dense(2, 40, tanh), Dense(40, 40, tanh), Dense(40, 1)):
// value-only, full-Hessian and forward-Laplacian jet sets on the neuron-split kernels
#include "spec_registry.hpp"
#define HESS2 (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1))
PINN_INSTANTIATE2(f2_h64n1d2_val, 64, 1, 2, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n1d2_hess, 64, 1, 2, 0x3, HESS2, 3, 1)
PINN_INSTANTIATE2_HI(f2_h64n1d2_lapc, 64, 1, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE2(f2_h64n2d2_val, 64, 2, 2, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h64n2d2_hess, 64, 2, 2, 0x3, HESS2, 3, 1)
PINN_INSTANTIATE2_HI(f2_h64n2d2_lapc, 64, 2, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
// 5 layers deep (4 hidden): the full-Hessian set next to the Poisson / Burgers / KS sets of inst2_h64_d2*.hip
PINN_INSTANTIATE2(f2_h64n3d2_hess, 64, 3, 2, 0x3, HESS2, 3, 1)
