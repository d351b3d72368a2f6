// width <= 16: the depths and input counts the unit-test file (inst_h16_small.hip) does not cover — 1 to 3 hidden layers, 1-2 inputs
#include "spec_registry.hpp"
#define HESS2 (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1))
PINN_INSTANTIATE(h16n0d1_val, 16, 0, 1, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h16n0d1_lap, 16, 0, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
PINN_INSTANTIATE_HI_MIX(h16n2d1_val, 16, 2, 1, 0x0, 0ull, 0, 2, 0u)
PINN_INSTANTIATE_HI_MIX(h16n2d1_lap, 16, 2, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1, 0u)
PINN_INSTANTIATE_HI_MIX(h16n2d2_val, 16, 2, 2, 0x0, 0ull, 0, 2, 0u)
PINN_INSTANTIATE_HI_MIX(h16n2d2_hess, 16, 2, 2, 0x3, HESS2, 3, 1, 0u)
PINN_INSTANTIATE_HI(h16n2d2_lapc, 16, 2, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
// 2-input, 2-hidden-layer nets with per-layer tanh / sigmoid (the uniform-activation kernels of this shape live in inst_h16_small.hip)
PINN_INSTANTIATE_HI_MIX(h16n1d2_val_mix, 16, 1, 2, 0x0, 0ull, 0, 2, 0u)
PINN_INSTANTIATE_HI_MIX(h16n1d2_hess_mix, 16, 1, 2, 0x3, HESS2, 3, 1, 0u)
