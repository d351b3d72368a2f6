// small nets used by the unit tests (2 hidden layers of <=16): full 2-D Hessian jet set and value-only; 1-D too
#include "spec_registry.hpp"
PINN_INSTANTIATE(h16n1d2_hess, 16, 1, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 1)
PINN_INSTANTIATE(h16n1d2_val, 16, 1, 2, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h16n1d1_lap, 16, 1, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1)
PINN_INSTANTIATE(h16n1d1_val, 16, 1, 1, 0x0, 0ull, 0, 2)
// single-hidden-layer nets (e.g. the reference's system-of-PDEs test chains Dense(2,15,tanh) -> Dense(15,1))
PINN_INSTANTIATE(h16n0d2_hess, 16, 0, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 1)
PINN_INSTANTIATE(h16n0d2_val, 16, 0, 2, 0x0, 0ull, 0, 2)
