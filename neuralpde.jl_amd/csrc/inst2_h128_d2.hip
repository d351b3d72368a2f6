// family 2, 128-wide nets (BASELINE config 4: 2-D, 5x128): two neuron tiles per wave, dW accumulated in the slab
#include "spec_registry.hpp"
PINN_INSTANTIATE2(f2_h128n4d2_lap, 128, 4, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 1)
PINN_INSTANTIATE2(f2_h128n4d2_val, 128, 4, 2, 0x0, 0ull, 0, 4)
// 2 hidden layers of 128: unit tests
PINN_INSTANTIATE2(f2_h128n1d2_lap, 128, 1, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 1)
PINN_INSTANTIATE2(f2_h128n1d2_val, 128, 1, 2, 0x0, 0ull, 0, 4)
// first derivatives only {u, u_x, u_y} (the pressure network of the cavity problem)
PINN_INSTANTIATE2(f2_h128n4d2_grad, 128, 4, 2, 0x3, 0ull, 0, 1)
PINN_INSTANTIATE2(f2_h128n1d2_grad, 128, 1, 2, 0x3, 0ull, 0, 1)
// one first derivative {u, u_x} / {u, u_y}: what the continuity equation reads from the velocity networks and the momentum equations
// from the pressure network (coupled equations launch one kernel per (equation, network) with that equation's own channel set)
#ifndef PINN_D1_PG
#define PINN_D1_PG 2
#endif
PINN_INSTANTIATE2(f2_h128n4d2_dx, 128, 4, 2, 0x1, 0ull, 0, PINN_D1_PG)
PINN_INSTANTIATE2(f2_h128n4d2_dy, 128, 4, 2, 0x2, 0ull, 0, PINN_D1_PG)
