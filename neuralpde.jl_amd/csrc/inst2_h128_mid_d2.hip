// 128-wide nets with 3 or 4 hidden layers, 2 inputs (between the 2 x 128 unit-test shape and BASELINE config 4's 5 x 128): value-only,
// {u, u_x, u_y, u_xx, u_yy} and the forward-Laplacian set
#include "spec_registry.hpp"
PINN_INSTANTIATE2(f2_h128n2d2_val, 128, 2, 2, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h128n2d2_lap, 128, 2, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 1)
PINN_INSTANTIATE2_HI(f2_h128n2d2_lapc, 128, 2, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE2(f2_h128n3d2_val, 128, 3, 2, 0x0, 0ull, 0, 4)
PINN_INSTANTIATE2(f2_h128n3d2_lap, 128, 3, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 1)
PINN_INSTANTIATE2_HI(f2_h128n3d2_lapc, 128, 3, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
