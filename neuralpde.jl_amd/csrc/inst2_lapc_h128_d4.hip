// forward-Laplacian sets, 128 wide, 4-D: {u, u_t, u_x, u_y, u_z, u_xx + u_yy + u_zz} (heat equation in (t, x, y, z), BASELINE config 5:
// 6 x 128; 2 x 128 for the unit tests)
#include "spec_registry.hpp"
PINN_INSTANTIATE2_HI(f2_h128n5d4_lapc, 128, 5, 4, 0xF, 0ull, 0, 1, PINN_LAP(0xE))
PINN_INSTANTIATE2_HI(f2_h128n1d4_lapc, 128, 1, 4, 0xF, 0ull, 0, 1, PINN_LAP(0xE))
