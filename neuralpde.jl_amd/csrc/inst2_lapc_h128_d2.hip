// forward-Laplacian sets, 128 wide, 2-D: {u, u_x, u_y, u_xx + u_yy} (Navier-Stokes momentum equations of BASELINE config 4: 5 x 128;
// 2 x 128 for the unit tests)
#include "spec_registry.hpp"
PINN_INSTANTIATE2_HI(f2_h128n4d2_lapc, 128, 4, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE2_HI(f2_h128n1d2_lapc, 128, 1, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
