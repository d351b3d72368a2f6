// family 2, BASELINE config 5: 4-D (t,x,y,z) heat inverse problem, 6x128 nets, jet set {u, u_t, u_x, u_y, u_z, u_xx, u_yy, u_zz}
#include "spec_registry.hpp"
PINN_INSTANTIATE2(f2_h128n5d4_heat, 128, 5, 4, 0xF, (PINN_PAIR(0, 1, 1) | PINN_PAIR(1, 2, 2) | PINN_PAIR(2, 3, 3)), 3, 1)
PINN_INSTANTIATE2(f2_h128n5d4_val, 128, 5, 4, 0x0, 0ull, 0, 4)
// unit-test size (2 hidden layers)
PINN_INSTANTIATE2(f2_h128n1d4_heat, 128, 1, 4, 0xF, (PINN_PAIR(0, 1, 1) | PINN_PAIR(1, 2, 2) | PINN_PAIR(2, 3, 3)), 3, 1)
PINN_INSTANTIATE2(f2_h128n1d4_val, 128, 1, 4, 0x0, 0ull, 0, 4)
