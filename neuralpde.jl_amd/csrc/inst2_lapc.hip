// family 2 kernels with a forward-Laplacian channel (JetSet::LAP): the pure second derivatives travel as ONE summed channel.
//   4x64 / 5x128, 2-D: {u, u_x, u_y, u_xx + u_yy}             (Poisson interior term, Navier-Stokes momentum equations)
//   6x128, 4-D:        {u, u_t, u_x, u_y, u_z, u_xx+u_yy+u_zz} (heat equation in (t, x, y, z))
#include "spec_registry.hpp"
PINN_INSTANTIATE2_HI_SIN(f2_h64n3d2_lapc, 64, 3, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE2_HI(f2_h128n4d2_lapc, 128, 4, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE2_HI(f2_h128n5d4_lapc, 128, 5, 4, 0xF, 0ull, 0, 1, PINN_LAP(0xE))
// unit-test sizes (2 hidden layers of 128)
PINN_INSTANTIATE2_HI(f2_h128n1d2_lapc, 128, 1, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE2_HI(f2_h128n1d4_lapc, 128, 1, 4, 0xF, 0ull, 0, 1, PINN_LAP(0xE))
