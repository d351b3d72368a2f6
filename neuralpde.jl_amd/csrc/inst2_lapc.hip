// family 2 kernels with a forward-Laplacian channel (JetSet::LAP): the pure second derivatives travel as ONE summed channel.
//   4x64, 2-D: {u, u_x, u_y, u_xx + u_yy}   (Poisson interior term: the bench kernel) — tanh, sigmoid and sin variants
// (the 128-wide sets live in inst2_lapc_h128*.hip: one translation unit per big spec keeps the parallel build balanced)
#include "spec_registry.hpp"
PINN_INSTANTIATE2_HI_SIN(f2_h64n3d2_lapc, 64, 3, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
