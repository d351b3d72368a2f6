// the networks of the reference's own GPU tests (test/CUDA/nnpde_cuda__*.jl): 5 x 20 sigma with one input (1-D ODE), 4 x 20 sigma with two
// inputs (1-D heat equation, Neumann conditions), 4 x 25 sigma with three inputs (2-D heat equation); deeper than the 1-3 hidden layers
// of the other tests' small nets
#include "spec_registry.hpp"
PINN_INSTANTIATE(h32n4d1_val, 32, 4, 1, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n4d1_d1, 32, 4, 1, 0x1, 0ull, 0, 2)                          // {u, u'}
PINN_INSTANTIATE(h32n3d2_val, 32, 3, 2, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h32n3d2_heat, 32, 3, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 1)          // {u, u_t, u_x, u_xx}
PINN_INSTANTIATE(h32n3d3_val, 32, 3, 3, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE_HI(h32n3d3_lapc, 32, 3, 3, 0x7, 0ull, 0, 1, PINN_LAP(0x6))     // {u, u_t, u_x, u_y, u_xx + u_yy}
