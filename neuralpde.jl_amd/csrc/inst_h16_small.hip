// small nets used by the unit tests (2 hidden layers of <=16): full 2-D Hessian jet set and value-only; 1-D too
#include "spec_registry.hpp"
PINN_INSTANTIATE_HI_SIN(h16n1d2_hess, 16, 1, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 1, 0u)
PINN_INSTANTIATE_HI_SIN(h16n1d2_val, 16, 1, 2, 0x0, 0ull, 0, 2, 0u)
// (+ the per-layer tanh / sigmoid variant: the reference's Lorenz parameter-estimation chains Dense(1, 8, tanh), Dense(8, 8, sigma), Dense(8, 1))
PINN_INSTANTIATE_HI_MIX(h16n1d1_lap, 16, 1, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1, 0u)
PINN_INSTANTIATE_HI_MIX(h16n1d1_val, 16, 1, 1, 0x0, 0ull, 0, 2, 0u)
// single-hidden-layer nets (e.g. the reference's system-of-PDEs test chains Dense(2,15,tanh) -> Dense(15,1))
PINN_INSTANTIATE(h16n0d2_hess, 16, 0, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 1)
PINN_INSTANTIATE(h16n0d2_val, 16, 0, 2, 0x0, 0ull, 0, 2)
// pure third / fourth derivatives: 1-D nets (the reference's 3rd-order ODE test, test/NNPDE1/nnpde__pde_iii_3rd_order_ode.jl)
// and the 2-D Kuramoto-Sivashinsky set u, u_t, u_x, u_xx, u_xxx, u_xxxx (docs/src/examples/ks.md:45-46)
PINN_INSTANTIATE_HI(h16n1d1_o4, 16, 1, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1, PINN_HI(0, 4))
PINN_INSTANTIATE_HI(h16n0d1_o4, 16, 0, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 1, PINN_HI(0, 4))
PINN_INSTANTIATE_HI(h16n1d2_ks, 16, 1, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 1, PINN_HI(1, 4))      // u(t, x)
PINN_INSTANTIATE_HI_SIN(h16n1d2_ks0, 16, 1, 2, 0x3, PINN_PAIR(0, 0, 0), 1, 1, PINN_HI(0, 4))     // u(x, t) as in docs/src/examples/ks.md
// 3-D value-only / gradient nets (the reference's heterogeneous-system test: u(x,y,z), v(y,x), h(z), p(x,z))
PINN_INSTANTIATE(h16n1d3_val, 16, 1, 3, 0x0, 0ull, 0, 2)
PINN_INSTANTIATE(h16n1d3_grad, 16, 1, 3, 0x7, 0ull, 0, 1)
// forward-Laplacian channel sets {u, u_x, u_y, u_xx + u_yy} for the small test nets
PINN_INSTANTIATE_HI_SIN(h16n1d2_lapc, 16, 1, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
PINN_INSTANTIATE_HI(h16n0d2_lapc, 16, 0, 2, 0x3, 0ull, 0, 1, PINN_LAP(0x3))
