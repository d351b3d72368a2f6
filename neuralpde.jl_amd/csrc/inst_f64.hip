// family 4: float64 evaluation, one lane per point (pinn_kernels4.hpp).  Layer widths and depth are run-time values: one kernel per
// (network inputs, jet set) x {tanh, sigmoid, sin}.  V = value only; the "Hessian" sets carry every first and second derivative of
// their axes (any residual of order <= 2 is served; the pair mask holds 8 pairs: the 4-D Hessian's 10 do not fit); the 1-D set with orders 3 and 4, r05: 2-D and 3-D Hessian sets with the pure
// third and fourth derivatives of every axis; the 4-D set with the pure second derivatives.
#include "spec_registry.hpp"
#include "pinn_kernels4.hpp"
PINN_INSTANTIATE_F64(f64_d1_v, 1, 0x0, 0ull, 0, 0u)
PINN_INSTANTIATE_F64(f64_d1_h, 1, 0x1, PINN_PAIR(0, 0, 0), 1, 0u)
PINN_INSTANTIATE_F64(f64_d1_h4, 1, 0x1, PINN_PAIR(0, 0, 0), 1, PINN_HI(0, 4))
PINN_INSTANTIATE_F64(f64_d2_v, 2, 0x0, 0ull, 0, 0u)
PINN_INSTANTIATE_F64(f64_d2_g, 2, 0x3, 0ull, 0, 0u)
PINN_INSTANTIATE_F64(f64_d2_p, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 0u)            // r05: {u, u_x, u_y, u_xx, u_yy} (Poisson: C = 5 instead of the Hessian set's 6)
PINN_INSTANTIATE_F64(f64_d2_b, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 0u)                                   // r05: {u, u_t, u_x, u_xx} (Burgers: C = 4)
PINN_INSTANTIATE_F64(f64_d2_h, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, 0u)
PINN_INSTANTIATE_F64(f64_d2_h4, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 1, 1)), 3, (PINN_HI(0, 4) | PINN_HI(1, 4)))      // r05: + pure third / fourth derivatives of both axes (C = 10)
PINN_INSTANTIATE_F64(f64_d3_v, 3, 0x0, 0ull, 0, 0u)
PINN_INSTANTIATE_F64(f64_d3_h, 3, 0x7, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2)), 6, 0u)
PINN_INSTANTIATE_F64(f64_d3_h4, 3, 0x7, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 0, 1) | PINN_PAIR(2, 0, 2) | PINN_PAIR(3, 1, 1) | PINN_PAIR(4, 1, 2) | PINN_PAIR(5, 2, 2)), 6, (PINN_HI(0, 4) | PINN_HI(1, 4) | PINN_HI(2, 4)))      // r05 (C = 16)
PINN_INSTANTIATE_F64(f64_d4_v, 4, 0x0, 0ull, 0, 0u)
PINN_INSTANTIATE_F64(f64_d4_p, 4, 0xf, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1) | PINN_PAIR(2, 2, 2) | PINN_PAIR(3, 3, 3)), 4, 0u)
