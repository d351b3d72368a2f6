// 2-D, 4x64-class nets (3 hidden->hidden layers, padded width 64): Poisson interior jet set {u, u_x, u_y, u_xx, u_yy}
#include "spec_registry.hpp"
PINN_INSTANTIATE(h64n3d2_lap, 64, 3, 2, 0x3, (PINN_PAIR(0, 0, 0) | PINN_PAIR(1, 1, 1)), 2, 1)
