// family 2, 4x64 nets with pure third / fourth derivatives along x (Kuramoto-Sivashinsky jet set: u, u_t, u_x, u_xx, u_xxx, u_xxxx)
#include "spec_registry.hpp"
PINN_INSTANTIATE2_HI(f2_h64n3d2_ks, 64, 3, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 1, PINN_HI(1, 4))
PINN_INSTANTIATE2_HI(f2_h64n3d2_ks0, 64, 3, 2, 0x3, PINN_PAIR(0, 0, 0), 1, 1, PINN_HI(0, 4))
