// 2-D (t,x), 4x64-class nets: Burgers interior jet set {u, u_t, u_x, u_xx}
#include "spec_registry.hpp"
PINN_INSTANTIATE(h64n3d2_burg, 64, 3, 2, 0x3, PINN_PAIR(0, 1, 1), 1, 1)
